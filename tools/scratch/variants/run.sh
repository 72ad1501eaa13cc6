#!/bin/bash
# times the headline workload (and float64) with every variant library in this directory
cd "$(dirname "$0")/../../.."
for lib in tools/scratch/variants/lib_*.so; do
  for w in m3v m3v64; do
    FDTDHIP_LIBRARY=$PWD/$lib timeout -k 10 100 python bench.py --workload $w --no-cpu-baseline --steps 40 --warmup 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', '$w', d['value'], d['roofline']['avg_launch_ms'])" || exit 1
  done
done
