#!/bin/bash
# Where the waves of the whole-step kernels spend their cycles (one rocprofv3 pass of SQ
# counters on `python3 bench.py --workload <w>`): parked at s_waitcnt / barriers (SQ_WAIT_ANY),
# stalled at issue (SQ_WAIT_INST_ANY), issuing (SQ_ACTIVE_INST_ANY); quad-cycles summed over waves.
# usage: tools/pmc_wait.sh <workload> [bench args]     -> gpurun_out/pmc_wait_<workload>.txt
set -e
cd "$(dirname "$0")/.."
w=$1; shift
case " $* " in *" --gpus "*) echo "profile one rank only: --gpus is not accepted here" >&2; exit 2;; esac
out=gpurun_out/pmc_wait_$w
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d "$out" -- python3 bench.py --workload "$w" --no-cpu-baseline --steps 10 --warmup 2 "$@" > "$out/run.log" 2>&1 || { echo "pass failed"; tail -5 "$out/run.log"; exit 1; }
python3 - "$out" "$w" <<'PY' | tee gpurun_out/pmc_wait_$w.txt
import csv, glob, sys, collections
out, w = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "fdtdhip::step_march" not in k and "fdtdhip::shell_" not in k and "fdtdhip::bulk_" not in k: continue
        agg[k.split("(")[0].replace("void fdtdhip::", "")[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name in sorted(agg):
    c = {k: sum(v[len(v)//2:]) / len(v[len(v)//2:]) for k, v in agg[name].items()}
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    print(f"{w} {name}: waves {c.get('SQ_WAVES', 0):.0f}, of the wave cycles: parked (waitcnt, barrier) "
          f"{c.get('SQ_WAIT_ANY', 0) / wc:.3f}, stalled at issue {c.get('SQ_WAIT_INST_ANY', 0) / wc:.3f}, "
          f"issuing {c.get('SQ_ACTIVE_INST_ANY', 0) / wc:.3f} (VALU {c.get('SQ_ACTIVE_INST_VALU', 0) / wc:.3f}); "
          f"VALU instructions per wave {c.get('SQ_INSTS_VALU', 0) / (c.get('SQ_WAVES', 1) or 1):.0f}")
PY
