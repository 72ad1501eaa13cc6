#!/bin/bash
# HBM-traffic counters of the adjoint kernels (tools/bench_adjoint.py <workload>): one rocprofv3
# pass per counter.   usage: tools/pmc_adjoint.sh [workload]   -> gpurun_out/pmc_adjoint.txt
set -e
cd "$(dirname "$0")/.."
w=${1:-xlarge}
out=gpurun_out/pmc_adjoint
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
for grp in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 240 rocprofv3 --pmc $grp --output-format csv -d "$out/$grp" -- python3 tools/bench_adjoint.py "$w" > "$out/$grp.log" 2>&1 || { echo "pass $grp failed"; tail -3 "$out/$grp.log"; exit 1; }
done
python3 - "$out" <<'PY' | tee gpurun_out/pmc_adjoint.txt
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "fdtdhip::" not in k: continue
        agg[k.split("(")[0].replace("void fdtdhip::", "")[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name in sorted(agg):
    c = {k: sum(v[len(v)//2:]) / len(v[len(v)//2:]) for k, v in agg[name].items()}
    f, w = c.get("FETCH_SIZE", 0), c.get("WRITE_SIZE", 0)
    print(f"{name}: FETCH_SIZE {f:.0f} KB, WRITE_SIZE {w:.0f} KB, bytes = (2*FETCH+WRITE)*1024 = {(2*f+w)*1024/1e9:.3f} GB over {len(agg[name].get('FETCH_SIZE', []))} launches")
PY
