#!/bin/bash
# HBM-traffic counters of the bulk kernels, one rocprofv3 pass per counter group.
# usage: tools/pmc.sh <outdir> <kbench args...>     (FDTDHIP_BULK etc. from the environment)
set -e
cd "$(dirname "$0")/.."
out=$1; shift
mkdir -p "$out"
export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  tag=$(echo $grp | tr ' ' '+')
  timeout -s KILL 120 rocprofv3 --pmc $grp --output-format csv -d "$out/$tag" -- ./tools/kbench "$@" > "$out/$tag.log" 2>&1 || { echo "pass $tag failed"; tail -3 "$out/$tag.log"; }
done
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- ./tools/kbench "$@" > "$out/trace.log" 2>&1 || true
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "bulk" not in k: continue
        name = "E" if "bulk_E" in k else "H"
        agg[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name in sorted(agg):
    for c, v in sorted(agg[name].items()):
        v = v[len(v)//2:]  # later launches (steady state)
        print(f"bulk_{name} {c}: mean {sum(v)/len(v):.6g} over {len(v)} launches")
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "bulk" in row["Name"]:
            print("trace:", row["Name"][:60], "calls", row["Calls"], "avg_ns", row["AverageNs"])
PY
