#!/bin/bash
# Dynamic instruction counts of the step kernels of tools/march_cost.py cases (one rocprofv3 pass).
# usage: tools/pmc_insts.sh <n> <dtype> case...     -> gpurun_out/pmc_insts.txt
set -e
cd "$(dirname "$0")/.."
out=gpurun_out/pmc_insts
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES SQ_INSTS_SMEM SQ_WAIT_INST_ANY --output-format csv -d "$out" -- python3 tools/march_cost.py "$@" > "$out/run.log" 2>&1 || { echo "pass failed"; tail -5 "$out/run.log"; exit 1; }
python3 - "$out" <<'PY' | tee gpurun_out/pmc_insts.txt
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "fdtdhip::step_march" not in k and "fdtdhip::bulk_" not in k: continue
        agg[k.split("(")[0][-70:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name in sorted(agg):
    c = {k: sum(v) / len(v) for k, v in agg[name].items()}
    w = c.get("SQ_WAVES", 1) or 1
    print(name, "launches", len(agg[name]["SQ_WAVES"]), "waves", int(w),
          " per wave:", {k[3:]: round(v / w, 1) for k, v in c.items() if k != "SQ_WAVES"})
PY
