#!/bin/bash
# HBM-traffic counters of the bulk kernels of one bench workload: one rocprofv3 pass per
# counter (FETCH_SIZE, WRITE_SIZE), the program being python3 itself.
# usage: tools/pmc_bench.sh <workload> [bench args]     -> gpurun_out/pmc_<workload>.txt
set -e
cd "$(dirname "$0")/.."
w=$1; shift
case " $* " in *" --gpus "*) echo "profile one rank only: --gpus is not accepted here" >&2; exit 2;; esac
out=gpurun_out/pmc_$w
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
for grp in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 240 rocprofv3 --pmc $grp --output-format csv -d "$out/$grp" -- python3 bench.py --workload "$w" --no-cpu-baseline --steps 10 --warmup 2 "$@" > "$out/$grp.log" 2>&1 || { echo "pass $grp failed"; tail -3 "$out/$grp.log"; exit 1; }
done
python3 - "$out" "$w" <<'PY' | tee gpurun_out/pmc_$w.txt
import csv, glob, sys, collections
out, w = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        name = next((n for n in ("bulk_E", "bulk_H", "step_march2_rows", "step_march2", "march2_detect_first_step",
                               "step_march_box", "step_march", "step_fused", "shell_E", "shell_H",
                               "detect_cells_batch_outside")
                     if "fdtdhip::" + n in k), None)
        if name is None: continue
        agg[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
tot = {}
# the dominant kernels: the two-step kernel where the run took it (bytes per launch of two steps;
# with --steps 12 --warmup 4 the later half of the shell launches all belong to pairs of steps,
# two launches of shell_E and two of shell_H per pair), else the whole-step kernel, else the
# two bulk kernels
whole = [n for n in ("step_march2_rows", "step_march2", "step_march_box", "step_march", "step_fused") if n in agg]
dominant = whole[:1] or ["bulk_E", "bulk_H"]
for name in sorted(agg):
    for c, v in sorted(agg[name].items()):
        v = v[len(v)//2:]  # later launches (steady state)
        if name in dominant:
            tot.setdefault(c, []).append(sum(v)/len(v))
        print(f"{w} {name} {c}: mean {sum(v)/len(v):.6g} KB over {len(v)} launches")
f = sum(tot["FETCH_SIZE"]) / len(tot["FETCH_SIZE"]); wr = sum(tot["WRITE_SIZE"]) / len(tot["WRITE_SIZE"])
print(f"{w} per launch (mean over {' and '.join(dominant)}): FETCH_SIZE {f:.0f} KB, WRITE_SIZE {wr:.0f} KB, bytes = (2*FETCH+WRITE)*1024 = {(2*f+wr)*1024:.0f}")
PY
