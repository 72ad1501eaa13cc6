#!/bin/bash
# The same sweeps as sweep_march.sh with the library's own choice of kernels.
# Output: gpurun_out/sweep_default.log
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
log=gpurun_out/sweep_default.log
: > $log
B=${SEED_BASE:-0}   # added to every first seed: other scenes on every run
run() { echo "== $*" >> $log; timeout -k 10 900 "$@" >> $log 2>&1 || { echo "FAILED: $*" >> $log; }; tail -1 $log; }
run python tools/fuzz_parity.py $((B + 50000)) 300 float64
run python tools/fuzz_parity.py $((B + 51000)) 150 float32
run python tools/fuzz_parity.py $((B + 52000)) 60 float64 big
run python tools/fuzz_parity.py $((B + 53000)) 40 float32 big
run python tools/fuzz_parity.py $((B + 54000)) 80 float64 crowded
run python tools/fuzz_parity.py $((B + 55000)) 100 float64 extras
run python tools/fuzz_operations.py $((B + 56000)) 200
run python tools/fuzz_slab.py 3 $((B + 57000)) 80 gpu
run python tools/fuzz_adjoint.py $((B + 58000)) 100
run python tools/fuzz_training.py $((B + 59000)) 60
run python tools/fuzz_slab_adjoint.py 3 $((B + 60000)) 40
echo "failed: $(grep -c FAILED $log)"
