#!/bin/bash
# Differentiable runs with the marching step applied to the forward pass wherever a plan
# allows it (FDTDHIP_FUSED=2): gradients against finite differences, training loops, slabs.
# Output: gpurun_out/sweep_march_adjoint.log
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
log=gpurun_out/sweep_march_adjoint.log
: > $log
export FDTDHIP_FUSED=2
B=${SEED_BASE:-0}   # added to every first seed: other scenes on every run
run() { echo "== $*" >> $log; timeout -k 10 900 "$@" >> $log 2>&1 || { echo "FAILED: $*" >> $log; }; tail -1 $log; }
run python tools/fuzz_adjoint.py $((B + 40000)) 150
run python tools/fuzz_adjoint.py $((B + 41000)) 30 big
run python tools/fuzz_adjoint.py $((B + 42000)) 50 crowded
run python tools/fuzz_training.py $((B + 43000)) 100
run python tools/fuzz_slab_adjoint.py 3 $((B + 44000)) 60
grep -c FAILED $log
