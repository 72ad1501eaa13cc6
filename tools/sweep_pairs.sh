#!/bin/bash
# Random-scene sweeps with marching forced (FDTDHIP_FUSED=2), so that every plan that allows it
# takes two steps per pass: random plans three ways bit for bit (fuzz_pairs.py), larger random
# scenes against the oracle, operation sequences between runs, slabs on 2 and 3 ranks, the
# adjoint against finite differences, training loops, slab gradients.
# Output: gpurun_out/sweep_pairs.log.  SEED_BASE: added to every first seed.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
log=gpurun_out/sweep_pairs.log
: > $log
export FDTDHIP_FUSED=2
B=${SEED_BASE:-0}
run() { echo "== $*" >> $log; timeout -k 10 500 "$@" >> $log 2>&1 || { echo "FAILED: $*" >> $log; }; tail -1 $log; }
run python tools/fuzz_pairs.py $((B + 10000)) 600 float32
run python tools/fuzz_pairs.py $((B + 11000)) 300 float64
run python tools/fuzz_parity.py $((B + 22000)) 80 float64 big
run python tools/fuzz_parity.py $((B + 23000)) 60 float32 big
run python tools/fuzz_parity.py $((B + 24000)) 40 float64 big crowded
run python tools/fuzz_operations.py $((B + 25000)) 100 big
FUZZ_BIG=1 run python tools/fuzz_slab.py 2 $((B + 28000)) 24 gpu
FUZZ_BIG=1 FUZZ_FLOAT32=1 run python tools/fuzz_slab.py 3 $((B + 29000)) 24 gpu
run python tools/fuzz_adjoint.py $((B + 31000)) 60 big
run python tools/fuzz_adjoint.py $((B + 32000)) 100
run python tools/fuzz_training.py $((B + 33000)) 40
FUZZ_BIG=1 run python tools/fuzz_slab_adjoint.py 2 $((B + 34000)) 16
FUZZ_BIG=1 run python tools/fuzz_slab_adjoint.py 3 $((B + 35000)) 16
echo "failed: $(grep -c FAILED $log)"
