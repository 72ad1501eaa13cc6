#!/bin/bash
# Random-scene sweeps with the marching step applied wherever a plan allows it
# (FDTDHIP_FUSED=2): single grids in both precisions, large extents, crowded scenes, state
# kept between runs, slabs on 2, 3 and 5 ranks.  Output: gpurun_out/sweep_march.log
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
log=gpurun_out/sweep_march.log
: > $log
export FDTDHIP_FUSED=2
B=${SEED_BASE:-0}   # added to every first seed: other scenes on every run
run() { echo "== $*" >> $log; timeout -k 10 900 "$@" >> $log 2>&1 || { echo "FAILED: $*" >> $log; }; tail -1 $log; }
run python tools/fuzz_parity.py $((B + 20000)) 300 float64
run python tools/fuzz_parity.py $((B + 21000)) 150 float32
run python tools/fuzz_parity.py $((B + 22000)) 60 float64 big
run python tools/fuzz_parity.py $((B + 23000)) 40 float32 big
run python tools/fuzz_parity.py $((B + 24000)) 80 float64 crowded
run python tools/fuzz_operations.py $((B + 25000)) 200
run python tools/fuzz_slab.py 2 $((B + 26000)) 80 gpu
run python tools/fuzz_slab.py 3 $((B + 27000)) 80 gpu
FUZZ_BIG=1 run python tools/fuzz_slab.py 2 $((B + 28000)) 24 gpu
FUZZ_FLOAT32=1 run python tools/fuzz_slab.py 3 $((B + 29000)) 60 gpu
FUZZ_THIN_X=5 run python tools/fuzz_slab.py 5 $((B + 30000)) 30 gpu
grep -c FAILED $log
