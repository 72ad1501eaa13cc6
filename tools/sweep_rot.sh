#!/bin/bash
# Row-kernel z-segment rotation sweep (FDTDHIP_ROT = log2 of block rows per rotation step).
# Each configuration is run twice: run-to-run spread on this machine is about 3 %.
cd "$(dirname "$0")/.."
for u in 1 0; do
  for shape in "512 512 512 f32 30" "512 512 512 f32 30 10" "512 512 512 f64 16"; do
    for rot in 0 2 4 6 30; do
      for rep in 1 2; do
        echo -n "rot=$rot "
        KB_UNIFORM=$u FDTDHIP_ROT=$rot FDTDHIP_BULK=${1:-rows+nt} timeout -k 5 60 ./tools/kbench $shape | awk '{print $3,$4,$5,$6,$(NF-4),$(NF-3)}'
      done
    done
  done
done
