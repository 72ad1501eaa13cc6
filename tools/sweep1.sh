#!/bin/bash
# kernel strategy sweep on the 512^3 float32 update
set -e
cd "$(dirname "$0")/.."
for v in generic band march1 march2 march4; do
  FDTDHIP_BULK=$v timeout -k 5 60 ./tools/kbench 512 512 512 f32 20
done
for xc in 16 32 128 512; do
  FDTDHIP_BULK=march2 FDTDHIP_XCHUNK=$xc timeout -k 5 60 ./tools/kbench 512 512 512 f32 20
done
for v in generic band march1 march2; do
  FDTDHIP_BULK=$v timeout -k 5 60 ./tools/kbench 256 256 256 f64 40
done
for v in band march2; do
  FDTDHIP_BULK=$v timeout -k 5 60 ./tools/kbench 512 512 512 f32 20 10
done
