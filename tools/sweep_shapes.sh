#!/bin/bash
# Lane-mapping choice across grid shapes (float32): default (rows) against flat.
cd "$(dirname "$0")/.."
for shape in "256 256 256 f32 100" "384 384 384 f32 50" "500 500 500 f32 30" "128 1024 1024 f32 30" "1024 1024 128 f32 30" "1024 128 1024 f32 30" "640 640 96 f32 60" "200 300 1000 f32 40"; do
  for v in rows+nt flat+nt; do
    echo -n "$v "
    KB_UNIFORM=1 FDTDHIP_BULK=$v timeout -k 5 60 ./tools/kbench $shape | awk '{print $4,$5,$(NF-4),$(NF-3)}'
  done
done
