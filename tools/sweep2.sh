#!/bin/bash
cd "$(dirname "$0")/.."
for shape in "512 512 512 f32 30" "512 512 512 f32 30 10" "256 256 256 f64 60" "256 256 256 f64 60 10" "512 512 512 f64 16"; do
  for v in generic rows+nont rows+ntl rows+nts rows+nt flat+nont flat+ntl flat+nts flat+nt; do
    FDTDHIP_BULK=$v timeout -k 5 60 ./tools/kbench $shape
  done
done
