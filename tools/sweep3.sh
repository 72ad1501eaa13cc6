#!/bin/bash
cd "$(dirname "$0")/.."
for v in flat+nt flat+nont flat+ntl flat+nts; do
  for c in 0 4 8 16 32; do
    echo -n "chunk=$c "; FDTDHIP_CHUNK=$c FDTDHIP_BULK=$v timeout -k 5 60 ./tools/kbench 512 512 512 f32 30
  done
done
