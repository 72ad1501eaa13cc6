#!/bin/bash
# float64 lane mapping / non-temporal choice, each run twice.
cd "$(dirname "$0")/.."
for u in 1 0; do
  for shape in "512 512 512 f64 16" "256 256 256 f64 60 10"; do
    for v in flat+nts rows+nts rows+nt rows+nont; do
      for rep in 1 2; do
        echo -n "$v "
        KB_UNIFORM=$u FDTDHIP_ROT=2 FDTDHIP_BULK=$v timeout -k 5 60 ./tools/kbench $shape | awk '{print $3,$4,$5,$6,$(NF-4),$(NF-3)}'
      done
    done
  done
done
