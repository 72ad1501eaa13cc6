#!/bin/bash
# usage: tools/adj_shell_cost.sh <size> <axes> [<axes> ...]   (kernel times of the reverse sweep per face set)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
n=$1; shift
for axes in "$@"; do
  out=gpurun_out/prof_adjshell_$axes
  rm -rf "$out"; mkdir -p "$out"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o t -- python3 tools/adj_shell_cost.py $axes $n $OBJ > "$out/log.txt" 2>&1 || exit 1
  echo "== faces $axes, $n^3"
  python tools/kstats.py "$out" adj_ step_march shell_ | head -8
done
