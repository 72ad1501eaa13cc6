#!/bin/bash
# rocprofv3 kernel trace + stats of the default bench command; summary copied to profiles/.
# usage: tools/profile_bench.sh <tag> [bench args]
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
case " $* " in *" --gpus "*) echo "profile one rank only: --gpus is not accepted here" >&2; exit 2;; esac
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 bench.py --no-cpu-baseline "$@" > "$out/bench.log" 2>&1
grep "^{\"metric\"" "$out/bench.log" | cut -c1-300
f=$(find "$out" -name "*kernel_stats.csv" | head -1)
mkdir -p gpurun_out/profiles_$tag
cp "$f" gpurun_out/profiles_$tag/kernel_stats.csv
grep "^{\"metric\"" "$out/bench.log" > gpurun_out/profiles_$tag/bench_line.json
head -8 "$f"
