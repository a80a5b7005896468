#!/bin/bash
# rocprofv3 kernel-trace summary of the bench command (run on the GPU box through gpurun).
# usage: tools/profile_bench.sh <tag>   -> gpurun_out/prof_<tag>/ ... and a per-kernel summary text file
set -e
tag=${1:-r02}
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o bench -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-traffic ${BENCH_ARGS} > "$out/bench_stdout.log" 2>&1
python3 tools/summarize_rocprof.py "$out" > "$out/kernel_summary.txt"
python3 tools/summarize_rocprof.py "$out" --by-grid > "$out/kernel_summary_by_grid.txt"
cat "$out/kernel_summary.txt"
