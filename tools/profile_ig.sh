#!/bin/bash
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_ig
rm -rf "$out"; mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o ig -- python3 tools/bench_ig.py 16 160 ${1:-large} > "$out/stdout.log" 2>&1
python3 tools/summarize_rocprof.py "$out" | head -30
tail -1 "$out/stdout.log"
