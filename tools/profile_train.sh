#!/bin/bash
# rocprofv3 kernel trace of the mask-decoder training step (tools/bench_train.py), default precision (f32-class); $1 = batch
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_train_r3
rm -rf "$out"; mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o train -- python3 tools/bench_train.py ${1:-64} 4 > "$out/stdout.log" 2>&1
python3 tools/summarize_rocprof.py "$out" | head -45
tail -1 "$out/stdout.log"
