#!/bin/bash
# rocprofv3 kernel-trace summary of the HiFi-GAN bench (BASELINE config 3); run on the GPU box through gpurun.
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_hifigan
rm -rf "$out"; mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o hf -- python3 tools/bench_hifigan.py > "$out/stdout.log" 2>&1
python3 tools/summarize_rocprof.py "$out" | head -30
tail -1 "$out/stdout.log"
