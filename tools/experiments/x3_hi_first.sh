#!/bin/bash
# Round-3 experiment: gemm_x3_kernel with the hi planes of every K-step issued first and the hi x hi MFMAs started before the lo
# planes have landed (-DADVH_X3_HI_FIRST, csrc/gemm.hip).  Builds a second library next to the product one and runs
# tools/gemm_x3_perf.py against both (same box, warm clock).  Run from the repo root on the GPU box (or build here, run there).
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
CS=$ROOT/xai-audio-deepfakes_amd/csrc
make -C $CS -j8 >/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -I$ROOT/include -I$CS -DADVH_X3_HI_FIRST -c $CS/gemm.hip -o $CS/build/gemm_hifirst.o
OBJS=$(ls $CS/build/*.o | grep -v "/gemm.o" | grep -v "gemm_hifirst.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS $CS/build/gemm_hifirst.o -o $ROOT/tools/experiments/libadvh_hifirst.so
if [ "$1" = "run" ]; then
    cd $ROOT
    echo "== product library"; python tools/gemm_x3_perf.py
    echo "== -DADVH_X3_HI_FIRST"; ADDVISOR_HIP_LIB=$ROOT/tools/experiments/libadvh_hifirst.so python tools/gemm_x3_perf.py
fi
