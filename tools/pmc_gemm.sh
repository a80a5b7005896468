#!/bin/bash
# PMC passes for the GEMM kernels (separate --pmc runs, kernel-trace only).  usage: pmc_gemm.sh <tile> M K N [lda]
export TMPDIR=/tmp
t=$1; shift
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  d=gpurun_out/pmc_tmp
  rm -rf $d
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o p -- python3 tools/gemm_pmc.py $t "$@" > /dev/null 2>&1
  python3 - "$d" "$t $*" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(list)
for fn in f:
    for r in csv.DictReader(open(fn)):
        if "gemm" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], {k: round(sum(v) / len(v) / 1e6, 2) for k, v in agg.items()}, "(millions)")
PY
done
