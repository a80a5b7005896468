#!/bin/bash
# Mean PMC counters per launch of the kernels whose name contains $1, over the command "$2 ..." (a python script + arguments): separate
# rocprofv3 --pmc passes (one counter set each, --kernel-trace only), the program directly after `--`.  Run on the GPU box through gpurun.
#   tools/pmc_kernel.sh conv_taps_x3 tools/bench_hifigan.py 256
pat=$1; shift
cd /tmp; export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_$pat
rm -rf $OUT; mkdir -p $OUT
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$tag -o p -- python3 /root/repo/$1 "${@:2}" > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
per = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "$pat" in r["Kernel_Name"]:
            per[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in per.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
