#!/bin/bash
cd /tmp; export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_attn_bwd
rm -rf $OUT; mkdir -p $OUT
python3 /root/repo/tools/attn_bwd_perf.py 2>&1 | grep -v amdgpu
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$tag -o p -- python3 /root/repo/tools/attn_bwd_perf.py > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
per = collections.defaultdict(list)
for fn in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "attention_bwd_" in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(per.items()):
    print(f"    {c:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
