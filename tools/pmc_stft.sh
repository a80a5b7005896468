#!/bin/bash
# Where do the waves of the masked inverse STFT spend their cycles?  rocprofv3 PMC passes (counters only, --kernel-trace) over
# tools/stft_perf.py; SQ_WAVE_CYCLES = WAIT_ANY (parked: s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY.
cd /tmp; export TMPDIR=/tmp
OUT=${1:-/root/repo/gpurun_out/pmc_stft}
mkdir -p $OUT
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$tag -o p -- python3 /root/repo/tools/stft_perf.py 8 > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
per = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if "stft" in k:
            per[k.split("(")[0][:60] + " grid=" + r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(per.items()):
    print(k)
    for c, v in sorted(d.items()):
        print(f"    {c:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
