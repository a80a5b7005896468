#!/bin/bash
# LDS bank-conflict share per kernel in the bench command (SQ_LDS_BANK_CONFLICT = extra LDS cycles, SQ_LDS_IDX_ACTIVE = all
# LDS-array cycles; MI355X_MICROARCH.md, LDS section).  Counter pass only (kernel-trace), run on the GPU box through gpurun.
export TMPDIR=/tmp
d=gpurun_out/pmc_lds
rm -rf $d
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $d -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-traffic ${BENCH_ARGS} > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for fn in glob.glob("gpurun_out/pmc_lds/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        agg[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
rows = [(v.get("SQ_LDS_IDX_ACTIVE", 0.0), v.get("SQ_LDS_BANK_CONFLICT", 0.0), k) for k, v in agg.items() if v.get("SQ_LDS_IDX_ACTIVE", 0.0) > 0]
print("# LDS bank conflicts per kernel over the bench command (3 steps): conflict cycles / LDS-array cycles")
for act, conf, k in sorted(rows, reverse=True)[:16]:
    print(f"{100.0 * conf / act:6.2f} %  {act / 1e6:10.1f} M LDS cycles  {k[:110]}")
PY
