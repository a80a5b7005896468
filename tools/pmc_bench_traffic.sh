#!/bin/bash
# HBM traffic of the dominant kernel in the bench command, per MI355X_MICROARCH.md §HBM: separate --pmc passes
# (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2), kernel-trace only; FETCH_SIZE x2 (gfx950 counts 128-B requests
# as 64 B for wide coalesced reads), units KiB.  Writes gpurun_out/r02_gemm_traffic.json (copy it to profiles/).
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  d=gpurun_out/pmc_bench_$c
  rm -rf $d
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-traffic ${BENCH_ARGS} > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, json, collections
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for fn in glob.glob(f"gpurun_out/pmc_bench_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] == c:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        res.setdefault(k, {})[c] = (sum(v) / len(v), len(v))
out = {}
for k, d in res.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        fetch_kib, n = d["FETCH_SIZE"]
        write_kib, _ = d["WRITE_SIZE"]
        out[k[:80]] = {"launches": n, "fetch_bytes_per_launch_corrected": 2 * fetch_kib * 1024, "write_bytes_per_launch": write_kib * 1024,
                       "hbm_bytes_per_launch": 2 * fetch_kib * 1024 + write_kib * 1024}
json.dump(out, open("gpurun_out/r02_gemm_traffic.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:8]:
    print(k[:60], v)
PY
