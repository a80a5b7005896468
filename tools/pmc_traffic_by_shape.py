#!/usr/bin/env python3
"""HBM traffic of the dominant GEMM kernel PER LAUNCH SHAPE inside the explanation step: two `rocprofv3 --pmc` passes (FETCH_SIZE,
WRITE_SIZE; separate runs, --kernel-trace only, as MI355X_MICROARCH.md prescribes; FETCH_SIZE x2 on gfx950, both in KiB) of
`bench.py --steps 1`, dispatches grouped by grid size, next to the ALGORITHMIC bytes of the launches with that grid (every operand
and result once; split planes = 4 B per element) taken from an unprofiled run's launch list.
    python tools/pmc_traffic_by_shape.py [f32|f16]  ->  stdout table (profiles/r03_gemm_traffic_by_shape.txt)"""
import collections, csv, glob, json, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
KERNEL = "gemm_x3_kernel<128, 128, 2, 2, true>" if prec == "f32" else "gemm_f16_kernel<128, 128, 2, 2, 4, true>"

if len(sys.argv) > 2 and sys.argv[2] == "--shapes":                      # child: one unprofiled step, print the launch list
    sys.path.insert(0, os.path.join(ROOT, "xai-audio-deepfakes_amd"))
    import torch
    from addvisor_hip import gemm as G, pipeline as P, synthetic as syn
    torch.set_grad_enabled(False)
    dev = torch.device("cuda:0")
    cfg = syn.base_config()
    pipe = P.ExplainPipeline(cfg, syn.embedder_weights(cfg), *syn.logreg_weights(cfg.hidden_size), syn.unet_weights(), dev, audio_length=4, precision=prec)
    w = syn.make_clips(64, 64000).to(dev)
    pipe.explain(w); torch.cuda.synchronize()
    G.PROFILE.reset(True)
    pipe.explain(w); torch.cuda.synchronize()
    out = []
    for (M, N, K, nz, name), (ms, fl, n) in G.PROFILE.by_shape().items():
        out.append(dict(M=M, N=N, K=K, nz=nz, tile=name, launches=n, us=1e3 * ms / n))
    print("SHAPES " + json.dumps(out))
    sys.exit(0)

tmp = tempfile.mkdtemp(prefix="advh_pmcs_", dir="/tmp")
per = collections.defaultdict(lambda: collections.defaultdict(list))
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    d = os.path.join(tmp, counter)
    subprocess.run(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
                    os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--no-traffic", "--precision", prec],
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900, env=dict(os.environ, TMPDIR="/tmp"))
    for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(fn)):
            if row["Counter_Name"] == counter and KERNEL in row["Kernel_Name"]:
                per[int(row["Grid_Size"]) // 256][counter].append(float(row["Counter_Value"]))
shutil.rmtree(tmp, ignore_errors=True)
r = subprocess.run([sys.executable, os.path.abspath(__file__), prec, "--shapes"], capture_output=True, text=True)
shapes = json.loads([l for l in r.stdout.splitlines() if l.startswith("SHAPES ")][0][7:])
bpe = 4 if prec == "f32" else 2
print(f"# {KERNEL}: HBM bytes per launch by launch shape, explanation step B = 64, {prec}\n# tiles  M        N     K     launches/step  us     algorithmic MB   measured MB (2*FETCH+WRITE)   ratio")
tot_a = tot_m = 0.0
for s in sorted((s for s in shapes if s["tile"].startswith("128x128+")), key=lambda s: -s["us"] * s["launches"]):
    tiles = -(-s["M"] // 128) * -(-s["N"] // 128) * s["nz"]
    alg = (s["M"] * s["K"] + s["N"] * s["K"] + s["M"] * s["N"]) * bpe * s["nz"]
    f, w = per.get(tiles, {}).get("FETCH_SIZE", []), per.get(tiles, {}).get("WRITE_SIZE", [])
    if not f or not w:
        print(f"{tiles:6d} {s['M']:8d} {s['N']:5d} {s['K']:5d} {s['launches']:4d}   (no counter rows for this grid)")
        continue
    meas = 2 * 1024 * sum(f) / len(f) + 1024 * sum(w) / len(w)
    tot_a += alg * s["launches"]; tot_m += meas * s["launches"]
    print(f"{tiles:6d} {s['M']:8d} {s['N']:5d} {s['K']:5d} {s['launches']:4d} {s['us']:10.1f} {alg / 1e6:12.1f} {meas / 1e6:14.1f} {meas / alg:10.2f}")
print(f"# all affine-row launches of the step: algorithmic {tot_a / 1e9:.2f} GB, measured {tot_m / 1e9:.2f} GB, ratio {tot_m / max(tot_a, 1):.2f}")
