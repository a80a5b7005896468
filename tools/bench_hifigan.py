#!/usr/bin/env python3
"""BASELINE config 3: HiFi-GAN V1 vocoder, batch 256 x 4 s (mel [256, 80, 251]) on one MI355X."""
import json, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xai-audio-deepfakes_amd"))
import torch
from addvisor_hip import gemm as G, synthetic as syn
from addvisor_hip.hifigan import HipHifigan
torch.set_grad_enabled(False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T, steps = 251, 5
dev = torch.device("cuda:0")
cfg = syn.HifiganConfig()
net = HipHifigan(cfg, syn.hifigan_weights(cfg), dev, precision=os.environ.get("HIFIGAN_PRECISION", "f32"))
mel = torch.randn(B, 80, T, device=dev) * 2 - 4
net.decode_batch(mel); torch.cuda.synchronize()
G.PROFILE.reset(True)
t0 = time.perf_counter()
for _ in range(steps):
    w = net.decode_batch(mel)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
ms, fl, n = G.PROFILE.summary()
print(json.dumps({"workload": f"HiFi-GAN V1 decode_batch, B={B}, T={T} mel frames", "clips_per_s": round(B / dt, 1), "ms_per_batch": round(dt * 1e3, 2),
                  "gflop_per_clip": round(net.flops(B, T) / B / 1e9, 1), "tflops": round(net.flops(B, T) / dt / 1e12, 1),
                  "gemm128_tflops": round(fl / (ms * 1e-3) / 1e12, 1) if ms else None}))
if os.environ.get("HIFIGAN_DETAIL"):
    # per-launch timing (HIP events on the current stream), grouped by (kernel kind, channels, taps)
    ws = net._workspace(B, T)
    agg = {}
    for kind, plan, src, resid, dst, dst2 in ws["steps"]:
        if kind != "gemm":
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            plan.run(src.t, out_h=dst.t, resid=None if resid is None else resid.t, out_h2=None if dst2 is None else dst2.t)
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1) / 3
        if isinstance(plan, G.ResblockPairX3Plan):
            key = ("fused_x3", 32, plan.desc.k, plan.desc.dil)
        elif isinstance(plan, G.ResblockPairPlan):
            key = ("fused", plan.Cn, plan.desc.k, plan.desc.dil)
        elif isinstance(plan, G.TapsPlan):
            key = ("taps_x3" if getattr(plan, "split", False) else "taps", plan.Cn, plan.desc.ntap, abs(plan.desc.toff[0]), resid is not None)
        else:
            key = ("gemm", plan.desc.N, plan.K, G.TILE_NAMES[plan.tile], int(plan.ktab_host[plan.desc.a_c0[0] * 0 + max(1, plan.K // 8 // max(1, plan.K // (8 * 8))) - 1] if False else plan.ktab_host[-1]), resid is not None)
        a = agg.setdefault(key, [0, 0.0, 0.0])
        a[0] += 1; a[1] += ms; a[2] += plan.flops
    for key, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{str(key):40s} n={n:3d} total {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s")
