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
net = HipHifigan(cfg, syn.hifigan_weights(cfg), dev)
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
