"""Ablation script of the reverted conv_taps_x3 experiment (commit 7374ba2; profiles/r03_conv_taps_x3_experiment.txt): needs that commit's kernel."""
import os, sys, torch, ctypes as C
sys.path.insert(0, "/root/repo/xai-audio-deepfakes_amd")
from addvisor_hip import _lib, gemm as G
_lib.init()
dev = torch.device("cuda:0")
B, T, k, dil = 256, 32128, 11, 5
src = G.Map1D(B, T, 64, 32, split=True).alloc(dev); dst = G.Map1D(B, T, 64, 32, split=True).alloc(dev)
src.t.normal_()
w = torch.randn(64, 64, k) * 0.05
plan = G.plan_conv1d_taps(src, dst, w, torch.zeros(64), dilation=dil, act="leaky", slope=0.1, device=dev)
for exp in (0, 2, 3, 4):
    plan.desc.pre_act = exp
    for _ in range(2): plan.run(src.t, out_h=dst.t)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): plan.run(src.t, out_h=dst.t)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"exp {exp}: {ms:.3f} ms, {plan.flops / ms / 1e9:.1f} TFLOP/s", flush=True)
