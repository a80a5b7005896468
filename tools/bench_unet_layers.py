#!/usr/bin/env python3
"""Per-layer device time of the inference U-Net (B = 64, 512 x 196) on one MI355X."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xai-audio-deepfakes_amd"))
import torch
from addvisor_hip import gemm as G, synthetic as syn
from addvisor_hip.unet import HipUNet
torch.set_grad_enabled(False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
net = HipUNet(syn.unet_weights(), dev, fuse_up=os.environ.get("UNET_FUSE_UP", "1") != "0", precision=os.environ.get("UNET_PRECISION", "f32"))
mag = torch.rand(B, 513, 199, device=dev)
net.forward(mag); torch.cuda.synchronize()
ws = net._workspace(B, 512, 196)
m = ws["maps"]
tot = 0.0
for plan, srcs, dst in ws["steps"]:
    a0 = m[srcs[0]].t; a1 = m[srcs[1]].t if len(srcs) > 1 else None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        plan.run(a0, a1, out_h=m[dst].t)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 5
    tot += ms
    kind = ("taps2d" if isinstance(plan, G.Taps2dPlan) else "upconv21" if isinstance(plan, G.UpconvTilePlan)
            else "conv53s21" if isinstance(plan, G.ConvS21TilePlan)
            else G.TILE_NAMES[plan.tile] + ("*" if isinstance(plan, G.PlanGroup) else ""))
    print(f"{'+'.join(srcs):8s} -> {dst:4s} {kind:9s} {ms*1e3:8.1f} us  {plan.flops/ms/1e9:7.1f} TFLOP/s  ({plan.flops/1e9:6.1f} GF)")
print(f"total GEMM-shaped layers {tot:.3f} ms")
