"""Diagnostic: end-to-end gradient agreement of the HIP U-Net training step vs the fp32 oracle as a function of the
LeakyReLU slope (slope 1.0 removes the kink: any remaining disagreement would be a kernel error, not fp16 slope flips)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xai-audio-deepfakes_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn.functional as F
import addvisor_hip.unet_train as UT
import test_gpu_unet_train as T
dev = torch.device("cuda:0")
for slope in (0.2, 0.6, 1.0):
    UT.SLOPE = slope; T.SLOPE = slope
    import oracle.unet_ref as R
    orig = F.leaky_relu
    R.F.leaky_relu = lambda x, s=0.2, **kw: orig(x, slope)          # the oracle hard-codes 0.2
    try:
        net, params, mag, target, dmask, ref_mask, ref_grads = T.setup(dev, 3, 64, 24, 7)
        mask = net.forward(mag.to(dev), H=64, W=24)
        grads = net.backward(dmask.to(dev))
    finally:
        R.F.leaky_relu = orig
    worst = (1.0, "")
    for k, (r32, rq) in ref_grads.items():
        if r32 is None or grads[k].abs().max() == 0: continue
        c = F.cosine_similarity(grads[k].cpu().flatten().double(), r32.flatten().double(), dim=0).item()
        if c < worst[0]: worst = (c, k)
    print(f"slope {slope}: mask err {(mask.cpu()-ref_mask).abs().max():.2e}; worst cosine vs fp32 oracle {worst[0]:.6f} at {worst[1]}", flush=True)
