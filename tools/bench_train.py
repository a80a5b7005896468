#!/usr/bin/env python3
"""Training step of the mask decoder (train_addvisor.py:364-381) with the drop-in modules on one MI355X:
U-Net forward/backward (HIP training kernels) + HIP LMAC loss forward/backward (ISTFT x2, wav2vec2-base x2 with saves, their
input-gradient chain, ISTFT adjoint x2) + Adam.  usage: bench_train.py [B] [audio_length_s]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xai-audio-deepfakes_amd"))
os.environ.setdefault("ADDVISOR_EMBEDDER", "base")
import torch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
AL = int(sys.argv[2]) if len(sys.argv) > 2 else 4
os.environ["ADDVISOR_AUDIO_LENGTH"] = str(AL)
import addvisor, loss_function
from addvisor_hip import synthetic as syn
ap = loss_function.audio_processor
ap.audio_length = AL
dev = torch.device("cuda:0")
L = AL * 16000
w = syn.make_clips(B, L, seed=3).to(dev)
_, mag, ph = ap.compute_stft(w)
_, p = ap.classify(w)
T4 = 4 * (mag.shape[2] // 4)
x = mag[:, :512, :T4].unsqueeze(1).contiguous()
net = addvisor.UNet().to(dev); net.train()
loss = loss_function.LMACLoss().to(dev)
opt_m = torch.optim.Adam(net.parameters(), lr=3e-5); opt_w = torch.optim.Adam(loss.parameters(), lr=1e-4)

if os.environ.get("TRAIN_BENCHMARK"):
    torch.backends.cudnn.benchmark = True            # MIOpen find mode for the torch U-Net convolutions
if os.environ.get("TRAIN_CHANNELS_LAST"):
    net = net.to(memory_format=torch.channels_last)
    x = x.contiguous(memory_format=torch.channels_last)
AMP = bool(os.environ.get("TRAIN_AMP"))

def step(timing=None):
    t0 = time.perf_counter()
    with torch.autocast("cuda", dtype=torch.float16, enabled=AMP):
        mask = net(x)
    mask = mask.float()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    total, terms, _ = loss.loss_function(mask, mag, ph, p)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    opt_m.zero_grad(); opt_w.zero_grad()
    total.backward()
    opt_m.step(); opt_w.step()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    if timing is not None:
        timing.append((t1 - t0, t2 - t1, t3 - t2))
    return total.item()

for _ in range(2):
    step()
tm, vals = [], []
for _ in range(5):
    vals.append(step(tm))
u, l, b = (1e3 * sum(t[i] for t in tm) / len(tm) for i in range(3))
print(json.dumps({"workload": f"mask-decoder training step, batch {B} x {AL} s, wav2vec2-base (frozen), HIP U-Net training kernels, HIP LMAC loss fwd/bwd",
                  "ms_per_step": round(u + l + b, 2), "clips_per_s": round(B / (u + l + b) * 1e3, 1),
                  "unet_forward_ms": round(u, 2), "hip_loss_forward_and_grad_ms": round(l, 2),
                  "unet_backward_plus_adam_ms": round(b, 2), "loss_values": [round(v, 4) for v in vals]}))
