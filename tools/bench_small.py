#!/usr/bin/env python3
"""Micro-benchmarks of the HBM-bound kernels (STFT, masked ISTFT, conv0 front end, attention) at the BASELINE batch."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xai-audio-deepfakes_amd"))
import torch
from addvisor_hip import _lib, ops, synthetic as syn
torch.set_grad_enabled(False)
_lib.init()
dev = torch.device("cuda:0")
B, L = 64, 64000

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

w = syn.make_clips(B, L).to(dev)
mask = torch.rand(B, 512, 196, device=dev)
for fb in (16, 8):
    _lib.check(_lib.lib().advh_set_option(b"stft_frames_per_workgroup", fb), "opt")
    _, mag, ph = ops.stft_forward(w, L, want_complex=False)
    us = timeit(lambda: ops.stft_forward(w, L, want_complex=False))
    print(f"FB={fb:2d} stft_forward (mag+phase)  {us:8.1f} us  {B*(256000+513*199*8)/us/1e6:7.2f} TB/s")
    us = timeit(lambda: ops.istft_masked(mag, ph, mask, L, domain="log1p"))
    print(f"FB={fb:2d} istft_masked in+out       {us:8.1f} us  {2*B*1.47e6/us/1e6:7.2f} TB/s")
from addvisor_hip.embedder import HipEmbedder
cfg = syn.base_config()
emb = HipEmbedder(cfg, syn.embedder_weights(cfg), *syn.logreg_weights(cfg.hidden_size), dev)
ws = emb._workspace(3 * B, L)
x = torch.cat([w, w, w])
lib = _lib.lib(); st = torch.cuda.current_stream().cuda_stream
ln0 = emb.fe_ln[0]
def front():
    lib.advh_w2v2_frontend(x.data_ptr(), x.stride(0), L, 3 * B, L, emb.w0.data_ptr(), None, ln0.g.data_ptr(), ln0.b.data_ptr(), 0, 1,
                           ws["stats"].data_ptr(), ws["norm"].data_ptr(), ws["mr"].data_ptr(), ws["fe"][0].data_ptr(), ws["Ls"][0], ws["P"][0], 512, st)
us = timeit(front)
print(f"w2v2_frontend 3B=192 clips     {us:8.1f} us  {192*12800*512*2/us/1e6:7.2f} TB/s (output bytes)")
qkv = torch.randn(3 * B * 199, 2304, device=dev).half()
def att():
    lib.advh_attention_f16(qkv.data_ptr(), ws["ctx"].data_ptr(), 3 * B, 199, 768, 12, st)
us = timeit(att)
print(f"attention fwd 3B x 12 heads    {us:8.1f} us  {4.0*192*199*199*768/us/1e6:7.1f} TFLOP/s")
