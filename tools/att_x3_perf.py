"""Timing of the split (fp32-class) attention at the pipeline's shape: 192 clips x 199 frames, 12 heads of 64."""
import sys, torch
sys.path.insert(0, "xai-audio-deepfakes_amd")
from addvisor_hip import _lib, gemm as G
_lib.init()
dev = torch.device("cuda:0")
B, T, heads, dm = 192, 199, 12, 64
H = heads * dm
qkv = G.split_planes(torch.randn(B * T, 3 * H)).to(dev)
ctx = torch.zeros(2, B * T, H, dtype=torch.float16, device=dev)
st = torch.cuda.current_stream().cuda_stream
def run():
    _lib.check(_lib.lib().advh_attention_split(qkv.data_ptr(), qkv.stride(0), ctx.data_ptr(), ctx.stride(0), B, T, H, heads, st), "att")
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
fl = 4.0 * B * heads * T * T * dm
print(f"split attention B={B} T={T} {heads}x{dm}: {us:.1f} us, {fl / us / 1e6:.1f} TFLOP/s (fp32-class FLOPs)")
