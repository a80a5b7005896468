"""Split-arithmetic transposeless wgrad tile kernel (csrc/conv_wgrad.hip) on one channel-slice pair of a 64-clip batch; argv: CI CO H W.
Times advh_conv_wgrad2d_split (tile kernel + partial reduction) with HIP events."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "xai-audio-deepfakes_amd"))
from addvisor_hip import _lib, gemm as G
from addvisor_hip.unet_train import Wgrad2dDesc
_lib.init()
CI, CO, H, W = (int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (64, 64, 128, 196)))
B, dev = 64, torch.device("cuda:0")
x = G.FMap(B, H, W, CI, 1, 1, split=True).alloc(dev)
z = G.FMap(B, H, W, CO, 1, 1, split=True).alloc(dev)
g = torch.Generator(device="cpu").manual_seed(0)
x.t[:, :, 1:1 + H, 1:1 + W] = G.split_planes(torch.randn(B, H, W, CI, generator=g)).to(dev)
z.t[:, :, 1:1 + H, 1:1 + W] = G.split_planes(torch.randn(B, H, W, CO, generator=g)).to(dev)
lib = _lib.lib()
parts = lib.advh_conv_wgrad2d_split_parts(CI, CO, B, H, W)
part = torch.empty(parts * 9 * CI * CO, dtype=torch.float32, device=dev)
dw = torch.empty(9, CO, CI, dtype=torch.float32, device=dev)
d = Wgrad2dDesc(B=B, H=H, W_=W, PHx=1, PWx=1, PHz=1, PWz=1)
d.X, d.DZ, d.partial = x.t.data_ptr(), z.t.data_ptr(), part.data_ptr()
st = torch.cuda.current_stream().cuda_stream
run = lambda: _lib.check(lib.advh_conv_wgrad2d_split(C.byref(d), CI, CO, CI, 0, CO, 0, x.t.stride(0), z.t.stride(0), dw.data_ptr(), st), "wgrad")
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record(); e1.synchronize()
us = e0.elapsed_time(e1) / 10 * 1e3
fl = 2.0 * B * H * W * 9 * CI * CO
ref = torch.einsum("bhwo,bhwi->oi", G.join_planes(z.t[:, :, 1:1 + H, 1:1 + W].cpu()).double(), G.join_planes(x.t[:, :, 1:1 + H, 1:1 + W].cpu()).double()) if B * H * W <= 2e6 else None
err = float((dw[4].cpu().double() - ref).abs().max() / ref.abs().max()) if ref is not None else float("nan")
print(f"wgrad2d_split CI={CI} CO={CO} {B}x{H}x{W}: {us:.1f} us, {fl / us / 1e6:.1f} TFLOP/s useful (x3 peak 833), centre-tap err {err:.1e}")
