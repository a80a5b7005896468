"""fp32-class attention backward (csrc/attention_bwd_x3.hip; ATT_BWD_F32=1: the fp32-MFMA kernel of attention_bwd_f32.hip) at the IntegratedGradients chunk shape: 64 rows x 16 heads x 199 frames,
head dim 64 (wav2vec2-large); optional argv: B T H heads."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "xai-audio-deepfakes_amd"))
from addvisor_hip import _lib, gemm as G
_lib.init()
if os.environ.get("ATT_BWD_F32"):
    _lib.check(_lib.lib().advh_set_option(b"attention_bwd_mfma_f32", 1), "advh_set_option")
B, T, H, heads = (int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (64, 199, 1024, 16)))
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
qkv = G.split_planes(torch.randn(B * T, 3 * H, generator=g) * 0.7).to(dev)
dctx = G.split_planes(torch.randn(B * T, H, generator=g)).to(dev)
out = torch.zeros(2, B * T, 3 * H, dtype=torch.float16, device=dev)
st = torch.cuda.current_stream().cuda_stream
run = lambda: _lib.check(_lib.lib().advh_attention_bwd_split(qkv.data_ptr(), qkv.stride(0), dctx.data_ptr(), dctx.stride(0), out.data_ptr(), out.stride(0),
                                                             B, T, H, heads, st), "attention_bwd_split")
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record(); e1.synchronize()
us = e0.elapsed_time(e1) / 10 * 1e3
fl = 5 * 2.0 * B * heads * T * T * (H // heads)
print(f"attention_bwd_split B={B} T={T} H={H} heads={heads}: {us:.1f} us, {fl / us / 1e6:.1f} TFLOP/s of the 5 products (fp32 MFMA peak 157; split-arithmetic peak 833)")
if os.environ.get("ATT_BWD_CHECK"):
    a = out.clone()
    _lib.check(_lib.lib().advh_set_option(b"attention_bwd_mfma_f32", 0 if os.environ.get("ATT_BWD_F32") else 1), "advh_set_option")
    run(); torch.cuda.synchronize()
    ja, jb = G.join_planes(a.cpu()), G.join_planes(out.cpu())
    print("max |x3 - f32 kernel| / max|ref| =", float((ja - jb).abs().max() / jb.abs().max()))
