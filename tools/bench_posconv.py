"""Positional-conv layer alone: line-tile kernel vs implicit GEMM for the base (48-channel groups) and large (64) geometry."""
import ctypes, sys, torch
sys.path.insert(0, "xai-audio-deepfakes_amd")
from addvisor_hip import gemm as G, _lib
from addvisor_hip.embedder import PosconvDesc
import numpy as np
_lib.init()
dev = torch.device("cuda:0")
lib = _lib.lib()
for H, B, T in ((768, 192, 199), (1024, 64, 199), (1024, 192, 199)):
    Gp, K = 16, 128
    Cg, cc = H // Gp, H // Gp // 8
    M = B * T
    g = torch.Generator().manual_seed(0)
    w = torch.randn(Gp, Cg, K * Cg, generator=g) / (K * Cg) ** 0.5
    bias = torch.zeros(H)
    h = torch.randn(M, H, generator=g).to(dev)
    xg = torch.zeros(Gp, B, T + K, Cg, dtype=torch.float16, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.advh_posconv_gather(h.data_ptr(), xg.data_ptr(), B, T, H, Gp, K, K // 2, None, st), "gather")
    plan = G.GemmPlan(M=M, N=Cg, w2=w, ktab=np.arange(K * cc, dtype=np.int64),
                      sources=[G.Source((T + K) * cc, 0, cc, 0, sZ=B * (T + K) * cc)], Hg=1, Wg=T, window=(0, 1, 0, T), halo_zero=False,
                      out=(T * H, 0, H, 0), n_div=G.round_up(Cg, 4), o_sZ=Cg, nz=Gp, bias=bias, bias_sZ=Cg, act="gelu", device=dev)
    wt = w.view(Gp, Cg, K * Cg // 32, 32).permute(0, 2, 1, 3).contiguous().half().to(dev)
    bd = bias.to(dev)
    out = torch.empty_like(h)
    d = PosconvDesc()
    d.xg, d.W, d.bias, d.resid, d.out = xg.data_ptr(), wt.data_ptr(), bd.data_ptr(), h.data_ptr(), out.data_ptr()
    d.B, d.T, d.H, d.G, d.K = B, T, H, Gp, K
    def run_gemm(): plan.run(xg, out_f=out, resid=h)
    def run_tile(): _lib.check(lib.advh_posconv_tile_f16(ctypes.byref(d), st), "tile")
    res = []
    for name, fn in (("gemm", run_gemm), ("tile", run_tile)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        res.append(f"{name} {ms*1e3:7.1f} us {2.0*M*Cg*K*Cg*Gp/ms/1e9:6.1f} TFLOP/s")
    print(f"H={H} B={B} T={T}: " + " | ".join(res), flush=True)
