"""fp16 vs fp32-class (split, 3 MFMAs per product) instance of the 128x128 implicit-GEMM tile on the pipeline's plain shapes,
each with the epilogue the embedder gives it (h: fp16-side output; g: GELU; r: fp32 residual stream), after a 1 s warm-up
(the clock settles only after ~1 s of load: a cold measurement reads 20-25 % low)."""
import sys, time, torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "xai-audio-deepfakes_amd"))
from addvisor_hip import gemm as G, _lib
_lib.init()
dev = torch.device("cuda:0")
WARM_S = 1.0


def bench(name, M, K, N, epi="h"):
    g = torch.Generator().manual_seed(0)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    a = torch.randn(M + 1024, K, generator=g)
    bias = torch.randn(N, generator=g)
    res = []
    ref = (a[:M].double() @ w.double().T + bias.double())
    if epi == "g":
        ref = torch.nn.functional.gelu(ref)
    for split, tile in ((False, G.TILE_128x128), (True, G.TILE_128x128)):
        p = G.plan_linear(M, w, bias, device=dev, split=split, act="gelu" if epi == "g" else "none")
        p.tile = tile
        A = (G.split_planes(a) if split else a.half()).to(dev)
        if epi == "r":
            out = torch.zeros(M, N, dtype=torch.float32, device=dev)
            kw = dict(out_f=out, resid=out)
        else:
            out = torch.empty(((2,) if split else ()) + (M, N), dtype=torch.float16, device=dev)
            kw = dict(out_h=out)
        t0 = time.time()
        while time.time() - t0 < WARM_S:
            for _ in range(20):
                p.run(A, **kw)
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 40
        e0.record()
        for _ in range(n):
            p.run(A, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        if epi == "r":
            out.zero_()
            p.run(A, **kw)
            v = out
        else:
            v = G.join_planes(out) if split else out.float()
        err = ((v.double().cpu() - ref).abs().max() / ref.abs().max()).item()
        res.append(f"{('x3 ' + G.TILE_NAMES[tile]) if split else 'f16'}: {ms * 1e3:8.1f} us {2.0 * M * N * K / ms / 1e9:7.1f} TF err {err:.1e}")
    print(f"{name:12s} M={M:7d} K={K:5d} N={N:5d} | " + " | ".join(res), flush=True)


bench("qkv 3B", 3 * 12736, 768, 2304)
bench("out 3B r", 3 * 12736, 768, 768, "r")
bench("ffn1 3B g", 3 * 12736, 768, 3072, "g")
bench("ffn2 3B r", 3 * 12736, 3072, 768, "r")
bench("fe-like g", 64 * 3200, 1536, 512, "g")
bench("sq 8192", 8192, 8192, 8192)
