"""fp16 vs fp32-class (split, 3 MFMAs per product) instance of the 128x128 implicit-GEMM tile on the pipeline's plain shapes."""
import sys, torch
sys.path.insert(0, "xai-audio-deepfakes_amd")
from addvisor_hip import gemm as G, _lib
_lib.init()
dev = torch.device("cuda:0")


def bench(name, M, K, N):
    g = torch.Generator().manual_seed(0)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    a = torch.randn(M + 1024, K, generator=g)
    res = []
    ref = None
    for split, tile in ((False, G.TILE_128x128), (True, G.TILE_128x128)):
        p = G.plan_linear(M, w, torch.zeros(N), device=dev, split=split)
        p.tile = tile
        A = (G.split_planes(a) if split else a.half()).to(dev)
        out = torch.empty(((2,) if split else ()) + (M, N), dtype=torch.float16, device=dev)
        for _ in range(3):
            p.run(A, out_h=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for _ in range(n):
            p.run(A, out_h=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        chk = ""
        if split:
            v = G.join_planes(out)
            if ref is None:
                ref = v
            else:
                chk = f" d={((v - ref).abs().max() / ref.abs().max()).item():.1e}"
        res.append(f"{('x3 ' + G.TILE_NAMES[tile]) if split else 'f16'}: {ms * 1e3:8.1f} us {2.0 * M * N * K / ms / 1e9:7.1f} TF{chk}")
    print(f"{name:12s} M={M:7d} K={K:5d} N={N:5d} | " + " | ".join(res), flush=True)


bench("qkv 3B", 3 * 12736, 768, 2304)
bench("out 3B", 3 * 12736, 768, 768)
bench("ffn1 3B", 3 * 12736, 768, 3072)
bench("ffn2 3B", 3 * 12736, 3072, 768)
bench("fe-like", 64 * 3200, 1536, 512)
bench("sq 8192", 8192, 8192, 8192)
