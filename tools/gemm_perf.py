import sys, time, torch
sys.path.insert(0, "xai-audio-deepfakes_amd")
from addvisor_hip import gemm as G, _lib
_lib.init()
dev = torch.device("cuda:0")
def bench(name, M, K, N, lda=None):
    g = torch.Generator().manual_seed(0)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    p = G.plan_linear(M, w, torch.zeros(N), lda=lda, device=dev)
    rows = M * (lda or K) // K + 2048 if lda else M + 2048
    A = (torch.randn(M + 2048, lda or K, generator=g)).half().to(dev) if not lda else torch.randn((M*lda + 4*K + 2048*K), generator=g).half().to(dev)
    out = torch.empty(M, N, dtype=torch.float16, device=dev)
    res = []
    ref = None
    for tile in (G.TILE_128x128, G.TILE_256x128_W8, G.TILE_128x256_W8):
        if N < G.TILE_BN[tile] // 2: continue
        p.tile = tile
        out.zero_()
        for _ in range(3): p.run(A, out_h=out)
        torch.cuda.synchronize()
        if ref is None: ref = out.clone()
        ok = torch.equal(ref, out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n): p.run(A, out_h=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        res.append(f"{G.TILE_NAMES[tile]}: {ms*1e3:8.1f} us {2.0*M*N*K/ms/1e9:7.1f} TF {'ok' if ok else 'MISMATCH'}")
    print(f"{name:20s} M={M:7d} K={K:5d} N={N:5d} | " + " | ".join(res), flush=True)
bench("qkv", 12736, 768, 2304)
bench("out_proj", 12736, 768, 768)
bench("ffn1", 12736, 768, 3072)
bench("ffn2", 12736, 3072, 768)
bench("fe_layer1 (k3 s2)", 64 * 6400, 1536, 512, lda=1024)
bench("fe_layer2", 64 * 3200, 1536, 512, lda=1024)
bench("fe_layer5 (k2)", 64 * 400, 1024, 512, lda=1024)
bench("square 4096", 4096, 4096, 4096)
bench("square 8192", 8192, 8192, 8192)
bench("qkv 3B", 3*12736, 768, 2304)
bench("ffn1 3B", 3*12736, 768, 3072)
bench("ffn2 3B", 3*12736, 3072, 768)
bench("out 3B", 3*12736, 768, 768)
bench("unet d4a", 64*66*100, 3456, 256)
bench("unet d3a", 64*130*198, 1728, 128)
