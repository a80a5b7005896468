import sys, os, torch
sys.path.insert(0, "xai-audio-deepfakes_amd")
from addvisor_hip import gemm as G, _lib
_lib.init()
dev = torch.device("cuda:0")
def bench(name, M, K, N):
    g = torch.Generator().manual_seed(0)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    p = G.plan_linear(M, w, torch.zeros(N), device=dev)
    A = torch.randn(M + 2048, K, generator=g).half().to(dev)
    out = torch.empty(M, N, dtype=torch.float16, device=dev)
    other = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=dev)   # 256 MB: flush L2 / MALL between variants
    res = []; ref = None
    for sc in (8, 0, 4, 0, 8, 12, 0, 6):
        p.desc.sc = sc
        for _ in range(3): p.run(A, out_h=out)
        torch.cuda.synchronize()
        if ref is None: ref = out.clone()
        ok = torch.equal(ref, out)
        ts = []
        for _ in range(10):
            other.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); p.run(A, out_h=out); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ms = sorted(ts)[len(ts) // 2]
        res.append(f"sc={sc}: {ms*1e3:7.1f} us {2.0*M*N*K/ms/1e9:6.1f} TF {'ok' if ok else 'BAD'}")
    print(f"{name:10s} M={M} K={K} N={N} | " + " | ".join(res), flush=True)
bench("qkv 3B", 3*12736, 768, 2304)
bench("ffn1 3B", 3*12736, 768, 3072)
bench("ffn2 3B", 3*12736, 3072, 768)
bench("out 3B", 3*12736, 768, 768)
bench("fe_l1", 64*6400, 1536, 512)
