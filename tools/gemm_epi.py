"""Epilogue cost of the 128x128 GEMM on the transformer shapes: fp16 output vs fp32 residual in + fp32 out (+ fp16 copy),
with the operands rotated through `nbuf` buffers so the residual comes from HBM as in the pipeline."""
import sys, torch
sys.path.insert(0, "xai-audio-deepfakes_amd")
from addvisor_hip import gemm as G, _lib
_lib.init()
dev = torch.device("cuda:0")

def bench(name, M, K, N, nbuf=6):
    g = torch.Generator().manual_seed(0)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    p = G.plan_linear(M, w, torch.zeros(N), device=dev)
    A = [torch.randn(M + 2048, K, generator=g).half().to(dev) for _ in range(nbuf)]
    R = [torch.randn(M, N, device=dev) for _ in range(nbuf)]
    OF = [torch.empty(M, N, device=dev) for _ in range(nbuf)]
    OH = [torch.empty(M, N, dtype=torch.float16, device=dev) for _ in range(nbuf)]
    res = []
    for label, kw in (("h", lambda i: dict(out_h=OH[i])), ("f", lambda i: dict(out_f=OF[i])),
                      ("f+resid", lambda i: dict(out_f=OF[i], resid=R[i])), ("f+h+resid", lambda i: dict(out_f=OF[i], out_h=OH[i], resid=R[i])),
                      ("f+resid inplace", lambda i: dict(out_f=R[i], resid=R[i]))):
        for i in range(nbuf): p.run(A[i], **kw(i))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 4 * nbuf
        e0.record()
        for j in range(n): p.run(A[j % nbuf], **kw(j % nbuf))
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        res.append(f"{label}: {ms*1e3:7.1f} us {2.0*M*N*K/ms/1e9:6.1f} TF")
    print(f"{name:10s} M={M:7d} K={K:5d} N={N:5d} | " + " | ".join(res), flush=True)

bench("out 3B", 3 * 12736, 768, 768)
bench("ffn2 3B", 3 * 12736, 3072, 768)
bench("out 3B x1", 3 * 12736, 768, 768, nbuf=1)
