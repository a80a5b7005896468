"""GELU cost in the FFN1 epilogue: the same Linear (M = 38 208, K = 768, N = 3 072) with and without the activation,
operands rotated through 6 buffers so they come from HBM as in the pipeline."""
import sys, torch
sys.path.insert(0, "xai-audio-deepfakes_amd")
from addvisor_hip import gemm as G, _lib
_lib.init()
dev = torch.device("cuda:0")
M, K, N, nbuf = 3 * 12736, 768, 3072, 6
g = torch.Generator().manual_seed(0)
w = torch.randn(N, K, generator=g) / K ** 0.5
A = [torch.randn(M, K, generator=g).half().to(dev) for _ in range(nbuf)]
O = [torch.empty(M, N, dtype=torch.float16, device=dev) for _ in range(nbuf)]
for act in ("none", "gelu", "none", "gelu"):
    p = G.plan_linear(M, w, torch.zeros(N), act=act, device=dev)
    for i in range(nbuf): p.run(A[i], out_h=O[i])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 4 * nbuf
    e0.record()
    for j in range(n): p.run(A[j % nbuf], out_h=O[j % nbuf])
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"ffn1 3B act={act:5s}: {ms*1e3:7.1f} us {2.0*M*N*K/ms/1e9:6.1f} TFLOP/s", flush=True)
