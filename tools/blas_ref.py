"""Reference point: the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS, fp16 in, fp32 accumulate) on the pipeline's shapes."""
import torch
dev = torch.device("cuda:0")
flush = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=dev)
def bench(name, M, K, N, cold):
    a = torch.randn(M, K, device=dev, dtype=torch.float16)
    w = torch.randn(N, K, device=dev, dtype=torch.float16)
    for _ in range(3): torch.matmul(a, w.t())
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        if cold: flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); torch.matmul(a, w.t()); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = sorted(ts)[len(ts) // 2]
    print(f"{name:10s} M={M} K={K} N={N} {'cold' if cold else 'warm'}: {ms*1e3:8.1f} us {2.0*M*N*K/ms/1e9:7.1f} TFLOP/s", flush=True)
for cold in (False, True):
    bench("qkv 3B", 38208, 768, 2304, cold)
    bench("ffn1 3B", 38208, 768, 3072, cold)
    bench("ffn2 3B", 38208, 3072, 768, cold)
    bench("out 3B", 38208, 768, 768, cold)
    bench("fe_l1", 409600, 1536, 512, cold)
    bench("sq 8192", 8192, 8192, 8192, cold)
