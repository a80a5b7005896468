"""Does the activation row pitch matter for the K = 3072 Linear (FFN2)?  Same GEMM with the A rows 3072, 3072 + 64 and 3072 + 128 halfs apart."""
import sys, time, torch
sys.path.insert(0, "xai-audio-deepfakes_amd")
from addvisor_hip import gemm as G, _lib
_lib.init()
dev = torch.device("cuda:0")
M, K, N = 38208, 3072, 768
g = torch.Generator().manual_seed(0)
w = torch.randn(N, K, generator=g) / K ** 0.5
bias = torch.randn(N, generator=g)
for split in (True, False):
    for pad in (0, 64, 128, 192):
        lda = K + pad
        a = torch.randn(M + 1024, lda, generator=g)
        p = G.plan_linear(M, w, bias, device=dev, split=split, lda=lda)
        A = (G.split_planes(a) if split else a.half()).to(dev)
        out = torch.zeros(M, N, dtype=torch.float32, device=dev)
        kw = dict(out_f=out, resid=out)
        t0 = time.time()
        while time.time() - t0 < 1.0:
            for _ in range(20):
                p.run(A, **kw)
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            p.run(A, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 40
        print(f"{'x3 ' if split else 'f16'} lda = K + {pad:3d}: {ms * 1e3:7.1f} us  {2.0 * M * N * K / ms / 1e9:6.1f} TFLOP/s", flush=True)
