"""Does the 128x128 GEMM's time follow the tile count or its quantisation into waves of 1024 concurrent workgroups?"""
import sys, torch
sys.path.insert(0, "xai-audio-deepfakes_amd")
from addvisor_hip import gemm as G, _lib
_lib.init()
dev = torch.device("cuda:0")
K, N = 768, 2304
g = torch.Generator().manual_seed(0)
w = torch.randn(N, K, generator=g) / K ** 0.5
flush = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=dev)
for mt in (256, 284, 290, 299, 320, 341, 342, 360, 398, 399):
    M = mt * 128
    p = G.plan_linear(M, w, torch.zeros(N), device=dev)
    A = torch.randn(M + 1024, K, generator=g).half().to(dev)
    out = torch.empty(M, N, dtype=torch.float16, device=dev)
    for _ in range(3): p.run(A, out_h=out)
    ts = []
    for _ in range(9):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); p.run(A, out_h=out); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = sorted(ts)[4]
    tiles = mt * 18
    print(f"M-tiles {mt:4d} tiles {tiles:5d} = {tiles/1024:5.2f} waves: {ms*1e3:7.1f} us  {ms*1e3/tiles*1024:6.2f} us per 1024 tiles  {2.0*M*N*K/ms/1e9:6.1f} TF", flush=True)
