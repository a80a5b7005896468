import sys, torch
sys.path.insert(0, "xai-audio-deepfakes_amd")
from addvisor_hip import gemm as G, _lib
_lib.init()
dev = torch.device("cuda:0")
split = sys.argv[1].startswith("x3:")
tile = {"128": G.TILE_128x128, "256x128": G.TILE_256x128_W8, "128x256": G.TILE_128x256_W8}[sys.argv[1].replace("x3:", "")]
M, K, N = (int(v) for v in sys.argv[2:5])
lda = int(sys.argv[5]) if len(sys.argv) > 5 else None
g = torch.Generator().manual_seed(0)
w = torch.randn(N, K, generator=g) / K ** 0.5
p = G.plan_linear(M, w, None, lda=lda, device=dev, split=split)
p.tile = tile
A = torch.randn((M + 2048) * (lda or K), generator=g).half().to(dev)
out = torch.empty(M, N, dtype=torch.float16, device=dev)
if split:
    A = torch.stack([A, A]).contiguous()
    out = torch.empty(2, M, N, dtype=torch.float16, device=dev)
for _ in range(3):
    p.run(A, out_h=out)
torch.cuda.synchronize()
