"""fp16 tiles on the transformer shapes with a warm clock (1 s of load before timing): does a larger tile beat 128x128?"""
import sys, time, torch
sys.path.insert(0, "xai-audio-deepfakes_amd")
from addvisor_hip import gemm as G, _lib
_lib.init()
dev = torch.device("cuda:0")
TILES = [G.TILE_128x128, G.TILE_256x128_W8, G.TILE_128x256_W8]    # the ring / pipelined / persistent variants this script compared in round 2 (profiles/r02_f16_tiles.txt) were removed in round 3


def bench(name, M, K, N):
    g = torch.Generator().manual_seed(0)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    a = torch.randn(M + 1024, K, generator=g).half().to(dev)
    bias = torch.randn(N, generator=g)
    out = torch.empty(M, N, dtype=torch.float16, device=dev)
    res = []
    ref = None
    for tile in TILES:
        try:
            p = G.plan_linear(M, w, bias, device=dev)
            p.tile = tile
            p.run(a, out_h=out)
            torch.cuda.synchronize()
        except Exception as e:                       # a tile that refuses the descriptor
            res.append(f"{G.TILE_NAMES[tile]}: n/a")
            continue
        if ref is None:
            ref = out.clone()
        same = torch.equal(out, ref)
        t0 = time.time()
        while time.time() - t0 < 1.0:
            for _ in range(20):
                p.run(a, out_h=out)
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            p.run(a, out_h=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 40
        res.append(f"{G.TILE_NAMES[tile]}: {2.0 * M * N * K / ms / 1e9:5.0f}{'' if same else ' (differs)'}")
    print(f"{name:8s} M={M} K={K} N={N} | " + " | ".join(res), flush=True)


bench("qkv", 38208, 768, 2304)
bench("out", 38208, 768, 768)
bench("ffn1", 38208, 768, 3072)
bench("ffn2", 38208, 3072, 768)
bench("large1", 12736, 1024, 4096)
bench("large2", 12736, 4096, 1024)
