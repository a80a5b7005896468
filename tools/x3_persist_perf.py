"""gemm_x3_kernel: one tile per workgroup (hardware dispatcher) against a persistent launch (advh_set_option("x3_persist_slots", N):
N workgroups walk the tile list), pipeline shapes, warm clock.  Results must be bit-identical."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "xai-audio-deepfakes_amd"))
from addvisor_hip import gemm as G, _lib
_lib.init()
dev = torch.device("cuda:0")


def run(name, M, K, N, epi):
    g = torch.Generator().manual_seed(0)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    a = G.split_planes(torch.randn(M + 1024, K, generator=g)).to(dev)
    p = G.plan_linear(M, w, torch.randn(N, generator=g), device=dev, split=True, act="gelu" if epi == "g" else "none")
    res, outs = [], []
    for slots in (0, 512, 768, 1024):
        _lib.check(_lib.lib().advh_set_option(b"x3_persist_slots", slots), "set_option")
        if epi == "r":
            out = torch.zeros(M, N, dtype=torch.float32, device=dev); kw = dict(out_f=out, resid=torch.ones(M, N, device=dev))
        else:
            out = torch.empty(2, M, N, dtype=torch.float16, device=dev); kw = dict(out_h=out)
        t0 = time.time()
        while time.time() - t0 < 0.7:
            for _ in range(20):
                p.run(a, **kw)
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            p.run(a, **kw)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 40
        outs.append(out.clone())
        res.append(f"{'dispatcher' if not slots else str(slots) + ' persistent'}: {ms * 1e3:7.1f} us {2.0 * M * N * K / ms / 1e9:6.1f} TF")
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    print(f"{name:10s} M={M:7d} K={K:5d} N={N:5d} | " + " | ".join(res) + (" | bit-identical" if same else " | RESULTS DIFFER"), flush=True)
    _lib.check(_lib.lib().advh_set_option(b"x3_persist_slots", 0), "set_option")


run("qkv 3B", 3 * 12736, 768, 2304, "h")
run("out 3B r", 3 * 12736, 768, 768, "r")
run("ffn1 3B g", 3 * 12736, 768, 3072, "g")
run("ffn2 3B r", 3 * 12736, 3072, 768, "r")
run("fe-like g", 64 * 3200, 1536, 512, "g")
