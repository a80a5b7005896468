"""Which epilogue options the GEMM launches of one HiFi-GAN V1 decode use."""
import collections, sys
sys.path.insert(0, "xai-audio-deepfakes_amd")
import numpy as np, torch
from addvisor_hip import gemm as G, synthetic as syn
from addvisor_hip.hifigan import HipHifigan
dev = torch.device("cuda:0")
seen = collections.Counter()
orig = G.GemmPlan.run


def run(self, A0, A1=None, **kw):
    d = self.desc
    key = (G.TILE_NAMES.get(self.tile, self.tile), "plain" if d.plain else "rows", "wide" if d.wide else "narrow", ("none", "gelu", "leaky")[d.act],
           "bias" if self.bias is not None else "nobias", "+".join(k for k in ("out_h", "out_f", "out_h2", "resid", "out_pre") if kw.get(k) is not None),
           f"ph_r={d.ph_r}", "oneblk" if d.n_div >= d.N else f"n_div={d.n_div}", "halo_zero" if d.halo_zero else "-", f"M={d.M} N={d.N} K={d.Ktot} nz={d.nz}")
    seen[key] += 1
    return orig(self, A0, A1, **kw)


G.GemmPlan.run = run
cfg = syn.HifiganConfig()
net = HipHifigan(cfg, syn.hifigan_weights(cfg), dev)
mel = torch.from_numpy(np.random.default_rng(0).normal(-4, 2, size=(16, 80, 251)).astype(np.float32)).to(dev)
net.decode_batch(mel)
seen.clear()
net.decode_batch(mel)
for k, n in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(n, *k)
