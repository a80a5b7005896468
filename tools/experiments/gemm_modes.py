"""Which epilogue options the GEMM launches of one fp32-class explanation use (to decide which forms deserve their own code)."""
import collections, sys
sys.path.insert(0, "xai-audio-deepfakes_amd")
import torch
from addvisor_hip import gemm as G, pipeline as P, synthetic as syn
dev = torch.device("cuda:0")
seen = collections.Counter()
orig = G.GemmPlan.run


def run(self, A0, A1=None, **kw):
    d = self.desc
    key = (G.TILE_NAMES.get(self.tile, self.tile), "plain" if d.plain else "rows", "plain_out" if d.plain_out else "-", "wide" if d.wide else "narrow",
           ("none", "gelu", "leaky")[d.act], "bias" if self.bias is not None else "nobias",
           "+".join(k for k in ("out_h", "out_f", "out_h2", "resid", "out_pre", "dact_src") if kw.get(k) is not None),
           f"ph_r={d.ph_r}", "oneblk" if d.n_div >= d.N else f"n_div={d.n_div}", f"n_sub={d.n_sub}", "halo_zero" if d.halo_zero else "-",
           f"M={d.M} N={d.N} K={d.Ktot} nz={d.nz}")
    seen[key] += 1
    return orig(self, A0, A1, **kw)


G.GemmPlan.run = run
cfg = syn.base_config() if hasattr(syn, "base_config") else syn.tiny_config(False)
emb_sd, unet_sd = syn.embedder_weights(cfg), syn.unet_weights()
coef, icpt = syn.logreg_weights(cfg.hidden_size)
pipe = P.ExplainPipeline(cfg, emb_sd, coef, icpt, unet_sd, dev, audio_length=4, precision="f32")
w = syn.make_clips(8, 64000, seed=1).to(dev)
pipe.explain(w)
seen.clear()
pipe.explain(w)
for k, n in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(n, *k)
