#!/usr/bin/env python3
"""BASELINE config 5 on one GPU's share: IntegratedGradients, 50 steps, wav2vec2-large embedder (layer-norm
feature extractor, pre-LN encoder, hidden_states[9]); 128 clips / 8 GPUs = 16 clips per GPU."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xai-audio-deepfakes_amd"))
import torch
from addvisor_hip import gemm as G, synthetic as syn
from addvisor_hip.attribution import HipAttribution
from addvisor_hip.embedder import HipEmbedder
torch.set_grad_enabled(False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 64
which = sys.argv[3] if len(sys.argv) > 3 else "large"
dev = torch.device("cuda:0")
cfg = syn.large_config() if which == "large" else syn.base_config()
sd = syn.embedder_weights(cfg)
coef, icpt = syn.logreg_weights(cfg.hidden_size)
att = HipAttribution(HipEmbedder(cfg, sd, coef, icpt, dev))
w = syn.make_clips(B, 64000).to(dev)
att.integrated_gradients(w, n_steps=max(1, chunk // B), internal_batch_size=chunk)      # warm-up: builds the one chunk-shaped workspace
torch.cuda.synchronize()
t0 = time.perf_counter()
attr = att.integrated_gradients(w, n_steps=50, internal_batch_size=chunk)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
fwd = att.emb.flops(min(chunk, 50 * B) // B * B if chunk >= B else B, 64000) / (min(chunk, 50 * B) // B * B if chunk >= B else B)
print(json.dumps({"workload": f"IntegratedGradients n_steps=50, wav2vec2-{which}, {B} clips x 4 s, internal batch {chunk}",
                  "clips_per_s": round(B / dt, 3), "seconds": round(dt, 3), "path_points_per_s": round(50 * B / dt, 1),
                  "fwd_gflop_per_point": round(fwd / 1e9, 1), "approx_tflops_fwd_plus_dgrad": round(2 * fwd * 50 * B / dt / 1e12, 1),
                  "finite": bool(torch.isfinite(attr).all().item())}))
