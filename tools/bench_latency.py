#!/usr/bin/env python3
"""Latency of one explanation step at small batch sizes: eager launch loop vs HIP-graph replay (wav2vec2-base, 4 s)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xai-audio-deepfakes_amd"))
import torch
from addvisor_hip import pipeline as P, synthetic as syn
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
cfg = syn.base_config()
pipe = P.ExplainPipeline(cfg, syn.embedder_weights(cfg), *syn.logreg_weights(cfg.hidden_size), syn.unet_weights(), dev, audio_length=4)
for B in (1, 4, 16):
    w = syn.make_clips(B, 64000, seed=B).to(dev)
    pipe.capture(B)
    def run(fn, n=30):
        for _ in range(3): fn(w)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn(w)
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    e, g = run(pipe.explain), run(pipe.explain_graphed)
    print(json.dumps({"batch": B, "eager_ms": round(e, 3), "graph_ms": round(g, 3), "explanations_per_s_graph": round(B / g * 1e3, 1)}), flush=True)
