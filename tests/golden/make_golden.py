#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE'S OWN MODULES.

Run in the build container only (``python tests/golden/make_golden.py``): it needs
``/root/reference`` and ``transformers``; neither exists on the GPU box, which only ever reads
the committed ``*.npz`` files.  Nothing from the reference is copied: the fixtures hold inputs'
seeds and the reference's numeric outputs.

What is executed unmodified from /root/reference:
  classifier_embedder.py  (zero_mean_unit_var_norm, TorchLogReg)
  audioprocessor.py       (AudioProcessor.compute_stft / compute_invert_stft / extract_features)
  addvisor.py             (ConvBlock, UNet)
  loss_function.py        (LMACLoss.loss_function)

What is redirected, and why: classifier_embedder.py:12-16 loads three artefacts that are private
or remote (a logreg ``.joblib``, the HF feature-extractor config by model NAME, a truncated
XLS-R checkpoint under /mnt/QNAP).  Their loader calls are pointed at seeded synthetic weights
(``addvisor_hip.synthetic``) -- the loaders are I/O, not arithmetic.  ``torchaudio`` is not
installed; audioprocessor.py only needs it for file I/O and for an unused ``MelSpectrogram``
member (audioprocessor.py:38-44), so an empty placeholder module stands in for it.

LMAC_metrics.py cannot be imported as a module (it imports a non-existent ``ADDvisor`` class and loads a
private checkpoint at import), but its six metric functions (LMAC_metrics.py:31-73) are pure: ``metric_functions``
below parses the file with ``ast``, compiles ONLY those six ``FunctionDef`` nodes with ``torch`` / ``F`` / ``eps`` /
``device`` in scope and evaluates them on a seeded table -> ``lmac_metrics.npz``.

Not importable at all, hence no fixtures (parity unpinned, see oracle/__init__.py):
  captum_saliency.py (captum absent), hifigan.py (speechbrain / librosa absent).
"""
import hashlib
import os
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "xai-audio-deepfakes_amd"))
from addvisor_hip import synthetic as syn  # noqa: E402

REF = "/root/reference"
warnings.filterwarnings("ignore")
torch.set_grad_enabled(False)


class _SkLogReg:  # what joblib.load returns in the reference: an object with coef_ / intercept_
    def __init__(self, hidden):
        self.coef_, self.intercept_ = syn.logreg_weights(hidden)


def build_hf(cfg):
    import transformers
    m = transformers.Wav2Vec2Model(transformers.Wav2Vec2Config(**cfg.hf_kwargs()))
    m.load_state_dict(syn.embedder_weights(cfg), strict=True)
    return m.eval()


def import_reference(cfg):
    import joblib
    import transformers
    ta = types.ModuleType("torchaudio")
    tat = types.ModuleType("torchaudio.transforms")

    class _Unused:
        def __init__(self, *a, **k):
            pass
    tat.MelSpectrogram = _Unused
    tat.Resample = _Unused
    ta.transforms = tat
    sys.modules["torchaudio"] = ta
    sys.modules["torchaudio.transforms"] = tat
    joblib.load = lambda path: _SkLogReg(cfg.hidden_size)
    transformers.AutoFeatureExtractor.from_pretrained = classmethod(lambda cls, name, **k: None)
    transformers.Wav2Vec2Model.from_pretrained = classmethod(lambda cls, path, **k: build_hf(cfg))
    sys.path.insert(0, REF)
    import classifier_embedder, audioprocessor, addvisor, loss_function  # noqa: E401
    return classifier_embedder, audioprocessor, addvisor, loss_function


METRIC_DEFS = ("compute_fidelity", "get_score_for_predicted_class", "compute_faithfulness", "compute_AD", "compute_AI",
               "compute_AG")


def metric_functions():
    """The reference's own metric functions (LMAC_metrics.py:31-73), compiled from its source text node by node: the
    module body around them (checkpoint load, dataset walk) is never executed."""
    import ast
    import torch.nn.functional as F
    path = os.path.join(REF, "LMAC_metrics.py")
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    defs = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in METRIC_DEFS]
    assert sorted(d.name for d in defs) == sorted(METRIC_DEFS), [d.name for d in defs]
    scope = {"torch": torch, "F": F, "eps": 1e-10, "device": torch.device("cpu")}      # LMAC_metrics.py:28 eps; audioprocessor.device
    exec(compile(ast.Module(body=defs, type_ignores=[]), path, "exec"), scope)
    return {n: scope[n] for n in METRIC_DEFS}


def metric_table(n=64, seed=61):
    """(p, theta, p_out) triples [n, 1] fp32: random probabilities plus the edge rows the formulas branch on -- ties at
    exactly 0.5, equal scores, 0 / 1 saturation, label flips either way."""
    g = np.random.Generator(np.random.PCG64(seed))
    t = g.uniform(0, 1, size=(n, 3)).astype(np.float32)
    edge = np.array([[0.5, 0.5, 0.5], [0.5, 0.7, 0.2], [0.7, 0.5, 0.5], [0.3, 0.5, 0.9], [0.5, 0.3, 0.5],
                     [1.0, 1.0, 0.0], [0.0, 0.0, 1.0], [1.0, 0.0, 1.0], [0.0, 1.0, 0.0], [0.8, 0.8, 0.8],
                     [0.2, 0.2, 0.2], [0.8, 0.2, 0.6], [0.2, 0.8, 0.4], [0.6, 0.4, 0.6], [0.4, 0.6, 0.4],
                     [0.50000006, 0.49999997, 0.5]], dtype=np.float32)
    t[:len(edge)] = edge
    return torch.from_numpy(t[:, 0:1].copy()), torch.from_numpy(t[:, 1:2].copy()), torch.from_numpy(t[:, 2:3].copy())


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrays.items()})
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def main():
    only = sys.argv[1:]                                  # e.g. `make_golden.py lmac_metrics large`: regenerate these fixtures only
    if only:
        for name in only:
            PARTS[name]()
        return
    tiny_g = syn.tiny_config(stable=False)
    ce, apm, adv, lf = import_reference(tiny_g)

    # ---- 1. normaliser + logreg (classifier_embedder.py:21-63)
    x = syn.make_clips(3, 4000, seed=11)
    feats = torch.from_numpy(np.random.Generator(np.random.PCG64(12)).standard_normal((5, tiny_g.hidden_size)).astype(np.float32))
    lg, pr = ce.TorchLogReg()(feats)
    save("norm_logreg.npz", normed=ce.zero_mean_unit_var_norm(x), logits=lg, probs=pr)

    # ---- 2. STFT / ISTFT (audioprocessor.py:82-131): 1 s clip in full, 4 s and 5 s subsampled
    ap1 = apm.AudioProcessor(audio_length=1)
    w1 = syn.make_clips(1, 16000, seed=21)
    X, mag, ph = ap1.compute_stft(w1)
    Xs, mags, phs = ap1.compute_stft(w1[0, :12000])           # 1-D, shorter than audio_length: padded
    save("stft_1s.npz", X_re=X.real, X_im=X.imag, mag=mag, phase=ph, single_mag=mags,
         istft=ap1.compute_invert_stft(X), shape_single=np.array(Xs.shape))
    for sec in (4, 5):
        ap = apm.AudioProcessor(audio_length=sec)
        w = syn.make_clips(2, sec * 16000 + 777, seed=22)      # longer than audio_length: cropped
        X, mag, ph = ap.compute_stft(w)
        # a spectrogram that is NOT a valid STFT (masked): exercises the overlap-add for real
        g = np.random.Generator(np.random.PCG64(23))
        m = torch.from_numpy(g.uniform(0, 1, size=tuple(mag.shape)).astype(np.float32))
        inv = ap.compute_invert_stft((m * mag) * torch.exp(1j * ph))
        save(f"stft_{sec}s.npz", shape=np.array(X.shape), X_re=X.real[:, ::19, ::7], X_im=X.imag[:, ::19, ::7],
             mag=mag[:, ::19, ::7], phase=ph[:, ::19, ::7], mag_sum=mag.double().sum(), mag_max=mag.max(),
             istft_masked=inv[:, ::13], istft_roundtrip=ap.compute_invert_stft(X)[:, ::13])

    # ---- 3. embedder through the reference's extract_features (audioprocessor.py:69-77), tiny shapes
    for tag, cfg in (("group", tiny_g), ("layer", syn.tiny_config(stable=True))):
        apm.wav2vec2 = build_hf(cfg)
        ap = apm.AudioProcessor(audio_length=1)
        w = syn.make_clips(2, 16000, seed=31)
        save(f"embedder_tiny_{tag}.npz", feats_b2=ap.extract_features(w), feats_b1=ap.extract_features(w[:1]))
    # layer_index == num_hidden_layers (the final LayerNorm of the stable encoder applies)
    cfg9 = syn.tiny_config(stable=True, num_hidden_layers=9)
    apm.wav2vec2 = build_hf(cfg9)
    save("embedder_tiny_layer_depth9.npz", feats_b2=apm.AudioProcessor(1).extract_features(syn.make_clips(2, 16000, seed=31)))

    # ---- 4. full-size wav2vec2-base, one 4 s clip: moments + a corner of hidden_states[9]
    base = syn.base_config()
    apm.wav2vec2 = build_hf(base)
    w = syn.make_clips(1, 64000)
    f = apm.AudioProcessor(audio_length=4).extract_features(w)          # [199, 768]
    save("embedder_base_4s.npz", shape=np.array(f.shape), mean=f.double().mean(), absmax=f.abs().max(),
         std=f.double().std(), corner=f[:8, :16], pooled=f.mean(0))
    apm.wav2vec2 = build_hf(tiny_g)

    # ---- 5. U-Net (addvisor.py:12-84)
    net = adv.UNet()
    net.load_state_dict(syn.unet_weights(), strict=True)
    net.eval()
    g = np.random.Generator(np.random.PCG64(41))
    xa = torch.from_numpy(g.uniform(0, 3, size=(2, 1, 32, 8)).astype(np.float32))
    xb = torch.from_numpy(g.uniform(0, 3, size=(1, 1, 64, 16)).astype(np.float32))
    out_a, out_b = net(xa), net(xb)
    _, mag4, _ = apm.AudioProcessor(audio_length=4).compute_stft(syn.make_clips(1, 64000))
    xin = mag4[:, None, :512, :196]
    full = net(xin)
    net.train()                                                            # last: it updates running stats
    out_train = net(xa)                                                    # batch-stat BN
    net.eval()
    idx = (full > 0.5).numpy().astype(np.uint8)
    save("unet.npz", out_a=out_a, out_b=out_b, out_train=out_train, full_shape=np.array(full.shape),
         full_mean=full.double().mean(), full_sub=full[0, 0, ::17, ::5], full_gt_half=int(idx.sum()),
         full_idx_sha256=np.frombuffer(hashlib.sha256(idx.tobytes()).digest(), dtype=np.uint8),
         full_band=int(((full - 0.5).abs() < 1e-3).sum()))

    # ---- 6. LMAC loss forward (loss_function.py:32-66), module-global 5 s AudioProcessor, tiny embedder
    w = syn.make_clips(2, 80000, seed=51)
    _, mag, ph = lf.audio_processor.compute_stft(w)
    g = np.random.Generator(np.random.PCG64(52))
    xhat = torch.from_numpy(g.uniform(0, 1, size=(2, 1, 513, 249)).astype(np.float32))
    feats = lf.audio_processor.extract_features(w)
    _, p = lf.torch_logreg(feats.mean(dim=1))
    total, losses, wts = lf.LMACLoss().loss_function(xhat, mag, ph, p)
    save("lmac_loss.npz", class_pred=p, total=total, losses=losses, w=wts)

    # ---- 7. the five LMAC metric formulas (LMAC_metrics.py:31-73) on a 64-triple table incl. ties at 0.5
    part_lmac_metrics()
    part_large()
    part_unet5()
    part_xlsr2b()


def part_lmac_metrics():
    fn = metric_functions()
    p, th, po = metric_table()
    per = dict(faithfulness=fn["compute_faithfulness"](p, po), fidelity=fn["compute_fidelity"](th, p).squeeze(1),
               AD=fn["compute_AD"](th, p), AI=fn["compute_AI"](th, p), AG=fn["compute_AG"](th, p))
    save("lmac_metrics.npz", predictions=p, theta_out=th, masked_predictions=po,
         score=fn["get_score_for_predicted_class"](p.squeeze(1)),
         means=np.array([per[k].float().mean().item() for k in ("faithfulness", "fidelity", "AD", "AI", "AG")], dtype=np.float64),
         **per)


def part_large():
    """Full-size wav2vec2-LARGE (layer-norm feature extractor, pre-LN encoder; BASELINE config 5's embedder), one 4 s clip
    through the reference's own extract_features: moments, a corner and the pooled vector of hidden_states[9], plus the
    per-layer moments of hidden_states[0..9] from the HF model the reference calls (SURVEY.md §8(c) fixture (ii))."""
    large = syn.large_config()
    ce, apm, adv, lf = import_reference(large)
    m = build_hf(large)
    apm.wav2vec2 = m
    w = syn.make_clips(1, 64000)
    f = apm.AudioProcessor(audio_length=4).extract_features(w)          # [199, 1024]
    hs = m(ce.zero_mean_unit_var_norm(w), output_hidden_states=True).hidden_states
    save("embedder_large_4s.npz", shape=np.array(f.shape), mean=f.double().mean(), absmax=f.abs().max(), std=f.double().std(),
         corner=f[:8, :16], pooled=f.mean(0),
         layer_mean=np.array([h.double().mean().item() for h in hs[:10]]), layer_std=np.array([h.double().std().item() for h in hs[:10]]),
         layer_absmax=np.array([h.abs().max().item() for h in hs[:10]]))


def part_xlsr2b():
    """The reference's OWN embedder shape (classifier_embedder.py:13-16, 25: XLS-R-2B -- hidden 1920, 16 heads x 120, FFN 7680,
    layer-norm feature extractor, pre-LN encoder -- truncated; `nn.Linear(1920, 1)` head), 10 encoder layers (layers beyond
    hidden_states[9] need no weights), one 4 s clip through the reference's own extract_features: moments, a corner and the
    pooled vector of hidden_states[9], the per-layer moments of hidden_states[0..9] from the HF model the reference calls,
    and the reference TorchLogReg's logit / probability of the pooled vector."""
    cfg = syn.xlsr2b_config(num_hidden_layers=10)
    ce, apm, adv, lf = import_reference(cfg)
    m = build_hf(cfg)
    apm.wav2vec2 = m
    w = syn.make_clips(1, 64000)
    f = apm.AudioProcessor(audio_length=4).extract_features(w)          # [199, 1920]
    hs = m(ce.zero_mean_unit_var_norm(w), output_hidden_states=True).hidden_states
    logit, prob = ce.TorchLogReg()(f.mean(0, keepdim=True))
    save("embedder_xlsr2b_4s.npz", shape=np.array(f.shape), mean=f.double().mean(), absmax=f.abs().max(), std=f.double().std(),
         corner=f[:8, :16], pooled=f.mean(0), logit=logit, prob=prob,
         layer_mean=np.array([h.double().mean().item() for h in hs[:10]]), layer_std=np.array([h.double().std().item() for h in hs[:10]]),
         layer_absmax=np.array([h.abs().max().item() for h in hs[:10]]))


def part_unet5():
    """The reference's default clip length (audio_length = 5: T = 249 frames, 512 x 248 U-Net grid, SURVEY.md §8 sizes in
    brackets), two clips: the reference UNet's mask (addvisor.py:12-84) on the reference STFT magnitude -- subsampled values,
    the `mask > 0.5` count per clip and the SHA-256 of the index set."""
    ce, apm, adv, lf = import_reference(syn.tiny_config(stable=False))
    net = adv.UNet()
    net.load_state_dict(syn.unet_weights(), strict=True)
    net.eval()
    _, mag5, _ = apm.AudioProcessor(audio_length=5).compute_stft(syn.make_clips(2, 80000, seed=71))
    full = net(mag5[:, None, :512, :248])
    idx = (full > 0.5).numpy().astype(np.uint8)
    save("unet_5s.npz", shape=np.array(full.shape), mean=full.double().mean(), sub=full[:, 0, ::17, ::5],
         gt_half=idx.reshape(2, -1).sum(1), idx_sha256=np.frombuffer(hashlib.sha256(idx.tobytes()).digest(), dtype=np.uint8),
         band=int(((full - 0.5).abs() < 1e-3).sum()), closest=float((full - 0.5).abs().min()))


PARTS = {"lmac_metrics": part_lmac_metrics, "large": part_large, "unet5": part_unet5, "xlsr2b": part_xlsr2b}


if __name__ == "__main__":
    main()
