"""Pin the CPU oracle (oracle/) against the golden vectors produced by the reference's own
modules (tests/golden/make_golden.py).  CPU only."""
import hashlib

import numpy as np
import torch

from addvisor_hip import synthetic as syn
from oracle import lmac_ref, signal_ref, unet_ref, wav2vec2_ref

torch.set_grad_enabled(False)


def T(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, atol, rtol=0.0):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs().max().item()
    assert err <= atol + rtol * b.abs().max().item(), err


def test_norm_and_logreg(golden):
    g = golden("norm_logreg.npz")
    x = syn.make_clips(3, 4000, seed=11)
    close(signal_ref.zero_mean_unit_var_norm(x), g["normed"], 1e-6)
    cfg = syn.tiny_config()
    feats = T(np.random.Generator(np.random.PCG64(12)).standard_normal((5, cfg.hidden_size)).astype(np.float32))
    coef, icpt = syn.logreg_weights(cfg.hidden_size)
    lg, pr = wav2vec2_ref.logreg(feats, coef, icpt)
    close(lg, g["logits"], 1e-6)
    close(pr, g["probs"], 1e-6)


def test_stft_1s_full(golden):
    g = golden("stft_1s.npz")
    w = syn.make_clips(1, 16000, seed=21)
    X, mag, ph = signal_ref.compute_stft(w, audio_length=1)
    close(X.real, g["X_re"], 1e-5)
    close(X.imag, g["X_im"], 1e-5)
    close(mag, g["mag"], 1e-5)
    close(ph, g["phase"], 1e-5)
    close(signal_ref.compute_invert_stft(X, audio_length=1), g["istft"], 1e-6)
    Xs, ms, _ = signal_ref.compute_stft(w[0, :12000], audio_length=1)
    assert tuple(Xs.shape) == tuple(g["shape_single"])
    close(ms, g["single_mag"], 1e-5)


def test_stft_full_length(golden):
    for sec in (4, 5):
        g = golden(f"stft_{sec}s.npz")
        w = syn.make_clips(2, sec * 16000 + 777, seed=22)
        X, mag, ph = signal_ref.compute_stft(w, audio_length=sec)
        assert tuple(X.shape) == tuple(g["shape"])
        close(X.real[:, ::19, ::7], g["X_re"], 1e-5)
        close(mag[:, ::19, ::7], g["mag"], 1e-5)
        close(ph[:, ::19, ::7], g["phase"], 1e-5)
        assert abs(mag.double().sum().item() - float(g["mag_sum"])) < 1e-6 * float(g["mag_sum"])
        m = T(np.random.Generator(np.random.PCG64(23)).uniform(0, 1, size=tuple(mag.shape)).astype(np.float32))
        rel, _ = signal_ref.apply_mask(m, mag, ph, "linear")
        close(signal_ref.compute_invert_stft(rel, audio_length=sec)[:, ::13], g["istft_masked"], 2e-6)
        close(signal_ref.compute_invert_stft(X, audio_length=sec)[:, ::13], g["istft_roundtrip"], 2e-6)


def test_embedder_tiny(golden):
    w = syn.make_clips(2, 16000, seed=31)
    for tag, cfg in (("group", syn.tiny_config(False)), ("layer", syn.tiny_config(True))):
        g = golden(f"embedder_tiny_{tag}.npz")
        sd = syn.embedder_weights(cfg)
        f2 = wav2vec2_ref.extract_features(w, sd, cfg)
        f1 = wav2vec2_ref.extract_features(w[:1], sd, cfg)
        assert f1.dim() == 2 and f2.dim() == 3          # squeeze(0)
        close(f2, g["feats_b2"], 2e-5)
        close(f1, g["feats_b1"], 2e-5)
    cfg9 = syn.tiny_config(True, num_hidden_layers=9)
    g = golden("embedder_tiny_layer_depth9.npz")
    close(wav2vec2_ref.extract_features(w, syn.embedder_weights(cfg9), cfg9), g["feats_b2"], 2e-5)


def test_embedder_base_full(golden):
    g = golden("embedder_base_4s.npz")
    cfg = syn.base_config()
    f = wav2vec2_ref.extract_features(syn.make_clips(1, 64000), syn.embedder_weights(cfg), cfg)
    assert tuple(f.shape) == tuple(g["shape"])
    close(f[:8, :16], g["corner"], 5e-5)
    close(f.mean(0), g["pooled"], 2e-5)
    assert abs(f.abs().max().item() - float(g["absmax"])) < 1e-4


def test_embedder_large_full(golden):
    """wav2vec2-large (layer-norm feature extractor, pre-LN encoder: BASELINE config 5's embedder), one 4 s clip, against
    the reference's own extract_features and the HF model's per-layer moments (tests/golden/make_golden.py large)."""
    g = golden("embedder_large_4s.npz")
    cfg = syn.large_config()
    sd = syn.embedder_weights(cfg)
    w = syn.make_clips(1, 64000)
    hs = wav2vec2_ref.hidden_states(wav2vec2_ref.zero_mean_unit_var_norm(w), sd, cfg, upto=9)
    f = hs[9][0]
    assert tuple(f.shape) == tuple(g["shape"])
    close(f[:8, :16], g["corner"], 1e-4)
    close(f.mean(0), g["pooled"], 5e-5)
    for i in range(10):
        assert abs(hs[i].double().mean().item() - g["layer_mean"][i]) < 1e-5 + 1e-4 * abs(g["layer_mean"][i]), i
        assert abs(hs[i].double().std().item() - g["layer_std"][i]) < 1e-4 * g["layer_std"][i], i


def test_embedder_xlsr2b_full(golden):
    """The reference's OWN embedder shape (classifier_embedder.py:13-16, 25: XLS-R-2B, hidden 1920, 16 heads x 120, FFN 7680),
    10 encoder layers, one 4 s clip: the oracle against the reference's extract_features / TorchLogReg and the HF model's
    per-layer moments (tests/golden/make_golden.py xlsr2b)."""
    g = golden("embedder_xlsr2b_4s.npz")
    cfg = syn.xlsr2b_config(num_hidden_layers=10)
    sd = syn.embedder_weights(cfg)
    w = syn.make_clips(1, 64000)
    hs = wav2vec2_ref.hidden_states(wav2vec2_ref.zero_mean_unit_var_norm(w), sd, cfg, upto=9)
    f = hs[9][0]
    assert tuple(f.shape) == tuple(g["shape"]) == (199, 1920)
    amax = float(g["absmax"])
    close(f[:8, :16], g["corner"], 2e-5 * amax + 1e-5)
    close(f.mean(0), g["pooled"], 1e-5 * amax + 1e-5)
    coef, icpt = syn.logreg_weights(cfg.hidden_size)
    logit, prob = wav2vec2_ref.logreg(f.mean(0, keepdim=True), coef, icpt)
    close(logit, g["logit"], 1e-4)
    close(prob, g["prob"], 1e-5)
    for i in range(10):
        assert abs(hs[i].double().mean().item() - g["layer_mean"][i]) < 1e-5 + 1e-4 * abs(g["layer_mean"][i]), i
        assert abs(hs[i].double().std().item() - g["layer_std"][i]) < 1e-4 * g["layer_std"][i], i


def test_unet(golden):
    g = golden("unet.npz")
    sd = syn.unet_weights()
    r = np.random.Generator(np.random.PCG64(41))
    xa = T(r.uniform(0, 3, size=(2, 1, 32, 8)).astype(np.float32))
    xb = T(r.uniform(0, 3, size=(1, 1, 64, 16)).astype(np.float32))
    close(unet_ref.unet_forward(xa, sd), g["out_a"], 2e-6)
    close(unet_ref.unet_forward(xb, sd), g["out_b"], 2e-6)
    close(unet_ref.unet_forward(xa, sd, bn_batch=True), g["out_train"], 5e-6)
    _, mag, _ = signal_ref.compute_stft(syn.make_clips(1, 64000), audio_length=4)
    full = unet_ref.unet_forward(unet_ref.crop_for_unet(mag), sd)
    assert tuple(full.shape) == tuple(g["full_shape"])
    close(full[0, 0, ::17, ::5], g["full_sub"], 5e-6)
    idx = (full > 0.5).numpy().astype(np.uint8)
    assert int(idx.sum()) == int(g["full_gt_half"])
    assert hashlib.sha256(idx.tobytes()).digest() == g["full_idx_sha256"].tobytes()   # bit-exact mask indices


def test_unet_5s_golden(golden):
    """Reference default length (audio_length = 5 -> 512 x 248 grid), two clips: oracle mask and `mask > 0.5` index set vs the
    reference's own (tests/golden/make_golden.py unet5)."""
    g = golden("unet_5s.npz")
    sd = syn.unet_weights()
    _, mag, _ = signal_ref.compute_stft(syn.make_clips(2, 80000, seed=71), audio_length=5)
    full = unet_ref.unet_forward(unet_ref.crop_for_unet(mag), sd)
    assert tuple(full.shape) == tuple(g["shape"])
    close(full[:, 0, ::17, ::5], g["sub"], 1e-5)
    idx = (full > 0.5).numpy().astype(np.uint8)
    assert idx.reshape(2, -1).sum(1).tolist() == g["gt_half"].tolist()
    assert hashlib.sha256(idx.tobytes()).digest() == g["idx_sha256"].tobytes()


def test_lmac_loss(golden):
    g = golden("lmac_loss.npz")
    cfg = syn.tiny_config()
    sd = syn.embedder_weights(cfg)
    coef, icpt = syn.logreg_weights(cfg.hidden_size)
    w = syn.make_clips(2, 80000, seed=51)
    _, mag, ph = signal_ref.compute_stft(w, audio_length=5)
    xhat = T(np.random.Generator(np.random.PCG64(52)).uniform(0, 1, size=(2, 1, 513, 249)).astype(np.float32))
    _, p = wav2vec2_ref.classify(w, sd, cfg, coef, icpt)
    close(p, g["class_pred"], 1e-6)
    total, losses, wts = lmac_ref.lmac_loss(xhat, mag, ph, p, [3.0, 0.5, 3.0], sd, cfg, coef, icpt, audio_length=5)
    close(losses, g["losses"], 2e-6)
    close(wts, g["w"], 1e-6)
    close(total, g["total"], 1e-5)


def test_lmac_metric_table():
    """Known-answer table for the five metrics (LMAC_metrics.py:31-73), incl. ties at 0.5."""
    p = T(np.array([[0.9], [0.2], [0.5], [0.5], [0.7]], dtype=np.float32))
    th = T(np.array([[0.8], [0.1], [0.6], [0.5], [0.3]], dtype=np.float32))
    po = T(np.array([[0.4], [0.6], [0.5], [0.2], [0.9]], dtype=np.float32))
    assert lmac_ref.compute_fidelity(th, p).view(-1).tolist() == [1, 1, 0, 1, 0]
    f = lmac_ref.compute_faithfulness(p, po)
    np.testing.assert_allclose(f.numpy(), [0.5, 0.4, 0.0, 0.0, -0.2], atol=1e-6)   # sign(0) = 0 at p = 0.5
    ad = lmac_ref.compute_AD(th, p)
    # pc = [.9,.8,.5,.5,.7]; oc = [.8,.9,.6,.5,.7]
    np.testing.assert_allclose(ad.numpy(), [100 * 0.1 / 0.9, 0, 0, 0, 0], atol=1e-4)
    assert lmac_ref.compute_AI(th, p).tolist() == [0, 100, 100, 0, 0]
    ag = lmac_ref.compute_AG(th, p)
    np.testing.assert_allclose(ag.numpy(), [0, 100 * 0.1 / 0.2, 100 * 0.1 / 0.5, 0, 0], atol=1e-3)


def test_lmac_metrics_golden(golden):
    """The oracle's five metric functions against the REFERENCE's own (LMAC_metrics.py:31-73 compiled from its source by
    tests/golden/make_golden.py): 64 (p, theta, p_out) triples incl. ties at exactly 0.5, saturation at 0 / 1 and label
    flips.  Bit-exact: the restatement is the same fp32 expression tree."""
    g = golden("lmac_metrics.npz")
    p, th, po = T(g["predictions"]), T(g["theta_out"]), T(g["masked_predictions"])
    assert torch.equal(lmac_ref.get_score_for_predicted_class(p.squeeze(1)), T(g["score"]))
    assert torch.equal(lmac_ref.compute_faithfulness(p, po), T(g["faithfulness"]))
    assert torch.equal(lmac_ref.compute_fidelity(th, p).view(-1), T(g["fidelity"]))
    assert torch.equal(lmac_ref.compute_AD(th, p), T(g["AD"]))
    assert torch.equal(lmac_ref.compute_AI(th, p), T(g["AI"]))
    assert torch.equal(lmac_ref.compute_AG(th, p), T(g["AG"]))
    m = lmac_ref.lmac_summary(p, th, po)
    for k, v in zip(("faithfulness", "fidelity", "AD", "AI", "AG"), g["means"]):
        assert abs(m[k] - float(v)) <= 1e-6 * max(1.0, abs(float(v))), (k, m[k], v)
