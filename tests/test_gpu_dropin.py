"""GPU: the drop-in modules behave like the reference's (shapes, squeeze rule, errors) and match the oracle /
the reference-generated golden vectors."""
import os

import numpy as np
import pytest
import torch

from addvisor_hip import runtime, synthetic as syn  # noqa: E402
from oracle import lmac_ref, signal_ref, unet_ref, wav2vec2_ref  # noqa: E402

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def tols():
    """Stated tolerances of the drop-in modules at the runtime's precision (ADDVISOR_PRECISION, default f32 = the reference's
    arithmetic class): hidden states / logits / probabilities / loss terms / metric means / attributions (relative)."""
    if runtime.hip_embedder().precision == "f32":
        return dict(hid=1e-4, logit=1e-4, mask=2e-5, loss=1e-4, metric=1e-5, pct=1e-3, attr=1e-4)
    return dict(hid=3e-2, logit=1e-2, mask=5e-3, loss=1e-2, metric=1e-2, pct=2.0, attr=3e-2)


@pytest.fixture(autouse=True)
def tiny_runtime():
    os.environ["ADDVISOR_EMBEDDER"] = "tiny"
    runtime.reset()
    yield
    os.environ.pop("ADDVISOR_EMBEDDER", None)
    runtime.reset()


def test_audioprocessor_api(gpu_device, golden):
    import audioprocessor
    ap = audioprocessor.AudioProcessor(audio_length=1)
    w = syn.make_clips(2, 16000, seed=31)
    X, mag, ph = ap.compute_stft(w)
    assert X.shape == (2, 513, 50) and X.dtype == torch.complex64 and X.is_cuda
    Xs, ms, _ = ap.compute_stft(w[0, :12000])                       # 1-D, short: padded, unbatched result
    assert Xs.shape == (513, 50)
    back = ap.compute_invert_stft(X)
    assert back.shape == (2, 16000) and (back.cpu() - w).abs().max() < 5e-6
    assert ap.compute_invert_stft(Xs).shape == (16000,)
    f2 = ap.extract_features(w)
    f1 = ap.extract_features(w[:1])
    assert f2.shape == (2, 49, 64) and f1.shape == (49, 64)         # squeeze(0) rule, SURVEY D10
    g = golden("embedder_tiny_group.npz")
    t = tols()
    assert (f2.cpu() - torch.from_numpy(g["feats_b2"])).abs().max() < t["hid"]
    assert (f1.cpu() - torch.from_numpy(g["feats_b1"])).abs().max() < t["hid"]
    # the raw module-level call of the reference: wav2vec2(normalised, output_hidden_states=True).hidden_states[9]
    x = audioprocessor.zero_mean_unit_var_norm(w)
    hs = audioprocessor.wav2vec2(x.to(gpu_device), output_hidden_states=True).hidden_states[9]
    assert (hs - f2).abs().max() < t["logit"]      # two fp32 normalisations of the same clip through the same network


def test_unet_module(gpu_device, golden):
    import addvisor
    net = addvisor.ADDvisor().to(gpu_device).eval()
    net.load_state_dict(syn.unet_weights())
    r = np.random.Generator(np.random.PCG64(41))
    xa = torch.from_numpy(r.uniform(0, 3, size=(2, 1, 32, 8)).astype(np.float32))
    out = net(xa.to(gpu_device))
    assert out.shape == (2, 1, 32, 8)
    assert (out.cpu() - torch.from_numpy(golden("unet.npz")["out_a"])).abs().max() < tols()["mask"]
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 1, 513, 249, device=gpu_device))          # the reference shape cannot run


def test_lmac_loss_golden(gpu_device, golden):
    import loss_function
    g = golden("lmac_loss.npz")
    w = syn.make_clips(2, 80000, seed=51)
    _, mag, ph = loss_function.audio_processor.compute_stft(w)
    xhat = torch.from_numpy(np.random.Generator(np.random.PCG64(52)).uniform(0, 1, size=(2, 1, 513, 249)).astype(np.float32))
    _, p = loss_function.audio_processor.classify(w)
    t = tols()
    assert (p.cpu() - torch.from_numpy(g["class_pred"])).abs().max() < t["logit"]
    total, losses, wts = loss_function.LMACLoss().loss_function(xhat.to(gpu_device), mag, ph, torch.from_numpy(g["class_pred"]))
    assert (losses.cpu() - torch.from_numpy(g["losses"])).abs().max() < t["loss"]
    assert torch.allclose(wts.cpu(), torch.from_numpy(g["w"]), atol=1e-6)
    assert abs(total.item() - float(g["total"])) < 5 * t["loss"]


def test_run_addvisor_metrics_synthetic_dataset(gpu_device, capsys):
    import LMAC_metrics
    LMAC_metrics.audio_processor.audio_length = 1

    class DS(torch.utils.data.Dataset):
        def __init__(self):
            self.w = syn.make_clips(6, 16000, seed=91)

        def __len__(self):
            return 6

        def __getitem__(self, i):
            return self.w[i].to(gpu_device), f"clip{i}.wav"

    try:
        m = LMAC_metrics.run_addvisor_metrics("", "", batch_size=4, dataset=DS())
    finally:
        LMAC_metrics.audio_processor.audio_length = 5
    printed = capsys.readouterr().out.strip().splitlines()
    assert [l.split(":")[0].strip() for l in printed] == ["faithfulness", "fidelity", "average drop", "average increase", "average gain"]
    cfg, sd = runtime.embedder_config_and_weights()
    clf = runtime.classifier()
    ref = lmac_ref.explain(DS().w, sd, cfg, clf.coef_, clf.intercept_, syn.unet_weights(), audio_length=1)
    r = lmac_ref.lmac_summary(ref["predictions"], ref["theta_out"], ref["masked_predictions"])
    t = tols()
    assert abs(m["faithfulness"] - r["faithfulness"]) < t["metric"] and abs(m["fidelity"] - r["fidelity"]) < 1e-6
    assert abs(m["AD"] - r["AD"]) < t["pct"] and abs(m["AI"] - r["AI"]) < 1e-3 and abs(m["AG"] - r["AG"]) < t["pct"]


def test_captum_compatible_api(gpu_device):
    """``from captum.attr import ...`` resolves to the HIP attribution path (captum_saliency.py:3, 116-135)."""
    import captum_saliency as cs
    from captum.attr import InputXGradient, IntegratedGradients, Saliency
    from oracle import attribution_ref
    model = cs.Wav2vec2LogReg(cs.audioprocessor, cs.TorchLogReg()).to(gpu_device)
    w = syn.make_clips(2, 16000, seed=12)
    cfg, sd = runtime.embedder_config_and_weights()
    clf = runtime.classifier()
    m = (sd, cfg, clf.coef_, clf.intercept_)
    x = w.to(gpu_device).clone().detach().requires_grad_(True)        # as captum_saliency.py:129
    sal = Saliency(model).attribute(inputs=x, target=None)
    ixg = InputXGradient(model).attribute(inputs=x, target=None)
    ig = IntegratedGradients(model).attribute(inputs=x, target=None, n_steps=8)
    rel = lambda a, b: ((a.cpu() - b).abs().max() / b.abs().max()).item()
    t = tols()
    assert rel(sal, attribution_ref.saliency(w, *m)) < t["attr"]
    assert rel(ixg, attribution_ref.input_x_gradient(w, *m)) < t["attr"]
    assert rel(ig, attribution_ref.integrated_gradients(w, *m, n_steps=8)) < t["attr"]
    assert (model(w.to(gpu_device)).cpu() - attribution_ref.model_logit(w, *m).detach()).abs().max() < t["logit"]
    p, t, o = cs.explain_waves(model, w, "saliency")
    assert p.shape == t.shape == o.shape == (2, 1)
