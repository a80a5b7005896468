"""GPU parity of the input-gradient kernels against torch autograd on the CPU oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from addvisor_hip import _lib, synthetic as syn
from addvisor_hip.embedder import HipEmbedder
from addvisor_hip.embedder_grad import EmbedderGrad
from oracle import attribution_ref, wav2vec2_ref

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def test_layernorm_bwd(gpu_device):
    _lib.init()
    g = torch.Generator().manual_seed(0)
    M, C = 37, 768
    x = torch.randn(M, C, generator=g, requires_grad=True)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    dy = torch.randn(M, C, generator=g)
    add = torch.randn(M, C, generator=g)
    for gelu in (0, 1):
        with torch.enable_grad():
            y = F.layer_norm(x, (C,), gamma, beta, 1e-5)
            if gelu:
                y = F.gelu(y)
            (ref,) = torch.autograd.grad(y, x, dy)
        d = gpu_device
        out = torch.empty(M, C, device=d)
        xd, dyd, gd, bd, ad = x.detach().to(d), dy.to(d), gamma.to(d), beta.to(d), add.to(d)     # keep the operands alive
        rc = _lib.lib().advh_layernorm_bwd(xd.data_ptr(), 1, dyd.data_ptr(), 1, gd.data_ptr(), bd.data_ptr(), gelu,
                                           ad.data_ptr(), None, out.data_ptr(), None, M, C, 1e-5, 0, 0,
                                           torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert rc == 0
        assert relerr(out.cpu(), ref + add) < 2e-5


@pytest.mark.parametrize("T,heads,D", [(199, 3, 64), (50, 2, 32), (249, 2, 64), (199, 2, 120), (60, 1, 40)])
def test_attention_bwd(gpu_device, T, heads, D):
    _lib.init()
    g = torch.Generator().manual_seed(T)
    B, H = 2, heads * D
    qkv = (torch.randn(B * T, 3 * H, generator=g) * 0.7).half()
    dctx = (torch.randn(B * T, H, generator=g)).half()
    with torch.enable_grad():
        x = qkv.float().requires_grad_(True)
        q, k, v = [t.view(B, T, heads, D).transpose(1, 2) for t in x.split(H, dim=1)]
        a = torch.softmax(q @ k.transpose(2, 3) * D ** -0.5, -1)
        ctx = (a @ v).transpose(1, 2).reshape(B * T, H)
        (ref,) = torch.autograd.grad(ctx, x, dctx.float())
    d = gpu_device
    out = torch.zeros(B * T, 3 * H, dtype=torch.float16, device=d)
    qd, dd = qkv.to(d), dctx.to(d)
    rc = _lib.lib().advh_attention_bwd_f16(qd.data_ptr(), dd.data_ptr(), out.data_ptr(), B, T, H, heads,
                                           torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert rc == 0
    err = relerr(out.float().cpu(), ref)
    print("attention bwd rel err", err)
    assert err < 1e-2            # fp16 P / dS operands and fp16 output


def grad_case(cfg, waves, dev, tol):
    sd = syn.embedder_weights(cfg)
    coef, icpt = syn.logreg_weights(cfg.hidden_size)
    eg = EmbedderGrad(HipEmbedder(cfg, sd, coef, icpt, dev))
    logit, _ = eg.forward(waves.to(dev))
    dx = eg.backward()
    ref = attribution_ref.input_gradient(waves, sd, cfg, coef, icpt)
    with torch.no_grad():
        ref_logit = attribution_ref.model_logit(waves, sd, cfg, coef, icpt)
    assert (logit.cpu() - ref_logit).abs().max().item() < 1e-2
    assert torch.isfinite(dx).all()
    err = relerr(dx.cpu(), ref)
    cos = F.cosine_similarity(dx.cpu().flatten(), ref.flatten(), dim=0).item()
    print(f"input gradient: max rel err {err:.3e}, cosine {cos:.6f}, |ref|max {ref.abs().max():.3e}")
    assert err < tol and cos > 0.999
    return eg, dx


@pytest.mark.parametrize("variant", ["group_postln", "layer_preln", "layer_preln_depth9"])
def test_input_gradient_tiny(gpu_device, variant):
    cfg = {"group_postln": syn.tiny_config(False), "layer_preln": syn.tiny_config(True),
           "layer_preln_depth9": syn.tiny_config(True, num_hidden_layers=9)}[variant]
    grad_case(cfg, syn.make_clips(2, 16000, seed=31), gpu_device, 3e-2)


def test_input_gradient_base_4s(gpu_device):
    eg, dx = grad_case(syn.base_config(), syn.make_clips(1, 64000), gpu_device, 5e-2)
    again = eg.backward()
    assert torch.equal(dx, again)                                  # deterministic


def test_attributions_tiny(gpu_device):
    """Saliency / InputXGradient / IntegratedGradients vs the oracle (Captum semantics restated)."""
    from addvisor_hip.attribution import HipAttribution
    cfg = syn.tiny_config(False)
    sd = syn.embedder_weights(cfg)
    coef, icpt = syn.logreg_weights(cfg.hidden_size)
    model = (sd, cfg, coef, icpt)
    att = HipAttribution(HipEmbedder(cfg, sd, coef, icpt, gpu_device))
    w = syn.make_clips(2, 16000, seed=12)
    wd = w.to(gpu_device)
    for name, ours, ref in (("saliency", att.saliency(wd), attribution_ref.saliency(w, *model)),
                            ("ixg", att.input_x_gradient(wd), attribution_ref.input_x_gradient(w, *model)),
                            ("ig4", att.integrated_gradients(wd, n_steps=4), attribution_ref.integrated_gradients(w, *model, n_steps=4)),
                            ("ig50", att.integrated_gradients(wd, n_steps=50, internal_batch_size=32),
                             attribution_ref.integrated_gradients(w, *model, n_steps=50))):
        err = relerr(ours.cpu(), ref)
        print(name, "max rel err", err)
        assert err < 3e-2, name
    attr = att.integrated_gradients(wd, n_steps=8)
    mask, win, wout = att.time_mask(attr, wd)
    ref_mask = attribution_ref.time_mask(attr.cpu())
    assert torch.allclose(mask.cpu(), ref_mask, atol=1e-6)
    assert torch.allclose((win + wout).cpu(), w, atol=1e-6) and (mask.amax(1) > 0.999).all()
    # The classifier normalises every clip (classifier_embedder.py:59-63), so F(alpha * x) = F(x) for alpha > 0:
    # the path integral of IG is ~0 (not F(x) - F(0): F jumps at alpha = 0).  Size-independent property:
    ig = att.integrated_gradients(wd, n_steps=50)
    assert ig.sum(1).abs().max().item() < 5e-2
