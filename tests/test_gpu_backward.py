"""GPU parity of the input-gradient kernels against torch autograd on the CPU oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from addvisor_hip import _lib, gemm as G, synthetic as syn
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


def test_layernorm_bwd_split(gpu_device):
    """Split-format operands (x, dy, dact_src, out_h as hi/lo plane pairs): 22-bit inputs, fp32 arithmetic."""
    _lib.init()
    g = torch.Generator().manual_seed(3)
    M, C = 37, 768
    d = gpu_device
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    xs, dys, zs = (G.split_planes(torch.randn(M, C, generator=g) * sc) for sc in (1.0, 3.0, 1.5))
    x, dy, zpre = G.join_planes(xs), G.join_planes(dys), G.join_planes(zs)
    add = torch.randn(M, C, generator=g)
    for gelu in (0, 1):
        with torch.enable_grad():
            xr = x.clone().requires_grad_(True)
            y = F.layer_norm(xr, (C,), gamma, beta, 1e-5)
            if gelu:
                y = F.gelu(y)
            (ref,) = torch.autograd.grad(y, xr, dy)
        with torch.enable_grad():
            zr = zpre.clone().requires_grad_(True)
            (gd,) = torch.autograd.grad(F.gelu(zr).sum(), zr)
        ref = ref * gd + add
        xd, dyd, zd, gd_, bd, ad = xs.to(d), dys.to(d), zs.to(d), gamma.to(d), beta.to(d), add.to(d)
        out_f = torch.empty(M, C, device=d)
        out_h = torch.zeros(2, M, C, dtype=torch.float16, device=d)
        rc = _lib.lib().advh_layernorm_bwd_split(xd.data_ptr(), 0, xd.stride(0), dyd.data_ptr(), 0, dyd.stride(0), gd_.data_ptr(),
                                                 bd.data_ptr(), gelu, ad.data_ptr(), zd.data_ptr(), zd.stride(0), out_f.data_ptr(),
                                                 out_h.data_ptr(), out_h.stride(0), M, C, 1e-5, 0, 0,
                                                 torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert rc == 0
        assert relerr(out_f.cpu(), ref) < 2e-5
        assert relerr(G.join_planes(out_h.cpu()), out_f.cpu()) < 1e-6          # the split output carries the fp32 result to ~2^-22


@pytest.mark.parametrize("kernel", ["default", "mfma_f32"])
@pytest.mark.parametrize("T,heads,D", [(199, 3, 64), (50, 2, 32), (249, 2, 64), (199, 2, 120), (60, 1, 40), (256, 1, 128), (33, 2, 16), (208, 1, 48),
                                       (209, 2, 64), (128, 2, 64), (17, 1, 8)])
def test_attention_bwd_split(gpu_device, T, heads, D, kernel):
    """fp32-class attention backward on split-format q | k | v and dctx vs fp64 autograd: head dims <= 64 in split arithmetic
    (attention_bwd_x3.hip: three fp16 MFMAs per product, transposing LDS reads; every tile-count instance incl. the 4-wavefront one
    for T > 208), larger head dims -- and every head dim under ``attention_bwd_mfma_f32`` -- on v_mfma_f32_16x16x4_f32."""
    _lib.init()
    if kernel == "mfma_f32" and D > 64:
        pytest.skip("head dims > 64 run the fp32-MFMA kernel by default")
    _lib.check(_lib.lib().advh_set_option(b"attention_bwd_mfma_f32", int(kernel == "mfma_f32")), "advh_set_option")
    g = torch.Generator().manual_seed(T + D)
    B, H = 2, heads * D
    qs = G.split_planes(torch.randn(B * T, 3 * H, generator=g) * 0.7)
    ds = G.split_planes(torch.randn(B * T, H, generator=g))
    with torch.enable_grad():
        x = G.join_planes(qs).double().requires_grad_(True)
        q, k, v = [t.view(B, T, heads, D).transpose(1, 2) for t in x.split(H, dim=1)]
        a = torch.softmax(q @ k.transpose(2, 3) * D ** -0.5, -1)
        ctx = (a @ v).transpose(1, 2).reshape(B * T, H)
        (ref,) = torch.autograd.grad(ctx, x, G.join_planes(ds).double())
    d = gpu_device
    out = torch.full((2, B * T, 3 * H), float("nan"), dtype=torch.float16, device=d)
    qd, dd = qs.to(d), ds.to(d)
    rc = _lib.lib().advh_attention_bwd_split(qd.data_ptr(), qd.stride(0), dd.data_ptr(), dd.stride(0), out.data_ptr(), out.stride(0),
                                             B, T, H, heads, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    _lib.lib().advh_set_option(b"attention_bwd_mfma_f32", 0)
    assert rc == 0
    got = G.join_planes(out.cpu()).double()
    assert torch.isfinite(got).all()                                       # every element of dqkv is written
    for name, sl in (("dq", slice(0, H)), ("dk", slice(H, 2 * H)), ("dv", slice(2 * H, 3 * H))):
        err = relerr(got[:, sl], ref[:, sl])
        print(f"attention bwd split {name} rel err {err:.2e}")
        assert err < 5e-6, name


# fp32-class chain (the reference's fp32 autograd class): 1e-4 of max|ref|, cosine > 0.999999; fp16 chain: round 1's tolerances
TOL = {"f32": (1e-4, 0.999999, 1e-4), "f16": (None, 0.999, 1e-2)}


def grad_case(cfg, waves, dev, tol, precision="f32"):
    sd = syn.embedder_weights(cfg)
    coef, icpt = syn.logreg_weights(cfg.hidden_size)
    eg = EmbedderGrad(HipEmbedder(cfg, sd, coef, icpt, dev, precision=precision))
    assert eg.precision == precision
    logit, _ = eg.forward(waves.to(dev))
    dx = eg.backward()
    ref = attribution_ref.input_gradient(waves, sd, cfg, coef, icpt)
    with torch.no_grad():
        ref_logit = attribution_ref.model_logit(waves, sd, cfg, coef, icpt)
    tol_f32, cos_min, tol_logit = TOL[precision]
    tol = tol_f32 if tol_f32 is not None else tol
    assert (logit.cpu() - ref_logit).abs().max().item() < tol_logit
    assert torch.isfinite(dx).all()
    err = relerr(dx.cpu(), ref)
    cos = F.cosine_similarity(dx.cpu().double().flatten(), ref.double().flatten(), dim=0).item()
    print(f"input gradient [{precision}]: max rel err {err:.3e}, cosine {cos:.8f}, |ref|max {ref.abs().max():.3e}")
    assert err < tol and cos > cos_min
    return eg, dx


@pytest.mark.parametrize("precision", ["f32", "f16"])
@pytest.mark.parametrize("variant", ["group_postln", "layer_preln", "layer_preln_depth9"])
def test_input_gradient_tiny(gpu_device, variant, precision):
    cfg = {"group_postln": syn.tiny_config(False), "layer_preln": syn.tiny_config(True),
           "layer_preln_depth9": syn.tiny_config(True, num_hidden_layers=9)}[variant]
    grad_case(cfg, syn.make_clips(2, 16000, seed=31), gpu_device, 3e-2, precision)


@pytest.mark.parametrize("precision", ["f32", "f16"])
def test_input_gradient_base_4s(gpu_device, precision):
    eg, dx = grad_case(syn.base_config(), syn.make_clips(1, 64000), gpu_device, 5e-2, precision)
    again = eg.backward()
    assert torch.equal(dx, again)                                  # deterministic


def test_f16_chain_next_to_f32_embedder(gpu_device):
    """``EmbedderGrad(emb, precision="f16")`` keeps the fp16 chain selectable under an fp32-class embedder."""
    cfg = syn.tiny_config(False)
    sd = syn.embedder_weights(cfg)
    coef, icpt = syn.logreg_weights(cfg.hidden_size)
    emb = HipEmbedder(cfg, sd, coef, icpt, gpu_device, precision="f32")
    assert EmbedderGrad(emb).precision == "f32" and EmbedderGrad(emb, precision="f16").precision == "f16"
    with pytest.raises(ValueError):
        EmbedderGrad(HipEmbedder(cfg, sd, coef, icpt, gpu_device, precision="f16"), precision="f32")


@pytest.mark.parametrize("precision", ["f32", "f16"])
def test_attributions_tiny(gpu_device, precision):
    """Saliency / InputXGradient / IntegratedGradients vs the oracle (Captum semantics restated)."""
    from addvisor_hip.attribution import HipAttribution
    cfg = syn.tiny_config(False)
    sd = syn.embedder_weights(cfg)
    coef, icpt = syn.logreg_weights(cfg.hidden_size)
    model = (sd, cfg, coef, icpt)
    att = HipAttribution(HipEmbedder(cfg, sd, coef, icpt, gpu_device, precision=precision))
    assert att.precision == precision
    tol = 1e-4 if precision == "f32" else 3e-2
    w = syn.make_clips(2, 16000, seed=12)
    wd = w.to(gpu_device)
    for name, ours, ref in (("saliency", att.saliency(wd), attribution_ref.saliency(w, *model)),
                            ("ixg", att.input_x_gradient(wd), attribution_ref.input_x_gradient(w, *model)),
                            ("ig4", att.integrated_gradients(wd, n_steps=4), attribution_ref.integrated_gradients(w, *model, n_steps=4)),
                            ("ig50", att.integrated_gradients(wd, n_steps=50, internal_batch_size=32),
                             attribution_ref.integrated_gradients(w, *model, n_steps=50))):
        err = relerr(ours.cpu(), ref)
        print(name, precision, "max rel err", err)
        assert err < tol, name
    attr = att.integrated_gradients(wd, n_steps=8)
    mask, win, wout = att.time_mask(attr, wd)
    ref_mask = attribution_ref.time_mask(attr.cpu())
    assert torch.allclose(mask.cpu(), ref_mask, atol=1e-6)
    assert torch.allclose((win + wout).cpu(), w, atol=1e-6) and (mask.amax(1) > 0.999).all()
    # The classifier normalises every clip (classifier_embedder.py:59-63), so F(alpha * x) = F(x) for alpha > 0:
    # the path integral of IG is ~0 (not F(x) - F(0): F jumps at alpha = 0).  Size-independent property:
    ig = att.integrated_gradients(wd, n_steps=50)
    assert ig.sum(1).abs().max().item() < (5e-3 if precision == "f32" else 5e-2)


def test_attribution_overflow_raises(gpu_device):
    """The planes between dgrad GEMMs have fp16's exponent range: a loss scale that overflows them must raise, not return
    inf / NaN attributions."""
    from addvisor_hip.attribution import HipAttribution
    cfg = syn.tiny_config(False)
    sd = syn.embedder_weights(cfg)
    coef, icpt = syn.logreg_weights(cfg.hidden_size)
    for precision in ("f32", "f16"):
        att = HipAttribution(HipEmbedder(cfg, sd, coef, icpt, gpu_device, precision=precision), loss_scale=2.0 ** 40)
        with pytest.raises(FloatingPointError):
            att.saliency(syn.make_clips(1, 16000, seed=5).to(gpu_device))
