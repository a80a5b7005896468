"""The compact GEMM epilogues (csrc/gemm.hip: gemm_epilogue_tight / _staged / _staged_f32 / _rows_tight, chosen once per
workgroup from the descriptor) against the generic epilogue, which a plan built with the narrow weight packing still runs:
same accumulators, same operation order => BIT-IDENTICAL outputs (GELU in the fp32-class mode: one fp32 ulp, FMA contraction), in both precisions, on ragged M / N, with and without bias,
for the three shapes the Linear layers use (fp16-side output, GELU, fp32 residual stream) and for convolutions with a written
zero halo (row-decomposing forms, leaky / GELU)."""
import pytest
import torch

from addvisor_hip import _lib, gemm as G

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _plans(monkeypatch, build):
    """The same plan twice: wide packing (compact forms) and narrow packing (generic epilogue)."""
    wide = build()
    monkeypatch.setattr(G, "WIDE_EPILOGUE", False)
    narrow = build()
    monkeypatch.undo()
    assert wide.desc.wide == 1 and narrow.desc.wide == 0
    return wide, narrow


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("epi", ["h", "g", "r"])
@pytest.mark.parametrize("M,K,N,bias", [(1000, 256, 192, True), (4776, 768, 768, True), (333, 128, 72, False)])
def test_linear_forms_match_generic(gpu_device, monkeypatch, split, epi, M, K, N, bias):
    _lib.init()
    g = torch.Generator().manual_seed(M + N)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g) if bias else None
    a = torch.randn(M + 256, K, generator=g)
    A = (G.split_planes(a) if split else a.half()).to(gpu_device)
    r0 = torch.randn(M, N, generator=g).to(gpu_device)
    wide, narrow = _plans(monkeypatch, lambda: G.plan_linear(M, w, b, device=gpu_device, split=split, act="gelu" if epi == "g" else "none"))
    outs = []
    for p in (wide, narrow):
        if epi == "r":
            o = r0.clone()
            p.run(A, out_f=o, resid=o)
        else:
            o = torch.full(((2,) if split else ()) + (M, N), 3.0, dtype=torch.float16, device=gpu_device)
            p.run(A, out_h=o)
        outs.append(o)
    if split and epi == "g":
        # the GELU polynomial is contracted into FMAs differently in the two code paths: fp32 values one ulp apart, visible
        # in the lo plane only
        a0, a1 = G.join_planes(outs[0]), G.join_planes(outs[1])
        assert (a0 - a1).abs().max().item() <= 2.5e-7 * a0.abs().max().item()
        assert torch.equal(outs[0][0], outs[1][0]) or ((outs[0][0] != outs[1][0]).float().mean().item() < 1e-3)
    else:
        assert torch.equal(outs[0], outs[1])
    ref = a[:M].double() @ w.double().T + (b.double() if bias else 0.0)
    if epi == "g":
        ref = torch.nn.functional.gelu(ref)
    if epi == "r":
        ref = ref + r0.double().cpu()
        got = outs[0].double().cpu()
    else:
        got = (G.join_planes(outs[0]) if split else outs[0].float()).double().cpu()
    tol = (2e-6 if split else 3e-3) * ref.abs().max().item()
    assert (got - ref).abs().max().item() <= tol


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("Cin,Cout,act,bias,halo", [(32, 64, "leaky", True, (1, 1)), (64, 128, "leaky", True, (2, 2)), (64, 32, "none", False, (0, 0)),
                                                     (32, 256, "leaky", True, (1, 1))])
def test_conv_forms_match_generic(gpu_device, monkeypatch, split, Cin, Cout, act, bias, halo):
    """3x3 same Conv2d on a zero-haloed map: the row-decomposing forms (staged for 64-column wavefront tiles, tight for the
    256x32 tile) write interior AND halo exactly like the generic epilogue."""
    _lib.init()
    B, H, W = 3, 21, 19
    g = torch.Generator().manual_seed(Cin + Cout)
    x = torch.randn(B, H, W, Cin, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5
    b = torch.randn(Cout, generator=g) if bias else None
    src = G.FMap(B, H, W, Cin, 1, 1, split=split).alloc(gpu_device)
    if split:
        src.t[:, :, 1:1 + H, 1:1 + W] = G.split_planes(x).to(gpu_device)
    else:
        src.interior()[:] = x.half().to(gpu_device)
    outs = []
    for wide_on in (True, False):
        if not wide_on:
            monkeypatch.setattr(G, "WIDE_EPILOGUE", False)
        dst = G.FMap(B, H, W, Cout, *halo, split=split).alloc(gpu_device)
        dst.t.fill_(5.0)
        p = G.plan_conv2d([src], dst, w, b, act=act, device=gpu_device)
        assert p.desc.wide == int(wide_on)
        p.run(src.t, out_h=dst.t)
        outs.append(dst.t.clone())
    monkeypatch.undo()
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("with_resid,with_h2,inplace", [(True, True, False), (True, False, True), (False, True, False)])
def test_resblock_forms_match_generic(gpu_device, monkeypatch, split, with_resid, with_h2, inplace):
    """bias, no activation, fp16-side residual and / or the leaky copy (the second convolution of a HiFi-GAN ResBlock step,
    `x = x + conv2(...)`; `inplace`: the output IS the residual buffer): the fp32-staged form against the generic epilogue."""
    _lib.init()
    B, H, W, Cin, Cout = 2, 9, 150, 64, 128
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, H, W, Cin, generator=g)
    r = torch.randn(B, H, W, Cout, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5
    b = torch.randn(Cout, generator=g)
    src = G.FMap(B, H, W, Cin, 1, 1, split=split).alloc(gpu_device)
    res = G.FMap(B, H, W, Cout, 1, 1, split=split).alloc(gpu_device)
    if split:
        src.t[:, :, 1:1 + H, 1:1 + W] = G.split_planes(x).to(gpu_device)
        res.t[:, :, 1:1 + H, 1:1 + W] = G.split_planes(r).to(gpu_device)
    else:
        src.interior()[:] = x.half().to(gpu_device)
        res.interior()[:] = r.half().to(gpu_device)
    outs = []
    for wide_on in (True, False):
        if not wide_on:
            monkeypatch.setattr(G, "WIDE_EPILOGUE", False)
        dst = G.FMap(B, H, W, Cout, 1, 1, split=split).alloc(gpu_device)
        dst2 = G.FMap(B, H, W, Cout, 1, 1, split=split).alloc(gpu_device)
        dst.t.fill_(5.0)
        dst2.t.fill_(6.0)
        rbuf = res.t.clone()
        out = rbuf if inplace else dst.t
        p = G.plan_conv2d([src], dst, w, b, act="none", device=gpu_device)
        p.desc.slope2 = 0.1
        assert p.desc.wide == int(wide_on)
        p.run(src.t, out_h=out, resid=rbuf if with_resid else None, out_h2=dst2.t if with_h2 else None)
        outs.append((out.clone(), dst2.t.clone()))
    monkeypatch.undo()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    ref = torch.nn.functional.conv2d((x.half().float() if not split else x).permute(0, 3, 1, 2), w.half().float() if not split else w, b, padding=1).permute(0, 2, 3, 1)
    if with_resid:
        ref = ref + (r.half().float() if not split else r)
    got = outs[0][0]
    got = (G.join_planes(got) if split else got.float())[:, 1:1 + H, 1:1 + W].cpu()
    assert (got - ref).abs().max().item() <= (3e-6 if split else 6e-3) * ref.abs().max().item()
