"""GPU parity of the U-Net TRAINING step on the HIP kernels (addvisor.py:12-84 in train()
mode under train_addvisor.py:364-378).

Default precision = the fp32-class mode (split-format maps and gradients, three MFMAs per product: the reference trains in
fp32), where BOTH levels below are tight: per layer 2e-5 of max|ref|, end to end against fp32 autograd through the CPU oracle
cosine >= 0.9999 and relative L2 <= 1e-3 for EVERY parameter with the reference's LeakyReLU(0.2).  The fp16 mode
(precision="f16") keeps round 1's two-level statement:

Two levels, because LeakyReLU makes an end-to-end gradient comparison at fp16 inherently loose: a pre-activation that
the fp16 forward rounds across zero flips the local slope (1 vs 0.2) on a ~4e-4 fraction of the elements, a sparse
error of relative L2 size sqrt(fraction) ~ 2 % per layer that no kernel can avoid (torch AMP shows the same).
  (1) per-layer, tight: every weight gradient and every activation gradient the HIP path produces is compared with
      torch's conv2d_weight / conv_transpose2d / batch_norm autograd evaluated on the HIP path's OWN saved operands
      (|err| <= 4e-3 * max|ref|): this pins the transposes, the split-K GEMM indexing, the dgrad plans, the
      BatchNorm backward and the skip accumulation for all 22 layers.
  (2) end-to-end vs autograd through the CPU oracle (oracle/unet_ref.py, batch-statistics BatchNorm): mask
      |err| <= 1.5e-2; parameter gradients cosine >= 0.95 with the reference slope 0.2, and -- the kink-free control,
      LeakyReLU slope 1.0 on both sides, same kernels and data flow -- cosine >= 0.9999, relative L2 <= 1.5e-2
      (measured 0.999998); a descent step along the HIP gradient lowers the loss."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F
from torch.nn import grad as nngrad

from addvisor_hip import gemm as G, synthetic as syn
from addvisor_hip.unet_train import HipUNetTrain, BN_EPS, SLOPE
from oracle import unet_ref

pytestmark = pytest.mark.gpu


def setup(dev, B, H, W, seed, precision="f32"):
    sd = syn.unet_weights(seed=seed)
    gen = torch.Generator().manual_seed(seed + 1)
    mag = torch.rand(B, H, W, generator=gen) * 3.0
    # a structured upstream gradient, as a real loss gives: d/dmask of mean((mask - target)^2) with a smooth target
    target = F.interpolate(torch.rand(B, 1, max(H // 8, 1), max(W // 4, 1), generator=gen), size=(H, W), mode="bilinear",
                           align_corners=False)[:, 0]
    names = [k for k in sd if k.endswith("weight") or k.endswith("bias")]
    with torch.enable_grad():
        ref_sd = {k: (v.clone().float().requires_grad_(True) if k in names else v.clone()) for k, v in sd.items()}
        ref_mask = unet_ref.unet_forward(mag[:, None], ref_sd, bn_batch=True)[:, 0]
        dmask = (2.0 * (ref_mask.detach() - target) / ref_mask.numel())
        ref_grads = dict(zip(names, torch.autograd.grad((ref_mask * dmask).sum(), [ref_sd[k] for k in names], allow_unused=True)))

    params = {k: v.clone().float().to(dev) for k, v in sd.items()}
    net = HipUNetTrain(params, dev, precision=precision)
    assert net.precision == precision
    return net, params, mag, target, dmask, ref_mask.detach(), ref_grads


def interior(f, t=None):
    t = f.t if t is None else t
    t = G.join_planes(t) if f.split else t.float()            # fp32-class mode: [2, B, Hp, Wp, C] plane pair
    return t[:, f.PH:f.PH + f.H, f.PW:f.PW + f.W].cpu().permute(0, 3, 1, 2).contiguous()     # NCHW fp32


def close(a, b, tol, what):
    scale = b.abs().max().item()
    err = (a - b).abs().max().item()
    assert err <= tol * scale + 1e-9, (what, err, scale)
    return err / (scale + 1e-30)


@pytest.mark.parametrize("precision", ["f32", "f16"])
@pytest.mark.parametrize("B,H,W,seed", [(2, 32, 8, 5), (3, 64, 24, 7)])
def test_every_layer_backward_is_exact_on_its_own_operands(gpu_device, B, H, W, seed, precision):
    net, params, mag, target, dmask, _, _ = setup(gpu_device, B, H, W, seed, precision)
    TOL = 2e-5 if precision == "f32" else 4e-3
    wq = (lambda w: w) if precision == "f32" else (lambda w: w.half().float())     # the operand precision of the dgrad weights
    net.forward(mag.to(gpu_device), H=H, W=W)
    grads = net.backward(dmask.to(gpu_device))
    ws = net._workspace(B, H, W)
    m, z, g = ws["maps"], ws["z"], ws["g"]
    worst = 0.0
    # loss scale: recover it from the head (grads are unscaled, maps are scaled)
    gy1 = interior(g["y1"])
    dl = ws["dlogit"].cpu()
    hw = params["mask_head.0.weight"].cpu().reshape(32)
    S = (gy1[:, 0] / (dl * hw[0] + 1e-30)).median().item()
    assert abs(np.log2(S) - round(np.log2(S))) < 1e-3        # a power of two
    acc = {}
    for L in ws["layers"]:
        if L["kind"] == "up":
            name, (sh, sw) = L["name"], L["stride"]
            x, gy = interior(L["src"]), interior(L["gdst"])
            w = params[name + ".weight"].cpu()
            dw = torch.einsum("bchw,bdhiwj->cdij", x, gy.view(B, gy.shape[1], x.shape[2], sh, x.shape[3], sw))
            worst = max(worst, close(grads[name + ".weight"].cpu() * S, dw, TOL, name + ".weight"))
            worst = max(worst, close(grads[name + ".bias"].cpu() * S, gy.sum((0, 2, 3)), TOL, name + ".bias"))
            a = acc.setdefault(id(L["gsrc"]), [L["gsrc"], 0.0])
            a[1] = a[1] + F.conv2d(gy, wq(w), stride=(sh, sw))
            continue
        cname, bname, dst = L["cname"], L["bname"], L["dst"]
        (KH, KW), (sh, sw), pad, dil = L["k"], L["stride"], L["pad"], L["dil"]
        dzm = L["dz"]
        dz = interior(dzm)[:, :, ::sh, ::sw] if L["srcs"] != ["mag"] else interior(dzm)
        dz = dz[:, :, :m[dst].H, :m[dst].W]
        # BatchNorm + LeakyReLU backward on HIP's own z and incoming gradient
        zz = interior(z[dst]).double()
        gin = interior(g[dst]).double()
        gamma, beta = params[bname + ".weight"].cpu().double(), params[bname + ".bias"].cpu().double()
        with torch.enable_grad():
            zr = zz.clone().requires_grad_(True)
            gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
            y = F.leaky_relu(F.batch_norm(zr, None, None, gr, br, True, 0.0, BN_EPS), SLOPE)
            dzr, dgr, dbr = torch.autograd.grad((y * gin).sum(), [zr, gr, br])
        worst = max(worst, close(dz.double(), dzr, TOL, cname + " dz"))
        worst = max(worst, close(grads[bname + ".weight"].cpu().double() * S, dgr, TOL, bname + ".weight"))
        worst = max(worst, close(grads[bname + ".bias"].cpu().double() * S, dbr, TOL, bname + ".bias"))
        # weight gradient on HIP's own input activations and dz
        w = params[cname + ".weight"].cpu()
        if L["srcs"] == ["mag"]:
            x = mag[:, None, :H, :W]
        else:
            x = torch.cat([interior(m[s]) for s in L["srcs"]], 1)[:, :w.shape[1]]
        dw = nngrad.conv2d_weight(x, w.shape, dz, stride=(sh, sw), padding=pad, dilation=dil)
        worst = max(worst, close(grads[cname + ".weight"].cpu() * S, dw, TOL, cname + ".weight"))
        assert grads[cname + ".bias"].abs().max().item() == 0.0
        if L["srcs"] == ["mag"]:
            continue
        # activation gradients: contributions of this layer to each source map
        dx = nngrad.conv2d_input(x.shape, wq(w), dz, stride=(sh, sw), padding=pad, dilation=dil)
        lo = 0
        for s in L["srcs"]:
            c = g[s].C
            a = acc.setdefault(id(g[s]), [g[s], 0.0])
            a[1] = a[1] + dx[:, lo:lo + c]
            lo += m[s].C
    for f, ref in acc.values():
        worst = max(worst, close(interior(f), ref, TOL, "activation gradient"))
    print(f"per-layer backward parity [{precision}] B={B} {H}x{W}: worst max-rel err {worst:.2e}; loss scale 2^{int(round(np.log2(S)))}")


@pytest.mark.parametrize("precision", ["f32", "f16"])
@pytest.mark.parametrize("B,H,W,seed,slope", [(2, 32, 8, 5, 0.2), (3, 64, 24, 7, 0.2), (3, 64, 24, 7, 1.0)])
def test_train_step_against_oracle_autograd(gpu_device, monkeypatch, B, H, W, seed, slope, precision):
    """slope 0.2 = the reference network.  slope 1.0 = the kink-free control: the same kernels, launches and data
    flow with LeakyReLU turned into the identity on both sides, where end-to-end agreement must be (and is) tight."""
    import addvisor_hip.unet_train as UT
    if slope != SLOPE:
        lrelu = F.leaky_relu
        monkeypatch.setattr(UT, "SLOPE", slope)
        monkeypatch.setattr(unet_ref.F, "leaky_relu", lambda x, s=0.2, **kw: lrelu(x, slope))
    net, params, mag, target, dmask, ref_mask, ref_grads = setup(gpu_device, B, H, W, seed, precision)
    f32 = precision == "f32"
    rm0 = params["e2.block.1.running_mean"].clone()
    mask = net.forward(mag.to(gpu_device), H=H, W=W)
    err = (mask.cpu() - ref_mask).abs().max().item()
    print(f"train-mode forward [{precision}] B={B} {H}x{W} slope {slope}: mask max err {err:.2e}")
    assert err <= (2e-5 if f32 else 1.5e-2)
    assert not torch.equal(params["e2.block.1.running_mean"], rm0)                   # running statistics were updated
    grads = net.backward(dmask.to(gpu_device))
    worst = (1.0, "", 0.0)
    for k, r in ref_grads.items():
        gk = grads[k].cpu().reshape(r.shape)
        assert torch.isfinite(gk).all(), k
        if gk.abs().max().item() == 0.0:                                            # conv bias before a batch-stat BatchNorm
            assert r.abs().max().item() <= 1e-6 * max(1.0, dmask.abs().sum().item()), k
            continue
        cos = F.cosine_similarity(gk.flatten().double(), r.flatten().double(), dim=0).item()
        rel2 = ((gk - r).norm() / r.norm()).item()
        if cos < worst[0]:
            worst = (cos, k, rel2)
        if f32:                                                  # fp32-class mode: tight with the reference's LeakyReLU(0.2) too
            assert cos >= 0.9999 and rel2 <= 1e-3, (k, cos, rel2)
        elif slope == 1.0:
            assert cos >= 0.9999 and rel2 <= 1.5e-2, (k, cos, rel2)
        else:
            assert cos >= 0.95, (k, cos, rel2)
    print(f"end-to-end parameter gradients [{precision}] (slope {slope}): worst cosine {worst[0]:.6f} (rel L2 {worst[2]:.4f}) at {worst[1]}")
    # a small step along the negative HIP gradient lowers the loss mean((mask - target)^2)
    loss0 = ((mask.cpu() - target) ** 2).mean().item()
    gn = max(v.abs().max().item() for v in grads.values())
    for k, v in grads.items():
        params[k].sub_(v.reshape(params[k].shape) * (2e-3 / gn))
    loss1 = ((net.forward(mag.to(gpu_device), H=H, W=W).cpu() - target) ** 2).mean().item()
    print(f"loss {loss0:.6f} -> {loss1:.6f}")
    assert loss1 < loss0


def test_backward_is_deterministic(gpu_device):
    net, params, mag, target, dmask, _, _ = setup(gpu_device, 2, 32, 8, 9)
    net.forward(mag.to(gpu_device), H=32, W=8); g1 = net.backward(dmask.to(gpu_device))
    net.forward(mag.to(gpu_device), H=32, W=8); g2 = net.backward(dmask.to(gpu_device))     # only the running buffers moved
    assert all(torch.equal(g1[k], g2[k]) for k in g1)


@pytest.mark.parametrize("CI,CO,Cx,cx0,Cz,cz0,B,H,W", [(32, 32, 32, 0, 32, 0, 2, 40, 24), (64, 64, 64, 0, 64, 0, 2, 24, 20), (64, 64, 192, 64, 128, 64, 3, 16, 12),
                                                       (32, 64, 96, 64, 128, 0, 2, 19, 33), (64, 32, 128, 0, 96, 64, 2, 9, 17), (32, 32, 64, 32, 96, 32, 1, 35, 50)])
def test_wgrad2d_split_slice_pairs(gpu_device, CI, CO, Cx, cx0, Cz, cz0, B, H, W):
    """``advh_conv_wgrad2d_split`` (csrc/conv_wgrad.hip: transposing LDS reads, three fp16 MFMAs per fragment pair) on one (input slice,
    output slice) pair of wider split-format maps with different halos, ragged tile edges included, against the fp64 weight gradient of a
    3x3 "same" convolution (addvisor.py:20-24 under train_addvisor.py:376) on the joined values."""
    import ctypes as C
    from addvisor_hip import _lib
    from addvisor_hip.unet_train import Wgrad2dDesc
    _lib.init()
    g = torch.Generator().manual_seed(CI + CO + H)
    x = G.FMap(B, H, W, Cx, 2, 1, split=True).alloc(gpu_device)
    z = G.FMap(B, H, W, Cz, 1, 3, split=True).alloc(gpu_device)
    xs, zs = G.split_planes(torch.randn(B, H, W, Cx, generator=g)), G.split_planes(torch.randn(B, H, W, Cz, generator=g) * 0.3)
    x.t[:, :, 2:2 + H, 1:1 + W] = xs.to(gpu_device)
    z.t[:, :, 1:1 + H, 3:3 + W] = zs.to(gpu_device)
    lib = _lib.lib()
    parts = lib.advh_conv_wgrad2d_split_parts(CI, CO, B, H, W)
    part = torch.empty(parts * 9 * CI * CO, dtype=torch.float32, device=gpu_device)
    dw = torch.full((9, CO, CI), float("nan"), dtype=torch.float32, device=gpu_device)
    d = Wgrad2dDesc(B=B, H=H, W_=W, PHx=2, PWx=1, PHz=1, PWz=3)
    d.X, d.DZ, d.partial = x.t.data_ptr(), z.t.data_ptr(), part.data_ptr()
    _lib.check(lib.advh_conv_wgrad2d_split(C.byref(d), CI, CO, Cx, cx0, Cz, cz0, x.t.stride(0), z.t.stride(0), dw.data_ptr(),
                                            torch.cuda.current_stream().cuda_stream), "advh_conv_wgrad2d_split")
    torch.cuda.synchronize()
    xj = G.join_planes(xs).double()[..., cx0:cx0 + CI].permute(0, 3, 1, 2)          # [B, CI, H, W]
    zj = G.join_planes(zs).double()[..., cz0:cz0 + CO].permute(0, 3, 1, 2)
    ref = nngrad.conv2d_weight(xj, (CO, CI, 3, 3), zj, padding=1)                    # [CO, CI, 3, 3]
    got = dw.cpu().double().view(3, 3, CO, CI).permute(2, 3, 0, 1)
    assert torch.isfinite(got).all()
    err = float((got - ref).abs().max() / ref.abs().max())
    print(f"wgrad2d split {CI}x{CO} slice of {Cx}x{Cz}: rel err {err:.2e}")
    assert err < 2e-6
