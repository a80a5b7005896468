"""GPU parity of the implicit-GEMM kernel (csrc/gemm.hip) against torch CPU convolutions / matmuls
evaluated on the same fp16-rounded operands (fp32 math).  Stated tolerance: outputs are fp16, so
|err| <= 2e-3 * max|ref| + 2e-3 (fp16 has 11 significand bits; accumulation is fp32)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from addvisor_hip import gemm as G, _lib

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def rnd(gen, *shape):
    return torch.randn(*shape, generator=gen)


def close(out, ref, tol=2e-3):
    out, ref = out.float().cpu(), ref.float()
    err = (out - ref).abs().max().item()
    assert err <= tol * ref.abs().max().item() + tol, (err, ref.abs().max().item())


@pytest.mark.parametrize("M,K,N,act", [(300, 768, 2304, "none"), (12736, 768, 768, "gelu"), (1000, 3072, 768, "none"),
                                      (257, 512, 64, "gelu"), (513, 64, 32, "leaky"), (128, 64, 48, "none")])
def test_linear(gpu_device, M, K, N, act):
    _lib.init()
    g = torch.Generator().manual_seed(M + N)
    a, w, b = rnd(g, M, K).half(), rnd(g, N, K) / K ** 0.5, rnd(g, N)
    res = rnd(g, M, N)
    p = G.plan_linear(M, w, b, act=act, device=gpu_device)
    pad = torch.zeros(1024, K, dtype=torch.float16)             # slack rows: tail tiles read a safe row, never OOB
    A = torch.cat([a, pad]).to(gpu_device)
    out_h = torch.zeros(M, N, dtype=torch.float16, device=gpu_device)
    out_f = torch.zeros(M, N, dtype=torch.float32, device=gpu_device)
    p.run(A, out_h=out_h, out_f=out_f, resid=res.to(gpu_device))
    y = a.float() @ w.half().float().T + b
    y = F.gelu(y) if act == "gelu" else (F.leaky_relu(y, 0.0) if act == "leaky" else y)
    close(out_f, y + res, 2e-4)
    close(out_h, y + res)
    # fp16 residual (the other branch of the affine-row kernel's lean epilogue) and no residual at all
    p.run(A, out_f=out_f, resid=res.half().to(gpu_device))
    close(out_f, y + res.half().float(), 2e-4)
    p.run(A[:M].contiguous(), out_h=out_h)                       # exactly M rows: the tail tile re-reads row M-1, never past the end
    close(out_h, y)


def test_conv1d_feature_encoder_shape(gpu_device):
    """Layer-1 shape of the wav2vec2 feature encoder (k=3, s=2, 512->512) on a short clip."""
    _lib.init()
    g = torch.Generator().manual_seed(1)
    B, C, L_in = 3, 512, 1599
    L_out = (L_in - 3) // 2 + 1
    P_out = L_out + 1
    P_in = 2 * P_out
    x, w = rnd(g, B, C, L_in), rnd(g, C, C, 3) / (3 * C) ** 0.5
    xin = torch.zeros(B, P_in, C, dtype=torch.float16)
    xin[:, :L_in] = x.transpose(1, 2).half()
    ref = F.gelu(F.conv1d(xin[:, :L_in].float().transpose(1, 2), w.half().float(), None, stride=2)).transpose(1, 2)
    p = G.plan_conv1d_cl(B, P_in, P_out, L_out, w, None, 2, device=gpu_device)
    out = torch.full((B, P_out, C), 7.0, dtype=torch.float16, device=gpu_device)
    p.run(xin.to(gpu_device), out_h=out)
    close(out[:, :L_out], ref)
    assert (out[:, L_out:] == 0).all()
    pc = G.plan_conv1d_cl(B, P_in, P_out, L_out, w, None, 2, compact_out=True, device=gpu_device)
    outc = torch.zeros((B, L_out, C), dtype=torch.float16, device=gpu_device)
    pc.run(xin.to(gpu_device), out_h=outc)
    close(outc, ref)


CASES = [
    # Cins, Cout, k, stride, pad, dil, H, W, halo_in, halo_out
    ([32], 32, (3, 3), (1, 1), (1, 1), (1, 1), 40, 28, (1, 1), (1, 1)),
    ([32], 64, (5, 3), (2, 1), (2, 1), (1, 1), 64, 20, (2, 1), (1, 1)),
    ([64], 128, (3, 3), (2, 2), (1, 1), (1, 1), 32, 28, (1, 1), (1, 1)),
    ([256], 512, (3, 3), (1, 1), (2, 2), (2, 2), 16, 13, (2, 2), (4, 4)),
    ([512], 512, (3, 3), (1, 1), (4, 4), (4, 4), 8, 13, (4, 4), (0, 0)),
    ([256, 128], 256, (3, 3), (1, 1), (1, 1), (1, 1), 16, 12, (1, 1), (1, 1)),
    ([64, 32], 64, (3, 3), (1, 1), (1, 1), (1, 1), 24, 20, (1, 1), (1, 1)),
    ([32, 8], 32, (3, 3), (1, 1), (1, 1), (1, 1), 24, 20, (1, 1), (0, 0)),
]


@pytest.mark.parametrize("case", CASES)
def test_conv2d(gpu_device, case):
    _lib.init()
    Cins, Cout, k, stride, pad, dil, H, W, halo_in, halo_out = case
    B = 3
    g = torch.Generator().manual_seed(Cout + H)
    xs = [rnd(g, B, c, H, W) for c in Cins]
    w, b = rnd(g, Cout, sum(Cins), *k) / (sum(Cins) * k[0] * k[1]) ** 0.5, rnd(g, Cout)
    srcs = []
    for c, x in zip(Cins, xs):
        f = G.FMap(B, H, W, c, *halo_in).alloc(gpu_device)
        f.interior()[:] = x.permute(0, 2, 3, 1).half().to(gpu_device)
        srcs.append(f)
    ref = F.leaky_relu(F.conv2d(torch.cat([x.half().float() for x in xs], 1), w.half().float(), b, stride=stride,
                                padding=pad, dilation=dil), 0.2)
    dst = G.FMap(B, ref.shape[2], ref.shape[3], Cout, *halo_out).alloc(gpu_device)
    dst.t.fill_(5.0)
    p = G.plan_conv2d(srcs, dst, w, b, stride=stride, padding=pad, dilation=dil, device=gpu_device)
    p.run(srcs[0].t, srcs[1].t if len(srcs) > 1 else None, out_h=dst.t)
    close(dst.interior().permute(0, 3, 1, 2), ref)
    t = dst.t.clone()
    t[:, dst.PH:dst.PH + dst.H, dst.PW:dst.PW + dst.W] = 0
    assert (t == 0).all()


@pytest.mark.parametrize("stride,Cin,Cout", [((2, 2), 512, 256), ((2, 1), 64, 32)])
def test_convT2d(gpu_device, stride, Cin, Cout):
    _lib.init()
    g = torch.Generator().manual_seed(Cin)
    B, H, W = 2, 9, 11
    x, w, b = rnd(g, B, Cin, H, W), rnd(g, Cin, Cout, *stride) / Cin ** 0.5, rnd(g, Cout)
    src = G.FMap(B, H, W, Cin, 2, 1).alloc(gpu_device)
    src.interior()[:] = x.permute(0, 2, 3, 1).half().to(gpu_device)
    dst = G.FMap(B, H * stride[0], W * stride[1], Cout + 8, 1, 1).alloc(gpu_device)
    p = G.plan_convT2d(src, dst, w, b, stride=stride, dst_c0=8, device=gpu_device)
    p.run(src.t, out_h=dst.t)
    ref = F.conv_transpose2d(x.half().float(), w.half().float(), b, stride=stride)
    close(dst.interior()[..., 8:].permute(0, 3, 1, 2), ref)
    assert (dst.interior()[..., :8] == 0).all()


@pytest.mark.parametrize("Cn,k,d,B,T", [(64, 3, 1, 2, 300), (64, 11, 5, 3, 1000), (32, 7, 3, 2, 777), (32, 11, 5, 1, 4096),
                                        (64, 7, 1, 1, 50), (32, 3, 5, 5, 129)])
def test_conv1d_line_tile(gpu_device, Cn, k, d, B, T):
    """csrc/conv_taps.hip (weights resident in LDS) against the CPU replay of its descriptor, incl. residual, the
    pre-activated second output and zeroed halo rows."""
    _lib.init()
    g = torch.Generator().manual_seed(Cn * k + d)
    src, dst = G.Map1D(B, T, Cn, 32), G.Map1D(B, T, Cn, 32)
    src.t = torch.zeros(B, src.P, Cn, dtype=torch.float16)
    src.interior()[:] = rnd(g, B, T, Cn).half()
    w, b = rnd(g, Cn, Cn, k) / (Cn * k) ** 0.5, rnd(g, Cn)
    res = rnd(g, B, src.P, Cn).half()
    pre = 0.3 if (k + d) % 2 else None                                      # LeakyReLU inside the line buffer on some cases
    p = G.plan_conv1d_taps(src, dst, w, b, dilation=d, act="leaky", slope=0.1, slope2=0.2, device=gpu_device, pre_slope=pre)
    xd, rd = src.t.to(gpu_device), res.to(gpu_device)
    o1 = torch.full((B, dst.P, Cn), float("nan"), dtype=torch.float16, device=gpu_device)
    o2 = torch.full_like(o1, float("nan"))
    p.run(xd, out_h=o1, resid=rd, out_h2=o2)
    torch.cuda.synchronize()
    ref = G.replay_taps_on_cpu(p, src.t, res).view(B, dst.P, Cn)
    close(o1, ref)
    close(o2, F.leaky_relu(ref, 0.2))
    assert (o1[:, :32] == 0).all() and (o1[:, 32 + T:] == 0).all()
    if pre is None:
        ref_gemm = G.plan_conv1d_same(src, dst, w, b, dilation=d, act="leaky", slope=0.1, slope2=0.2, device=gpu_device)
        o3 = torch.empty_like(o1)
        ref_gemm.run(xd, out_h=o3, resid=rd)
        close(o1, o3.cpu())


@pytest.mark.parametrize("tile", [G.TILE_128x256_W8, G.TILE_256x128_W8])
def test_alternative_tiles_are_bit_identical(gpu_device, tile):
    """Every tile variant walks K in the same order with fp32 accumulators, so all of them must reproduce the default
    128x128 tile bit for bit -- on a plain GEMM with M / N tails, a gathered Conv2d with two sources, and a grid-z
    batched ConvTranspose2d (the persistent kernel folds grid z and many tiles per workgroup into one K-step stream)."""
    _lib.init()
    g = torch.Generator().manual_seed(tile)
    dev = gpu_device
    # (1) linear: M = 70000 rows (M tail, > 256 tiles so persistent workgroups take several), N = 384, K = 192 (3 K-steps)
    M, K, N = 70000, 192, 384
    a = rnd(g, M + 1024, K).half().to(dev)
    p = G.plan_linear(M, rnd(g, N, K) / K ** 0.5, rnd(g, N), act="gelu", device=dev)
    o1, o2 = torch.empty(M, N, dtype=torch.float16, device=dev), torch.zeros(M, N, dtype=torch.float16, device=dev)
    r = rnd(g, M, N).to(dev)
    p.run(a, out_h=o1, resid=r)
    p.tile = tile
    p.run(a, out_h=o2, resid=r)
    assert torch.equal(o1, o2)
    # (2) conv2d, two concatenated sources, halo rows written as zeros
    B, H, W = 3, 40, 36
    s0, s1 = G.FMap(B, H, W, 64, 1, 1).alloc(dev), G.FMap(B, H, W, 32, 1, 1).alloc(dev)
    s0.interior()[:] = rnd(g, B, H, W, 64).half().to(dev)
    s1.interior()[:] = rnd(g, B, H, W, 32).half().to(dev)
    d1, d2 = G.FMap(B, H, W, 128, 2, 2).alloc(dev), G.FMap(B, H, W, 128, 2, 2).alloc(dev)
    pc = G.plan_conv2d([s0, s1], d1, rnd(g, 128, 96, 3, 3) * 0.05, rnd(g, 128), device=dev)
    pc.run(s0.t, s1.t, out_h=d1.t)
    pc.tile = tile
    d2.t.fill_(float("nan"))
    pc.run(s0.t, s1.t, out_h=d2.t)
    assert torch.equal(d1.t, d2.t)
    # (3) ConvTranspose2d kernel = stride (2, 2): grid z = 2, pixel-shuffle store
    src = G.FMap(B, 20, 18, 128, 1, 1).alloc(dev)
    src.interior()[:] = rnd(g, B, 20, 18, 128).half().to(dev)
    u1, u2 = G.FMap(B, 40, 36, 128, 1, 1).alloc(dev), G.FMap(B, 40, 36, 128, 1, 1).alloc(dev)
    pt = G.plan_convT2d(src, u1, rnd(g, 128, 128, 2, 2) * 0.05, rnd(g, 128), stride=(2, 2), device=dev)
    pt.run(src.t, out_h=u1.t)
    pt.tile = tile
    pt.run(src.t, out_h=u2.t)
    assert torch.equal(u1.t, u2.t)


@pytest.mark.parametrize("Cn,B,H,W,PH,PW", [(32, 2, 37, 50, 2, 1), (64, 3, 16, 16, 1, 1), (64, 1, 70, 33, 1, 2), (32, 1, 256, 196, 1, 1)])
def test_conv2d_line_tile(gpu_device, Cn, B, H, W, PH, PW):
    """csrc/conv_taps.hip 2-D variant (16 x 16 tiles, 18 x 18 line-buffer patch, weights in LDS) against torch's conv2d on
    the same fp16 operands, and bit-compatible geometry with the implicit-GEMM plan (partial edge tiles, halo untouched)."""
    _lib.init()
    g = torch.Generator().manual_seed(Cn + H)
    src, dst = G.FMap(B, H, W, Cn, PH, PW).alloc(gpu_device), G.FMap(B, H, W, Cn, PH, PW).alloc(gpu_device)
    x = rnd(g, B, Cn, H, W)
    src.interior()[:] = x.permute(0, 2, 3, 1).half().to(gpu_device)
    w, b = rnd(g, Cn, Cn, 3, 3) / (3 * Cn ** 0.5), rnd(g, Cn)
    assert G.taps2d_supported([src], dst, w)
    p = G.Taps2dPlan(src, dst, w, b, slope=0.2, device=gpu_device)
    p.run(src.t, out_h=dst.t)
    torch.cuda.synchronize()
    ref = F.leaky_relu(F.conv2d(x.half().float(), w.half().float(), b, padding=1), 0.2)
    close(dst.interior().permute(0, 3, 1, 2), ref)
    halo = dst.t.clone()
    halo[:, PH:PH + H, PW:PW + W] = 0
    assert (halo == 0).all()                                                  # nothing outside the interior is written
    d2 = G.FMap(B, H, W, Cn, PH, PW).alloc(gpu_device)
    G.plan_conv2d([src], d2, w, b, slope=0.2, device=gpu_device).run(src.t, out_h=d2.t)
    close(dst.t, d2.t.cpu())
    assert not G.taps2d_supported([src], G.FMap(B, H, W, Cn, PH + 1, PW), w)
    assert not G.taps2d_supported([src], dst, rnd(g, Cn, Cn, 3, 3), stride=(2, 2))


@pytest.mark.parametrize("Cn,k,d,B,T", [(32, 3, 1, 2, 500), (32, 7, 3, 1, 1000), (32, 11, 5, 3, 777), (64, 3, 5, 2, 300), (64, 3, 1, 1, 2049), (64, 7, 5, 2, 600)])
def test_fused_resblock_step(gpu_device, Cn, k, d, B, T):
    """csrc/resblock_pair.hip (x + conv2(lrelu(conv1(lrelu(x)))), intermediate map in LDS) against torch on the same fp16
    operands with the intermediate rounded to fp16 as the kernel stores it; halo rows come out zero."""
    _lib.init()
    g = torch.Generator().manual_seed(Cn + k + d)
    src, dst = G.Map1D(B, T, Cn, 32).alloc(gpu_device), G.Map1D(B, T, Cn, 32).alloc(gpu_device)
    x = rnd(g, B, Cn, T)
    src.interior()[:] = x.transpose(1, 2).half().to(gpu_device)
    w1, w2 = rnd(g, Cn, Cn, k) / (Cn * k) ** 0.5, rnd(g, Cn, Cn, k) / (Cn * k) ** 0.5
    b1, b2 = rnd(g, Cn) * 0.1, rnd(g, Cn) * 0.1
    assert G.resblock_pair_supported(src, dst, w1, w2, d)
    p = G.ResblockPairPlan(src, dst, w1, b1, w2, b2, dilation=d, slope=0.1, device=gpu_device)
    dst.t.fill_(float("nan"))
    p.run(src.t, out_h=dst.t)
    torch.cuda.synchronize()
    xh = x.half().float()
    a = F.leaky_relu(xh, 0.1).half().float()
    t = F.leaky_relu(F.conv1d(a, w1.half().float(), b1, padding=(k - 1) * d // 2, dilation=d), 0.1).half().float()
    ref = xh + F.conv1d(t, w2.half().float(), b2, padding=(k - 1) // 2)
    close(dst.interior().transpose(1, 2), ref, tol=3e-3)
    assert (dst.t[:, :32] == 0).all() and (dst.t[:, 32 + T:] == 0).all()
    assert not G.resblock_pair_supported(src, dst, rnd(g, 64, 64, 11), rnd(g, 64, 64, 11), 1) or Cn != 64    # 2 x 90 KB of weights


@pytest.mark.parametrize("Cg,G_,B,T", [(48, 16, 3, 199), (64, 4, 2, 249), (48, 2, 5, 17), (64, 2, 1, 256)])
def test_posconv_line_tile(gpu_device, Cg, G_, B, T):
    """advh_posconv_tile_f16 (clip rows staged in LDS, weights streamed) against torch's grouped Conv1d on the same fp16
    operands (modeling_wav2vec2.py:326-379: k = 128, padding 64, last frame dropped, GELU, residual add).
    Stated tolerance 1e-4 on values of O(1) (same fp16 operands on both sides, fp32 accumulation over K = 128 * Cg;
    measured 5e-6)."""
    import ctypes
    from addvisor_hip.embedder import PosconvDesc
    K, H = 128, Cg * G_
    g = torch.Generator().manual_seed(Cg + T)
    h = torch.randn(B, T, H, generator=g)
    w = torch.randn(H, Cg, K, generator=g) / (K * Cg) ** 0.5
    bias = torch.randn(H, generator=g) * 0.1
    lib = _lib.lib()
    _lib.init()
    assert lib.advh_posconv_tile_lds_bytes(Cg, T) > 0
    hd = h.to(gpu_device).contiguous()
    xg = torch.empty(G_, B, T + K, Cg, dtype=torch.float16, device=gpu_device)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.advh_posconv_gather(hd.data_ptr(), xg.data_ptr(), B, T, H, G_, K, K // 2, None, st), "gather")
    wt = w.view(G_, Cg, Cg, K).permute(0, 1, 3, 2).reshape(G_, Cg, K * Cg // 32, 32).permute(0, 2, 1, 3).contiguous().half().to(gpu_device)
    bd = bias.to(gpu_device)
    out = torch.empty_like(hd)
    d = PosconvDesc()
    d.xg, d.W, d.bias, d.resid, d.out = xg.data_ptr(), wt.data_ptr(), bd.data_ptr(), hd.data_ptr(), out.data_ptr()
    d.B, d.T, d.H, d.G, d.K = B, T, H, G_, K
    _lib.check(lib.advh_posconv_tile_f16(ctypes.byref(d), st), "advh_posconv_tile_f16")
    conv = F.conv1d(h.half().float().transpose(1, 2), w.half().float(), bias, padding=K // 2, groups=G_)[:, :, :T]
    ref = h + F.gelu(conv).transpose(1, 2)
    err = (out.cpu() - ref).abs().max().item()
    print(f"posconv line tile Cg={Cg} T={T}: max err {err:.2e}")
    assert err <= 1e-4
    assert lib.advh_posconv_tile_lds_bytes(40, T) == -1 and lib.advh_posconv_tile_lds_bytes(Cg, 300) == -1
