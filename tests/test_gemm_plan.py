"""CPU-only host-logic tests: every implicit-GEMM plan (K-chunk tables, strides, halo windows) is
replayed in numpy with the kernel's own addressing rule and compared with torch's convolutions."""
import numpy as np
import torch
import torch.nn.functional as F

from addvisor_hip import gemm as G

torch.manual_seed(0)


def rnd(*shape):
    return torch.randn(*shape)


def fill(f: G.FMap, x_nchw: torch.Tensor):
    f.t = torch.zeros((f.B, f.Hp, f.Wp, f.C), dtype=torch.float16)
    f.interior()[:] = x_nchw.permute(0, 2, 3, 1).to(torch.float16)
    return f


def test_linear_plan():
    M, K, N = 37, 64, 48
    a, w, b = rnd(M, K).half(), rnd(N, K), rnd(N)
    p = G.plan_linear(M, w, b, act="gelu")
    out = G.replay_on_cpu(p, a, None, M * N).view(M, N)
    ref = F.gelu(a.float() @ w.half().float().T + b)
    assert torch.allclose(out, ref, atol=1e-4)


def test_conv1d_channels_last_plan():
    B, Cin, Cout, k, s = 2, 16, 24, 3, 2
    L_in, L_out = 21, 10
    P_out, P_in = 12, 24
    x = rnd(B, Cin, L_in)
    w, b = rnd(Cout, Cin, k), rnd(Cout)
    xin = torch.zeros(B, P_in, Cin, dtype=torch.float16)
    xin[:, :L_in] = x.transpose(1, 2).half()
    ref = F.gelu(F.conv1d(xin[:, :L_in].float().transpose(1, 2), w.half().float(), b, stride=s))
    p = G.plan_conv1d_cl(B, P_in, P_out, L_out, w, b, s)
    out = G.replay_on_cpu(p, xin, None, B * P_out * Cout).view(B, P_out, Cout)
    assert torch.allclose(out[:, :L_out], ref.transpose(1, 2), atol=1e-4)
    assert (out[:, L_out:] == 0).all()                      # filler rows are written as zeros
    pc = G.plan_conv1d_cl(B, P_in, P_out, L_out, w, b, s, compact_out=True)
    outc = G.replay_on_cpu(pc, xin, None, B * L_out * Cout).view(B, L_out, Cout)
    assert torch.allclose(outc, ref.transpose(1, 2), atol=1e-4)


def conv_case(Cins, Cout, k, stride, pad, dil, H, W, halo_in, halo_out, B=2):
    xs = [rnd(B, c, H, W) for c in Cins]
    w, b = rnd(Cout, sum(Cins), *k) * 0.2, rnd(Cout)
    srcs = [fill(G.FMap(B, H, W, c, *halo_in), x) for c, x in zip(Cins, xs)]
    ref = F.leaky_relu(F.conv2d(torch.cat([x.half().float() for x in xs], 1), w.half().float(), b, stride=stride,
                                padding=pad, dilation=dil), 0.2)
    Ho, Wo = ref.shape[2:]
    dst = G.FMap(B, Ho, Wo, Cout, *halo_out)
    p = G.plan_conv2d(srcs, dst, w, b, stride=stride, padding=pad, dilation=dil)
    out = G.replay_on_cpu(p, srcs[0].t, srcs[1].t if len(srcs) > 1 else None, B * dst.Hp * dst.Wp * Cout)
    out = out.view(B, dst.Hp, dst.Wp, Cout)
    inner = out[:, dst.PH:dst.PH + Ho, dst.PW:dst.PW + Wo].permute(0, 3, 1, 2)
    assert torch.allclose(inner, ref, atol=2e-3), (inner - ref).abs().max()
    halo = out.clone()
    halo[:, dst.PH:dst.PH + Ho, dst.PW:dst.PW + Wo] = 0
    assert not torch.isnan(out).any() and (halo == 0).all()   # the whole padded map is produced, halo = 0


def test_conv2d_plans():
    conv_case([8], 8, (3, 3), (1, 1), (1, 1), (1, 1), 6, 5, (1, 1), (1, 1))
    conv_case([8], 16, (5, 3), (2, 1), (2, 1), (1, 1), 8, 5, (2, 1), (1, 1))       # e1/e2-style stride (2,1)
    conv_case([16], 8, (3, 3), (2, 2), (1, 1), (1, 1), 8, 6, (1, 1), (2, 2))       # e3/e4-style stride 2
    conv_case([8], 8, (3, 3), (1, 1), (2, 2), (2, 2), 6, 7, (2, 2), (4, 4))        # dilated bottleneck
    conv_case([8], 8, (3, 3), (1, 1), (4, 4), (4, 4), 6, 7, (4, 4), (0, 0))
    conv_case([16, 8], 8, (3, 3), (1, 1), (1, 1), (1, 1), 4, 6, (1, 1), (1, 1))    # skip-concat by pointer
    conv_case([8], 8, (3, 3), (1, 1), (1, 1), (1, 1), 4, 4, (2, 3), (1, 1))        # halo wider than needed


def test_convT2d_plan():
    for stride in ((2, 2), (2, 1)):
        B, Cin, Cout, H, W = 2, 16, 8, 3, 4
        x, w, b = rnd(B, Cin, H, W), rnd(Cin, Cout, *stride) * 0.3, rnd(Cout)
        src = fill(G.FMap(B, H, W, Cin, 1, 2), x)
        ref = F.conv_transpose2d(x.half().float(), w.half().float(), b, stride=stride)
        dst = G.FMap(B, H * stride[0], W * stride[1], Cout + 8, 1, 1)               # written at channel offset 4
        p = G.plan_convT2d(src, dst, w, b, stride=stride, dst_c0=4)
        out = G.replay_on_cpu(p, src.t, None, B * dst.Hp * dst.Wp * dst.C).view(B, dst.Hp, dst.Wp, dst.C)
        inner = out[:, 1:1 + dst.H, 1:1 + dst.W, 4:4 + Cout].permute(0, 3, 1, 2)
        assert torch.allclose(inner, ref, atol=2e-3)
        out[:, 1:1 + dst.H, 1:1 + dst.W, 4:4 + Cout] = float("nan")
        assert torch.isnan(out).all()                          # nothing else is touched


def upconv_case(stride, Cb, Cu, Cs, N, Hc, Wc, where, B=2, cpad=0):
    """Fused ConvTranspose2d -> cat -> Conv2d 3x3 -> LeakyReLU (addvisor.py:45-46,69-71) vs the three torch ops."""
    sh, sw = stride
    xb, xs = rnd(B, Cb, Hc, Wc), rnd(B, Cs, Hc * sh, Wc * sw)
    wt, bt = rnd(Cb, Cu, sh, sw) * 0.3, rnd(Cu)
    wc, bc = rnd(N, Cu + Cs, 3, 3) * 0.2, rnd(N)
    up = F.conv_transpose2d(xb.half().float(), wt, bt, stride=stride)
    ref = F.leaky_relu(F.conv2d(torch.cat([up, xs.half().float()], 1), wc, bc, padding=1), 0.2)
    cC = G.round_up(Cb, 8) + (8 if where == "coarse" else 0) + cpad     # cpad: unused channels up to a 128-byte pixel pitch
    sC = G.round_up(Cs + (1 if where == "skip" else 0), 8)
    coarse = fill_part(G.FMap(B, Hc, Wc, cC, 1, 2), xb)
    skip = fill_part(G.FMap(B, Hc * sh, Wc * sw, sC, 2, 1), xs)
    ich = G.round_up(Cb, 8) if where == "coarse" else Cs
    G.add_indicator(coarse if where == "coarse" else skip, ich)
    dst = G.FMap(B, Hc * sh, Wc * sw, N, 1, 1)
    grp = G.plan_upconv2d(coarse, skip, dst, wt, bt, wc, bc, stride=stride, coarse_C=Cb, skip_C=Cs, indicator=(where, ich))
    assert len(grp.plans) == 1 and grp.plans[0].desc.nz == sh * sw and grp.plans[0].desc.z_inner == 1
    out = torch.zeros(B * dst.Hp * dst.Wp * N)
    for p in grp.plans:
        out = G.replay_on_cpu(p, coarse.t, skip.t, out.numel(), out_init=out)
    out = out.view(B, dst.Hp, dst.Wp, N)
    inner = out[:, 1:1 + dst.H, 1:1 + dst.W].permute(0, 3, 1, 2)
    # composed weights are rounded to fp16 once (instead of wt, wc separately): compare at fp16 weight resolution
    assert torch.allclose(inner, ref, atol=6e-3), (inner - ref).abs().max()
    halo = out.clone()
    halo[:, 1:1 + dst.H, 1:1 + dst.W] = 0
    assert (halo == 0).all()                                   # only the interior is written
    return (inner - ref).abs().max().item()


def fill_part(f: G.FMap, x_nchw: torch.Tensor):
    f.t = torch.zeros((f.B, f.Hp, f.Wp, f.C), dtype=torch.float16)
    f.interior()[..., :x_nchw.shape[1]] = x_nchw.permute(0, 2, 3, 1).to(torch.float16)
    return f


def test_upconv2d_plans():
    errs = [upconv_case((2, 2), 16, 8, 8, 8, 3, 4, "coarse"),      # up4 + d4 / up3 + d3 style
            upconv_case((2, 1), 16, 8, 8, 16, 3, 5, "coarse"),     # up2 + d2: stride (2, 1), 3 column taps
            upconv_case((2, 1), 8, 8, 1, 8, 4, 3, "skip"),         # up1 + d1: 1-channel skip (the spectrogram) carries the indicator
            upconv_case((2, 2), 8, 16, 16, 8, 1, 1, "coarse"),     # single coarse pixel: every tap hits a border
            upconv_case((2, 2), 16, 8, 8, 8, 2, 3, "coarse", cpad=40)]   # 24 used channels in a 64-channel (128-byte) pitch
    assert max(errs) < 6e-3                                    # measured 1.2e-3 ... 1.9e-3


def test_conv1d_same_and_convT1d_plans():
    """HiFi-GAN layers: dilated "same" Conv1d and phase-decomposed ConvTranspose1d (k = 2*stride)."""
    B, T, C = 2, 13, 16
    x = rnd(B, C, T)
    src = G.Map1D(B, T, C, 6)
    src.t = torch.zeros(B, src.P, C, dtype=torch.float16)
    src.interior()[:] = x.transpose(1, 2).half()
    for k, d in ((3, 1), (7, 1), (3, 5), (5, 3)):
        w, b = rnd(8, C, k) * 0.2, rnd(8)
        dst = G.Map1D(B, T, 8, 2)
        p = G.plan_conv1d_same(src, dst, w, b, dilation=d, act="leaky", slope=0.1)
        out = G.replay_on_cpu(p, src.t, None, B * dst.P * 8).view(B, dst.P, 8)
        ref = F.leaky_relu(F.conv1d(x.half().float(), w.half().float(), b, padding=(k - 1) * d // 2, dilation=d), 0.1)
        assert torch.allclose(out[:, 2:2 + T].transpose(1, 2), ref, atol=2e-3)
        assert (out[:, :2] == 0).all() and (out[:, 2 + T:] == 0).all()
    for r in (2, 8):
        w, b = rnd(C, 8, 2 * r) * 0.2, rnd(8)
        dst = G.Map1D(B, T * r, 8, 5)
        init = torch.zeros(B * dst.P * 8)
        p = G.plan_convT1d(src, dst, w, b, stride=r)
        out = G.replay_on_cpu(p, src.t, None, B * dst.P * 8, out_init=init).view(B, dst.P, 8)
        ref = F.conv_transpose1d(x.half().float(), w.half().float(), b, stride=r, padding=r // 2)
        assert torch.allclose(out[:, 5:5 + T * r].transpose(1, 2), ref, atol=2e-3), (out[:, 5:5 + T * r].transpose(1, 2) - ref).abs().max()
        assert (out[:, :5] == 0).all() and (out[:, 5 + T * r:] == 0).all()       # the halo stays untouched


def test_lds_swizzles_are_conflict_free():
    """Bank-conflict check by enumeration of the ds_read_b128 lane groups (MI355X_MICROARCH.md §LDS) for the two
    tile layouts of csrc/gemm.hip: 128-byte rows with chunk ^= row & 7, 64-byte rows with chunk ^= (row >> 1) & 2."""
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
              [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59], [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63]]
    for kk in (0, 1):                                   # BK = 64: rows of 8 chunks
        for g in groups:
            slots = {((l & 15) * 128 + (((kk * 4 + (l >> 4)) ^ ((l & 15) & 7)) * 16)) % 256 // 16 for l in g}
            assert len(slots) == 16
    for g in groups:                                    # BK = 32: rows of 4 chunks
        slots = {((l & 15) * 64 + (((l >> 4) ^ (((l & 15) >> 1) & 2)) * 16)) % 256 // 16 for l in g}
        assert len(slots) == 16


def test_conv1d_taps_plan():
    """Weights-in-LDS line-tile plan (csrc/conv_taps.hip): same layer as plan_conv1d_same for C in {32, 64}."""
    B, T = 2, 37
    for C_, k, d in ((32, 3, 1), (64, 7, 3), (32, 11, 5), (64, 11, 5), (64, 3, 5)):
        x = rnd(B, C_, T)
        src, dst = G.Map1D(B, T, C_, 32), G.Map1D(B, T, C_, 32)
        src.t = torch.zeros(B, src.P, C_, dtype=torch.float16)
        src.interior()[:] = x.transpose(1, 2).half()
        w, b = rnd(C_, C_, k) * 0.1, rnd(C_)
        assert G.taps_supported(src, dst, w, d)
        p = G.plan_conv1d_taps(src, dst, w, b, dilation=d, act="leaky", slope=0.1)
        assert 0 < G.taps_lds_bytes(C_, k, (k - 1) * d) <= 160 * 1024
        res = rnd(B, src.P, C_).half()
        out = G.replay_taps_on_cpu(p, src.t, res).view(B, dst.P, C_)
        ref = F.leaky_relu(F.conv1d(x.half().float(), w.half().float(), b, padding=(k - 1) * d // 2, dilation=d), 0.1)
        ref = ref.transpose(1, 2) + res[:, 32:32 + T].float()
        assert torch.allclose(out[:, 32:32 + T], ref, atol=2e-3)
        assert (out[:, :32] == 0).all() and (out[:, 32 + T:] == 0).all()
    assert not G.taps_supported(G.Map1D(1, 8, 128, 32), G.Map1D(1, 8, 128, 32), rnd(128, 128, 3), 1)
    assert not G.taps_supported(G.Map1D(1, 8, 64, 32), G.Map1D(1, 8, 64, 16), rnd(64, 64, 3), 1)


def test_line_tile_swizzles_are_conflict_free():
    """conv_taps.hip reads 16 consecutive LDS rows starting at ANY row (tap offsets shift the base): the swizzles
    chunk ^= row & 7 (128-byte rows) and chunk ^= (row >> 1) & 2 (64-byte rows) stay conflict-free."""
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
              [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59], [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63]]
    for base in range(64):
        for ks in (0, 1):
            for g in groups:
                slots = {((base + (l & 15)) * 128 + (((ks * 4 + (l >> 4)) ^ ((base + (l & 15)) & 7)) * 16)) % 256 // 16 for l in g}
                assert len(slots) == 16
        for g in groups:
            slots = {((base + (l & 15)) * 64 + (((l >> 4) ^ (((base + (l & 15)) >> 1) & 2)) * 16)) % 256 // 16 for l in g}
            assert len(slots) == 16


def test_line_tile_host_rules_match_the_library():
    """The host copies of the LDS-footprint rules (which layers go to the line-tile / fused kernels) agree with the C
    entry points, and the eligibility predicates reject what the kernels do not take."""
    from addvisor_hip import _lib
    lib = _lib.lib()
    for Cn in (32, 64):
        for k in (3, 7, 11):
            for d in (1, 3, 5):
                assert G.taps_tile(Cn, k, (k - 1) * d) == lib.advh_conv_taps_tile(Cn, k, (k - 1) * d)
                assert G.taps_lds_bytes(Cn, k, (k - 1) * d) == lib.advh_conv_taps_lds_bytes(Cn, k, (k - 1) * d)
                c_bytes = lib.advh_resblock_pair_lds_bytes(Cn, k, d)
                assert G.resblock_pair_lds_bytes(Cn, k, d) == c_bytes or c_bytes > 160 * 1024
    s32, d32 = G.Map1D(1, 100, 32, 32), G.Map1D(1, 100, 32, 32)
    assert G.resblock_pair_supported(s32, d32, rnd(32, 32, 11), rnd(32, 32, 11), 5)
    assert not G.resblock_pair_supported(G.Map1D(1, 100, 64, 32), G.Map1D(1, 100, 64, 32), rnd(64, 64, 11), rnd(64, 64, 11), 1)
    assert not G.resblock_pair_supported(s32, G.Map1D(1, 100, 32, 16), rnd(32, 32, 3), rnd(32, 32, 3), 1)
    a, b = G.FMap(2, 16, 16, 64, 1, 1), G.FMap(2, 16, 16, 64, 1, 1)
    assert G.taps2d_supported([a], b, rnd(64, 64, 3, 3))
    assert not G.taps2d_supported([a], b, rnd(64, 64, 3, 3), dilation=(2, 2))
    assert not G.taps2d_supported([a, a], b, rnd(64, 128, 3, 3))
    assert not G.taps2d_supported([G.FMap(2, 16, 16, 128, 1, 1)], G.FMap(2, 16, 16, 128, 1, 1), rnd(128, 128, 3, 3))


def test_split_format_roundtrip():
    """Host packer of the fp32-class operand format (csrc/device_math.h): x = hi + lo * 2^-11 to 2^-21 relative for
    |x| >= 2^-14 (hi normal), to 2^-25 absolute below (hi flushed, lo carries the value)."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(20000, dtype=torch.float64, generator=g) * torch.logspace(-7, 4, 20000, dtype=torch.float64)
    x = x[x.abs() < 60000]
    t = G.split_planes(x)
    assert t.dtype == torch.float16 and t.shape == (2,) + x.shape
    r = t[0].double() + t[1].double() / 2048
    big = x.abs() >= 2.0 ** -14
    assert ((r - x).abs() / x.abs())[big].max().item() <= 2.0 ** -21
    assert (r - x).abs()[~big].max().item() <= 2.0 ** -25
    assert (t[0][~big] == 0).all()                       # no fp16 subnormal ever reaches the matrix cores through hi
    assert torch.equal(G.join_planes(t), r.float())


def test_super_column_rule():
    """Tile order of the Linear layers (gemm.super_columns): a super-column keeps `sc` weight-column tiles L2-resident while all
    row tiles pass; it is switched off when fewer than 4 (fp16) / 3 (fp32-class) column tiles fit the 1.6 MB budget, because the
    activations are then re-read too often (FFN2, K = 3072: 301 TFLOP/s at sc = 1, 362 with super-columns off)."""
    M = 38208
    assert G.super_columns(768, 768, M) == 0                          # whole weight under 3 MB: resident anyway
    assert G.super_columns(2304, 768, M) == 8                         # QKV fp16: 128 x 768 x 2 B = 196 608 B per column tile
    assert G.super_columns(2304, 768, M, split=True) == 4             # same in the split format (4 B per element)
    assert G.super_columns(3072, 768, M, split=True) == 4             # FFN1
    assert G.super_columns(768, 3072, M, split=True) == 0             # FFN2: 1.5 MB per column tile -> off
    assert G.super_columns(768, 3072, M) == 0                         # fp16: two tiles fit, fewer than four -> off
    assert G.super_columns(64, 4608, M, split=True) == 0              # a single column tile
