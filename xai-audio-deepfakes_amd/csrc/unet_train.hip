// Training-mode kernels of the U-Net mask decoder (addvisor.py:12-84 under train_addvisor.py:364-378; SURVEY.md §8(f)
// rank 1): batch-statistics BatchNorm forward and backward on zero-haloed channels-last fp16 maps, the operand
// transposes of the weight-gradient GEMM, and the 1x1 sigmoid head's backward.  All HBM-bound row kernels; the
// contractions themselves (forward conv, dgrad, wgrad) are advh_gemm_f16 launches planned in addvisor_hip/unet_train.py.
//
// A map is [B][Hp][Wp][C] fp16 with the interior window rows h in [PH, PH+H), w in [PW, PW+W); halo elements are zero
// and are never written here.  Reductions over positions are two-stage and deterministic: NPART workgroups write
// per-channel partial sums, a second kernel adds them in a fixed order in fp64.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include "addvisor_hip.h"
#include "common.h"
#include "device_math.h"

namespace advh {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int NPART = 1024;

struct MapGeom { int B, Hp, Wp, C, PH, PW, H, W; };

__device__ __forceinline__ bool interior(const MapGeom& g, long row, int& b, int& h, int& w) {
    w = (int)(row % g.Wp);
    long r = row / g.Wp;
    h = (int)(r % g.Hp);
    b = (int)(r / g.Hp);
    return h >= g.PH && h < g.PH + g.H && w >= g.PW && w < g.PW + g.W;
}

// MODE 0: (sum z, sum z^2).  MODE 1 (BatchNorm backward): with y = scale*z + shift, dy^ = g * lrelu'(y),
// z^ = (z - mean) * invstd: (sum dy^, sum dy^ z^).  coef = [scale | shift | mean | invstd] (4 x C floats).
// One thread owns 8 channels of every (NPART * lanes_per_chunk)-th row.
// fp32, fp16 (lo == 0) or a split-format plane pair (lo = distance to the lo plane; device_math.h)
__device__ __forceinline__ void load_g8(const void* g, bool f32, long off, float (&o)[8], long lo = 0) {
    if (f32) {
        const float4 a = *(const float4*)((const float*)g + off), b = *(const float4*)((const float*)g + off + 4);
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    } else {
        load_h_rt<8>((const _Float16*)g, off, lo, o);
    }
}

// The incoming gradient g may be fp32: BatchNorm's backward subtracts its per-channel mean, so an fp16 g would lose
// exactly the part that survives (the dgrad GEMM accumulates in fp32 and can store fp32).
template <int MODE>
__global__ __launch_bounds__(256) void bn_partial_kernel(const _Float16* __restrict__ z, const void* __restrict__ g, bool g_f32,
                                                         const float* __restrict__ coef, float slope, MapGeom gm,
                                                         float* __restrict__ partial /*[NPART][2][C]*/, long z_lo, long g_lo) {
    __shared__ float red[256 * 16];
    const int CH = gm.C / 8, tid = threadIdx.x;
    const int ch = tid % CH, rl = tid / CH, RL = 256 / CH;       // CH in {4, 8, 16, 32, 64}
    const long rows = (long)gm.B * gm.Hp * gm.Wp;
    float s1[8], s2[8], sc[8], sh[8], mu[8], is[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        s1[j] = s2[j] = 0.f;
        if (MODE == 1) {
            const int c = ch * 8 + j;
            sc[j] = coef[c]; sh[j] = coef[gm.C + c]; mu[j] = coef[2 * gm.C + c]; is[j] = coef[3 * gm.C + c];
        }
    }
    if (rl < RL) {
        for (long row = (long)blockIdx.x * RL + rl; row < rows; row += (long)gridDim.x * RL) {
            int b, h, w;
            if (!interior(gm, row, b, h, w)) continue;
            float zv[8];
            load_h_rt<8>(z, row * gm.C + ch * 8, z_lo, zv);
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { float v = zv[j]; s1[j] += v; s2[j] = fmaf(v, v, s2[j]); }
            } else {
                float gv[8];
                load_g8(g, g_f32, row * gm.C + ch * 8, gv, g_lo);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float zz = zv[j], y = fmaf(sc[j], zz, sh[j]);
                    float d = gv[j] * (y > 0.f ? 1.f : slope);
                    s1[j] += d;
                    s2[j] = fmaf(d, (zz - mu[j]) * is[j], s2[j]);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[tid * 16 + j] = s1[j]; red[tid * 16 + 8 + j] = s2[j]; }
    __syncthreads();
    if (tid < CH) {                                             // fixed-order sum over the row lanes of this chunk
        for (int j = 0; j < 16; ++j) {
            float a = 0.f;
            for (int r = 0; r < RL; ++r) a += red[(r * CH + tid) * 16 + j];
            partial[((long)blockIdx.x * 2 + (j >> 3)) * gm.C + tid * 8 + (j & 7)] = a;
        }
    }
}

// second stage: one wavefront per output, lanes stride over the partials, fixed shuffle tree in fp64 (deterministic)
__global__ __launch_bounds__(256) void bn_reduce_kernel(const float* __restrict__ partial, int nparts, int C, float* __restrict__ sums) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;      // i in [0, 2C)
    if (i >= 2 * C) return;
    double a = 0.0;
    for (int p = lane; p < nparts; p += 64) a += (double)partial[(long)p * 2 * C + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if (lane == 0) sums[i] = (float)a;
}

// per-channel coefficients from the reduced sums (one thread per channel): replaces a dozen tiny tensor ops per layer.
// forward: coef = [scale | shift | mean | invstd], running statistics updated as nn.BatchNorm2d does in train() mode
// (momentum, unbiased variance).  backward: coef_b = [k1 | m1 | m2], dgamma = sum dy^ z^ / S, dbeta = sum dy^ / S.
__global__ void bn_coef_kernel(const float* __restrict__ sums, const float* __restrict__ gamma, const float* __restrict__ beta, int C,
                               float n, float eps, float momentum, float* __restrict__ rmean, float* __restrict__ rvar,
                               long* __restrict__ nbt, float* __restrict__ coef) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && nbt) *nbt += 1;
    if (c >= C) return;
    const float mean = sums[c] / n;
    const float var = fmaxf(sums[C + c] / n - mean * mean, 0.f);
    const float invstd = rsqrtf(var + eps), scale = gamma[c] * invstd;
    coef[c] = scale; coef[C + c] = beta[c] - mean * scale; coef[2 * C + c] = mean; coef[3 * C + c] = invstd;
    if (rmean) {
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * var * (n / fmaxf(n - 1.f, 1.f));
    }
}

__global__ void bn_bwd_coef_kernel(const float* __restrict__ sums, const float* __restrict__ coef, int C, float n, float inv_scale,
                                   float* __restrict__ coef_b, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    coef_b[c] = coef[c]; coef_b[C + c] = sums[c] / n; coef_b[2 * C + c] = sums[C + c] / n;
    dbeta[c] = sums[c] * inv_scale; dgamma[c] = sums[C + c] * inv_scale;
}

// forward: a = lrelu(scale*z + shift) on the interior.  backward: dz = k1 * (dy^ - m1 - z^ * m2), written into `dst`
// at element offset d_c0 + b*d_sB + (h-PH)*d_sH + (w-PW)*d_sW (dense map of the same or another halo, or the
// zero-upsampled grid a strided convolution's dgrad / wgrad read).  coef_b = [k1 | m1 | m2] (3 x C floats).
template <int BWD>
__global__ __launch_bounds__(256) void bn_apply_kernel(const _Float16* __restrict__ z, const void* __restrict__ g, bool g_f32,
                                                       const float* __restrict__ coef, const float* __restrict__ coef_b,
                                                       float slope, MapGeom gm, _Float16* __restrict__ dst, long d_sB, long d_sH,
                                                       long d_sW, long d_c0, long z_lo, long g_lo, long d_lo) {
    const int CH = gm.C / 8;
    const long total = (long)gm.B * gm.Hp * gm.Wp * CH;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int ch = (int)(i % CH);
    const long row = i / CH;
    int b, h, w;
    if (!interior(gm, row, b, h, w)) return;
    float zv[8], o[8];
    load_h_rt<8>(z, row * gm.C + ch * 8, z_lo, zv);
    if (!BWD) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = ch * 8 + j;
            float y = fmaf(coef[c], zv[j], coef[gm.C + c]);
            o[j] = y > 0.f ? y : slope * y;
        }
    } else {
        float gv[8];
        load_g8(g, g_f32, row * gm.C + ch * 8, gv, g_lo);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = ch * 8 + j;
            float zz = zv[j], y = fmaf(coef[c], zz, coef[gm.C + c]);
            float d = gv[j] * (y > 0.f ? 1.f : slope);
            float zh = (zz - coef[2 * gm.C + c]) * coef[3 * gm.C + c];
            o[j] = coef_b[c] * (d - coef_b[gm.C + c] - zh * coef_b[2 * gm.C + c]);
        }
    }
    store_h_rt<8>(dst, d_c0 + (long)b * d_sB + (long)(h - gm.PH) * d_sH + (long)(w - gm.PW) * d_sW + ch * 8, d_lo, o);
}

// Operand transpose of the weight-gradient GEMM.  dst[(t*nC + c)][col0 + p] = src[b][PHs + y][PWs + x][c0 + c] with
// p = (b*Hg + hg)*Wg + wg enumerating the COMMON grid of the layer, (y, x) = (sy*(hg-GH) + oy[t], sx*(wg-GW) + ox[t]);
// zero where (y, x) falls outside the source interior (the source's zero padding, grid halo rows, pad columns).  One workgroup
// moves a 64-position x 8*NC8-channel slab through LDS so both sides are 16-byte / 128-byte accesses.
struct TrArgs {
    int B, Hg, Wg, GH, GW, H, W;          // common grid and its interior
    int Hs, Ws, PHs, PWs, Cs, c0, nC;     // source map interior size, halo, channel count, channel slice
    int sy, sx, ntap, oy[16], ox[16];
    long ld, col0;                        // dst row pitch (elements) and first column
    int rpt, r0;                          // dst row of (tap t, channel c) = t*rpt + r0 + c
    int fuse;                             // all taps by one workgroup
};

__global__ __launch_bounds__(256) void transpose_gather_kernel(const _Float16* __restrict__ src, _Float16* __restrict__ dst, TrArgs a) {
    // [position][channel] tile, 144-byte rows; the 16-byte chunk ck of row r sits at slot ck ^ ((r >> 3) & 7) so that the
    // transposing 4-byte reads below (8 rows of one 8-row group x 4 channel pairs per 32-lane half) hit 32 distinct banks
    __shared__ __attribute__((aligned(16))) _Float16 tile[64][72];
    const long M = (long)a.B * a.Hg * a.Wg;
    const long p0 = (long)blockIdx.x * 64;
    const int cb = blockIdx.z * 64;                             // first channel of this 64-channel slab
    const int tid = threadIdx.x;
    const int nch = min(64, a.nC - cb);
    // taps: one per workgroup (grid y), or -- a.fuse: unit stride, the taps of a layer differ by a few positions -- all of them in
    // turn by ONE workgroup, so that taps 2.. re-read the slab from the CU's L1 instead of HBM (as separate workgroups they ran
    // M / 64 workgroups apart: three HBM reads of the map; profiles/r03_train_f32_kernel_summary.txt)
    const int t0 = a.fuse ? 0 : blockIdx.y, t1 = a.fuse ? a.ntap : t0 + 1;
    for (int t = t0; t < t1; ++t) {
    if (t > t0) __syncthreads();                                // the previous tap's scatter is done with the tile
    for (int i = tid; i < 64 * 8; i += 256) {                   // gather: 8 lanes read the 128 contiguous bytes of one position
        const int pl = i >> 3, ck = i & 7;
        f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        const long p = p0 + pl;
        if (p < M && ck * 8 < nch) {
            const int wg = (int)(p % a.Wg);
            const long r = p / a.Wg;
            const int hg = (int)(r % a.Hg), b = (int)(r / a.Hg);
            const int y = a.sy * (hg - a.GH) + a.oy[t], x = a.sx * (wg - a.GW) + a.ox[t];
            if (y >= 0 && y < a.Hs && x >= 0 && x < a.Ws) {
                const _Float16* s = src + (((long)b * (a.Hs + 2 * a.PHs) + y + a.PHs) * (a.Ws + 2 * a.PWs) + x + a.PWs) * a.Cs + a.c0 + cb + ck * 8;
                if (nch - ck * 8 >= 8) v = *(const f16x8*)s;
                else for (int j = 0; j < nch - ck * 8; ++j) v[j] = s[j];
            }
        }
        *(f16x8*)&tile[pl][(ck ^ ((pl >> 3) & 7)) * 8] = v;
    }
    __syncthreads();
    // scatter: lane = (channel pair within a quad, 8-position group, quad); 8 dword reads -> two 16-byte stores; a wave
    // writes 128 contiguous bytes (64 positions) of 16 channel rows
    const int lane = tid & 63, wv = tid >> 6;
    const int pk = (lane >> 2) & 7, cq = wv * 2 + (lane >> 5), c2 = cq * 4 + (lane & 3);
    unsigned w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = *(const unsigned*)&tile[pk * 8 + j][((cq ^ pk) * 4 + (lane & 3)) * 2];
    const long col = p0 + pk * 8;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int c = 2 * c2 + e;
        if (c >= nch) continue;
        f16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned short h = e ? (unsigned short)(w[j] >> 16) : (unsigned short)(w[j] & 0xffffu);
            v[j] = __builtin_bit_cast(_Float16, h);
        }
        _Float16* d = dst + ((long)t * a.rpt + a.r0 + cb + c) * a.ld + a.col0 + col;
        if (col + 8 <= M) *(f16x8*)d = v;
        else for (int j = 0; j < 8 && col + j < M; ++j) d[j] = v[j];
    }
    }
}

// mask head backward (addvisor.py:57-60): dlogit = dmask * m (1 - m); d y1[b,h,w,c] = dlogit * w[c] (fp32, scaled by
// `scale`); dlogit itself is stored (fp32) for the weight / bias sums.
__global__ __launch_bounds__(256) void unet_head_bwd_kernel(const float* __restrict__ dmask, const float* __restrict__ mask,
                                                            const float* __restrict__ w32, float scale, long total,
                                                            float* __restrict__ dlogit, void* __restrict__ dy1, bool dy_f32, long dy_lo) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const float m = mask[i], d = dmask[i] * m * (1.f - m);
    dlogit[i] = d;
    const float ds = d * scale;
    if (dy_f32) {
#pragma unroll
        for (int c4 = 0; c4 < 8; ++c4)
            *(float4*)((float*)dy1 + i * 32 + c4 * 4) = make_float4(ds * w32[c4 * 4], ds * w32[c4 * 4 + 1], ds * w32[c4 * 4 + 2], ds * w32[c4 * 4 + 3]);
    } else {
#pragma unroll
        for (int c8 = 0; c8 < 4; ++c8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = ds * w32[c8 * 8 + j];
            store_h_rt<8>((_Float16*)dy1, i * 32 + c8 * 8, dy_lo, v);
        }
    }
}

// weight / bias gradient of the 1x1 mask head: dw[c] = sum_i dlogit[i] * y1[i][c] (c < 32), dw[32] = sum_i dlogit[i].
// Thread = (8-channel chunk, row lane); partial [NPART][64] (33 used), reduced by bn_reduce_kernel with C = 32.
__global__ __launch_bounds__(256) void head_wgrad_kernel(const float* __restrict__ dlogit, const _Float16* __restrict__ y1, long total,
                                                         float* __restrict__ partial, long y_lo) {
    __shared__ float red[64][36];
    const int ch = threadIdx.x & 3, rl = threadIdx.x >> 2;
    float acc[8], sd = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (long i = (long)blockIdx.x * 64 + rl; i < total; i += (long)gridDim.x * 64) {
        const float d = dlogit[i];
        float v[8];
        load_h_rt<8>(y1, i * 32 + ch * 8, y_lo, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fmaf(d, v[j], acc[j]);
        if (ch == 0) sd += d;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[rl][ch * 8 + j] = acc[j];
    if (ch == 0) red[rl][32] = sd;
    __syncthreads();
    if (threadIdx.x < 64) {
        float a = 0.f;
        if (threadIdx.x < 33) for (int r = 0; r < 64; ++r) a += red[r][threadIdx.x];
        partial[(long)blockIdx.x * 64 + threadIdx.x] = a;
    }
}

// weight gradient of the 1-channel stem e1.block.0 (Conv2d(1, 32, (5,3), stride (2,1), padding (2,1)), addvisor.py:31):
// dW[co][kh*3+kw] = sum_p dz[p][co] * mag[b][2ho+kh-2][w+kw-1].  Thread = (co, position lane); the 15 magnitudes of a
// position are the same for the 32 channel threads (broadcast loads).  Two-stage deterministic reduction.
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const _Float16* __restrict__ dz, const float* __restrict__ mag, int Fq, int Tq,
                                                         int B, int H, int W, int PH, int PW, float* __restrict__ partial /*[NPART][480]*/,
                                                         long dz_lo) {
    __shared__ float red[8][480];
    const int co = threadIdx.x & 31, pl = threadIdx.x >> 5, Ho = H / 2;
    const long total = (long)B * Ho * W;
    float acc[15];
#pragma unroll
    for (int k = 0; k < 15; ++k) acc[k] = 0.f;
    for (long i = (long)blockIdx.x * 8 + pl; i < total; i += (long)gridDim.x * 8) {
        const int w = (int)(i % W);
        const long r = i / W;
        const int ho = (int)(r % Ho), b = (int)(r / Ho);
        const long di = (((long)b * (Ho + 2 * PH) + ho + PH) * (W + 2 * PW) + w + PW) * 32 + co;
        const float d = dz_lo ? join_f32(dz[di], dz[di + dz_lo]) : (float)dz[di];
#pragma unroll
        for (int kh = 0; kh < 5; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int h = 2 * ho + kh - 2, ww = w + kw - 1;
                const float x = (h >= 0 && h < H && ww >= 0 && ww < W) ? mag[((long)b * Fq + h) * Tq + ww] : 0.f;
                acc[kh * 3 + kw] = fmaf(d, x, acc[kh * 3 + kw]);
            }
    }
#pragma unroll
    for (int k = 0; k < 15; ++k) red[pl][co * 15 + k] = acc[k];
    __syncthreads();
    for (int i = threadIdx.x; i < 480; i += 256) {
        float a = 0.f;
        for (int r = 0; r < 8; ++r) a += red[r][i];
        partial[(long)blockIdx.x * 480 + i] = a;
    }
}

// weight gradient of the ONE skip channel of d1.block.0 (Conv2d(33, 32, 3, padding 1) on torch.cat([up1(y2), x], 1), addvisor.py:57-60, 79):
// dW[co][kh*3+kw] = sum_p dz[p][co] * mag[b][h+kh-1][w+kw-1] -- the stem kernel's scheme (thread = (co, position lane), broadcast loads of
// the magnitudes) for a 3x3 stride-1 window.  The other 32 input channels of that layer go through the LDS-tile kernel of conv_wgrad.hip;
// before this kernel the 40-channel concat map kept the whole layer on operand transposes + a split-K GEMM (4.8 ms of the training step).
__global__ __launch_bounds__(256) void skip_wgrad_kernel(const _Float16* __restrict__ dz, const float* __restrict__ mag, int Fq, int Tq, int B, int H,
                                                         int W, int PH, int PW, float* __restrict__ partial /*[NPART][288]*/, long dz_lo) {
    __shared__ float red[8][288];
    const int co = threadIdx.x & 31, pl = threadIdx.x >> 5;
    const long total = (long)B * H * W;
    float acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.f;
    for (long i = (long)blockIdx.x * 8 + pl; i < total; i += (long)gridDim.x * 8) {
        const int w = (int)(i % W);
        const long r = i / W;
        const int h = (int)(r % H), b = (int)(r / H);
        const long di = (((long)b * (H + 2 * PH) + h + PH) * (W + 2 * PW) + w + PW) * 32 + co;
        const float d = dz_lo ? join_f32(dz[di], dz[di + dz_lo]) : (float)dz[di];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int hh = h + kh - 1, ww = w + kw - 1;
                const float x = (hh >= 0 && hh < H && ww >= 0 && ww < W) ? mag[((long)b * Fq + hh) * Tq + ww] : 0.f;
                acc[kh * 3 + kw] = fmaf(d, x, acc[kh * 3 + kw]);
            }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) red[pl][co * 9 + k] = acc[k];
    __syncthreads();
    for (int i = threadIdx.x; i < 288; i += 256) {
        float a = 0.f;
        for (int r = 0; r < 8; ++r) a += red[r][i];
        partial[(long)blockIdx.x * 288 + i] = a;
    }
}

}  // namespace advh

using namespace advh;

static int head_wgrad_launch(const float* dlogit, const void* y1, long y_lo, int64_t total, float* partial, float* dw33, advh_stream_t stream) {
    if (!dlogit || !y1 || !partial || !dw33 || total <= 0) return ADVH_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(head_wgrad_kernel, dim3(NPART), dim3(256), 0, s, dlogit, (const _Float16*)y1, (long)total, partial, y_lo);
    hipLaunchKernelGGL(bn_reduce_kernel, dim3(16), dim3(256), 0, s, partial, NPART, 32, dw33);    // 64 outputs, 33 used
    return ADVH_LAUNCH_CHECK();
}
extern "C" int advh_unet_head_wgrad(const float* dlogit, const void* y1, int64_t total, float* partial, float* dw33,
                                    advh_stream_t stream) {
    return head_wgrad_launch(dlogit, y1, 0, total, partial, dw33, stream);
}
extern "C" int advh_unet_head_wgrad_split(const float* dlogit, const void* y1, int64_t y_lo, int64_t total, float* partial, float* dw33,
                                          advh_stream_t stream) {
    if (y_lo <= 0 || y_lo % 8) return ADVH_EINVAL;
    return head_wgrad_launch(dlogit, y1, y_lo, total, partial, dw33, stream);
}

static int stem_wgrad_launch(const void* dz, long dz_lo, int Fq, int Tq, int B, int H, int W, const float* mag, int PH, int PW,
                             float* partial, float* dw, advh_stream_t stream) {
    if (!dz || !mag || !partial || !dw || B <= 0 || H <= 0 || (H & 1) || W <= 0 || H > Fq || W > Tq || PH < 0 || PW < 0) return ADVH_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(stem_wgrad_kernel, dim3(NPART), dim3(256), 0, s, (const _Float16*)dz, mag, Fq, Tq, B, H, W, PH, PW, partial, dz_lo);
    hipLaunchKernelGGL(bn_reduce_kernel, dim3(120), dim3(256), 0, s, partial, NPART, 240, dw);
    return ADVH_LAUNCH_CHECK();
}
extern "C" int advh_unet_stem_wgrad(const void* dz, int Fq, int Tq, int B, int H, int W, const float* mag, int PH, int PW,
                                    float* partial, float* dw, advh_stream_t stream) {
    return stem_wgrad_launch(dz, 0, Fq, Tq, B, H, W, mag, PH, PW, partial, dw, stream);
}
extern "C" int advh_unet_stem_wgrad_split(const void* dz, int64_t dz_lo, int Fq, int Tq, int B, int H, int W, const float* mag, int PH,
                                          int PW, float* partial, float* dw, advh_stream_t stream) {
    if (dz_lo <= 0) return ADVH_EINVAL;
    return stem_wgrad_launch(dz, dz_lo, Fq, Tq, B, H, W, mag, PH, PW, partial, dw, stream);
}

static int skip_wgrad_launch(const void* dz, long dz_lo, int Fq, int Tq, int B, int H, int W, const float* mag, int PH, int PW,
                             float* partial, float* dw, advh_stream_t stream) {
    if (!dz || !mag || !partial || !dw || B <= 0 || H <= 0 || W <= 0 || H > Fq || W > Tq || PH < 0 || PW < 0) return ADVH_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(skip_wgrad_kernel, dim3(NPART), dim3(256), 0, s, (const _Float16*)dz, mag, Fq, Tq, B, H, W, PH, PW, partial, dz_lo);
    hipLaunchKernelGGL(bn_reduce_kernel, dim3(72), dim3(256), 0, s, partial, NPART, 144, dw);
    return ADVH_LAUNCH_CHECK();
}
extern "C" int advh_unet_skip_wgrad(const void* dz, int Fq, int Tq, int B, int H, int W, const float* mag, int PH, int PW, float* partial,
                                    float* dw, advh_stream_t stream) {
    return skip_wgrad_launch(dz, 0, Fq, Tq, B, H, W, mag, PH, PW, partial, dw, stream);
}
extern "C" int advh_unet_skip_wgrad_split(const void* dz, int64_t dz_lo, int Fq, int Tq, int B, int H, int W, const float* mag, int PH,
                                          int PW, float* partial, float* dw, advh_stream_t stream) {
    if (dz_lo <= 0) return ADVH_EINVAL;
    return skip_wgrad_launch(dz, dz_lo, Fq, Tq, B, H, W, mag, PH, PW, partial, dw, stream);
}

static bool geom_ok(const advh_map_geom* g) {
    return g && g->B > 0 && g->H > 0 && g->W > 0 && g->PH >= 0 && g->PW >= 0 && g->C >= 32 && g->C <= 512 && (g->C & (g->C - 1)) == 0;
}
static MapGeom mk(const advh_map_geom* g) { return MapGeom{g->B, g->H + 2 * g->PH, g->W + 2 * g->PW, g->C, g->PH, g->PW, g->H, g->W}; }

extern "C" int advh_bn_partial_count(void) { return NPART; }

static int bn_stats_launch(const void* z, long z_lo, const advh_map_geom* g, float* partial, float* sums, advh_stream_t stream) {
    if (!z || !partial || !sums || !geom_ok(g)) return ADVH_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_partial_kernel<0>, dim3(NPART), dim3(256), 0, s, (const _Float16*)z, (const void*)nullptr, false,
                       (const float*)nullptr, 0.f, mk(g), partial, z_lo, 0L);
    hipLaunchKernelGGL(bn_reduce_kernel, dim3((2 * g->C + 3) / 4), dim3(256), 0, s, partial, NPART, g->C, sums);
    return ADVH_LAUNCH_CHECK();
}
extern "C" int advh_bn_stats(const void* z, const advh_map_geom* g, float* partial, float* sums, advh_stream_t stream) {
    return bn_stats_launch(z, 0, g, partial, sums, stream);
}
extern "C" int advh_bn_stats_split(const void* z, int64_t z_lo, const advh_map_geom* g, float* partial, float* sums, advh_stream_t stream) {
    if (z_lo <= 0 || z_lo % 8) return ADVH_EINVAL;
    return bn_stats_launch(z, z_lo, g, partial, sums, stream);
}

extern "C" int advh_bn_coef(const float* sums, const float* gamma, const float* beta, int C, float n, float eps, float momentum,
                            float* running_mean, float* running_var, int64_t* num_batches_tracked, float* coef, advh_stream_t stream) {
    if (!sums || !gamma || !beta || !coef || C <= 0 || n <= 0.f || (!running_mean) != (!running_var)) return ADVH_EINVAL;
    hipLaunchKernelGGL(bn_coef_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, gamma, beta, C, n, eps, momentum,
                       running_mean, running_var, (long*)num_batches_tracked, coef);
    return ADVH_LAUNCH_CHECK();
}

extern "C" int advh_bn_bwd_coef(const float* sums, const float* coef, int C, float n, float inv_scale, float* coef_b, float* dgamma,
                                float* dbeta, advh_stream_t stream) {
    if (!sums || !coef || !coef_b || !dgamma || !dbeta || C <= 0 || n <= 0.f) return ADVH_EINVAL;
    hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, coef, C, n, inv_scale, coef_b,
                       dgamma, dbeta);
    return ADVH_LAUNCH_CHECK();
}

static int bn_apply_launch(const void* z, long z_lo, const advh_map_geom* g, const float* coef, float slope, void* a, long a_lo,
                           advh_stream_t stream) {
    if (!z || !a || !coef || !geom_ok(g)) return ADVH_EINVAL;
    MapGeom m = mk(g);
    const long total = (long)m.B * m.Hp * m.Wp * (m.C / 8);
    hipLaunchKernelGGL(bn_apply_kernel<0>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const _Float16*)z, (const void*)nullptr, false, coef, (const float*)nullptr, slope, m, (_Float16*)a,
                       (long)m.Hp * m.Wp * m.C, (long)m.Wp * m.C, (long)m.C, ((long)m.PH * m.Wp + m.PW) * m.C, z_lo, 0L, a_lo);
    return ADVH_LAUNCH_CHECK();
}
extern "C" int advh_bn_apply(const void* z, const advh_map_geom* g, const float* coef, float slope, void* a, advh_stream_t stream) {
    return bn_apply_launch(z, 0, g, coef, slope, a, 0, stream);
}
extern "C" int advh_bn_apply_split(const void* z, int64_t z_lo, const advh_map_geom* g, const float* coef, float slope, void* a,
                                   int64_t a_lo, advh_stream_t stream) {
    if (z_lo <= 0 || a_lo <= 0 || z_lo % 8 || a_lo % 8) return ADVH_EINVAL;
    return bn_apply_launch(z, z_lo, g, coef, slope, a, a_lo, stream);
}

static int bn_bwd_sums_launch(const void* z, long z_lo, const void* g_a, int g_f32, long g_lo, const advh_map_geom* g, const float* coef,
                              float slope, float* partial, float* sums, advh_stream_t stream) {
    if (!z || !g_a || !coef || !partial || !sums || !geom_ok(g)) return ADVH_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_partial_kernel<1>, dim3(NPART), dim3(256), 0, s, (const _Float16*)z, g_a, g_f32 != 0, coef, slope,
                       mk(g), partial, z_lo, g_lo);
    hipLaunchKernelGGL(bn_reduce_kernel, dim3((2 * g->C + 3) / 4), dim3(256), 0, s, partial, NPART, g->C, sums);
    return ADVH_LAUNCH_CHECK();
}
extern "C" int advh_bn_bwd_sums(const void* z, const void* g_a, int g_f32, const advh_map_geom* g, const float* coef, float slope,
                                float* partial, float* sums, advh_stream_t stream) {
    return bn_bwd_sums_launch(z, 0, g_a, g_f32, 0, g, coef, slope, partial, sums, stream);
}
extern "C" int advh_bn_bwd_sums_split(const void* z, int64_t z_lo, const void* g_a, int64_t g_lo, const advh_map_geom* g, const float* coef,
                                      float slope, float* partial, float* sums, advh_stream_t stream) {
    if (z_lo <= 0 || g_lo <= 0 || z_lo % 8 || g_lo % 8) return ADVH_EINVAL;
    return bn_bwd_sums_launch(z, z_lo, g_a, 0, g_lo, g, coef, slope, partial, sums, stream);
}

static int bn_bwd_apply_launch(const void* z, long z_lo, const void* g_a, int g_f32, long g_lo, const advh_map_geom* g, const float* coef,
                               const float* coef_b, float slope, void* dz, long dz_lo, int64_t d_sB, int64_t d_sH, int64_t d_sW,
                               int64_t d_c0, advh_stream_t stream) {
    if (!z || !g_a || !dz || !coef || !coef_b || !geom_ok(g)) return ADVH_EINVAL;
    if ((d_sB | d_sH | d_sW | d_c0) & 7) return ADVH_EINVAL;
    MapGeom m = mk(g);
    const long total = (long)m.B * m.Hp * m.Wp * (m.C / 8);
    hipLaunchKernelGGL(bn_apply_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const _Float16*)z, g_a, g_f32 != 0, coef, coef_b, slope, m, (_Float16*)dz, (long)d_sB, (long)d_sH,
                       (long)d_sW, (long)d_c0, z_lo, g_lo, dz_lo);
    return ADVH_LAUNCH_CHECK();
}
extern "C" int advh_bn_bwd_apply(const void* z, const void* g_a, int g_f32, const advh_map_geom* g, const float* coef, const float* coef_b,
                                 float slope, void* dz, int64_t d_sB, int64_t d_sH, int64_t d_sW, int64_t d_c0,
                                 advh_stream_t stream) {
    return bn_bwd_apply_launch(z, 0, g_a, g_f32, 0, g, coef, coef_b, slope, dz, 0, d_sB, d_sH, d_sW, d_c0, stream);
}
extern "C" int advh_bn_bwd_apply_split(const void* z, int64_t z_lo, const void* g_a, int64_t g_lo, const advh_map_geom* g, const float* coef,
                                       const float* coef_b, float slope, void* dz, int64_t dz_lo, int64_t d_sB, int64_t d_sH, int64_t d_sW,
                                       int64_t d_c0, advh_stream_t stream) {
    if (z_lo <= 0 || g_lo <= 0 || dz_lo <= 0 || z_lo % 8 || g_lo % 8 || dz_lo % 8) return ADVH_EINVAL;
    return bn_bwd_apply_launch(z, z_lo, g_a, 0, g_lo, g, coef, coef_b, slope, dz, dz_lo, d_sB, d_sH, d_sW, d_c0, stream);
}

extern "C" int advh_transpose_gather(const void* src, void* dst, const advh_transpose_desc* d, advh_stream_t stream) {
    if (!src || !dst || !d || d->B <= 0 || d->Hg <= 0 || d->Wg <= 0 || d->ntap <= 0 || d->ntap > 16 || d->nC <= 0 ||
        d->c0 < 0 || d->c0 + d->nC > d->Cs || d->sy <= 0 || d->sx <= 0 || (d->ld & 7) || (d->col0 & 7) || d->r0 < 0 ||
        d->rpt < d->r0 + d->nC)
        return ADVH_EINVAL;
    if (d->nC >= 8 && ((d->nC & 7) || (d->c0 & 7) || (d->Cs & 7))) return ADVH_EINVAL;
    TrArgs a;
    a.B = d->B; a.Hg = d->Hg; a.Wg = d->Wg; a.GH = d->GH; a.GW = d->GW; a.H = d->H; a.W = d->W;
    a.Hs = d->Hs; a.Ws = d->Ws; a.PHs = d->PHs; a.PWs = d->PWs; a.Cs = d->Cs; a.c0 = d->c0; a.nC = d->nC;
    a.sy = d->sy; a.sx = d->sx; a.ntap = d->ntap;
    for (int t = 0; t < 16; ++t) { a.oy[t] = d->oy[t]; a.ox[t] = d->ox[t]; }
    a.ld = d->ld; a.col0 = d->col0; a.rpt = d->rpt; a.r0 = d->r0;
    const long M = (long)d->B * d->Hg * d->Wg;
    if (d->col0 + ((M + 7) / 8) * 8 > d->ld) return ADVH_EINVAL;
    a.fuse = d->ntap > 1 && d->sx == 1 && d->sy == 1;
    for (int t = 1; t < d->ntap; ++t) a.fuse = a.fuse && d->oy[t] == d->oy[0] && abs(d->ox[t] - d->ox[0]) <= 8;
    dim3 grid((unsigned)((M + 63) / 64), a.fuse ? 1 : d->ntap, (d->nC + 63) / 64);
    hipLaunchKernelGGL(transpose_gather_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const _Float16*)src, (_Float16*)dst, a);
    return ADVH_LAUNCH_CHECK();
}

static int head_bwd_launch(const float* dmask, const float* mask, const float* w32, float scale, int64_t total, float* dlogit, void* dy1,
                           int dy_f32, long dy_lo, advh_stream_t stream) {
    if (!dmask || !mask || !w32 || !dlogit || !dy1 || total <= 0) return ADVH_EINVAL;
    hipLaunchKernelGGL(unet_head_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dmask,
                       mask, w32, scale, (long)total, dlogit, dy1, dy_f32 != 0, dy_lo);
    return ADVH_LAUNCH_CHECK();
}
extern "C" int advh_unet_head_bwd(const float* dmask, const float* mask, const float* w32, float scale, int64_t total,
                                  float* dlogit, void* dy1, int dy_f32, advh_stream_t stream) {
    return head_bwd_launch(dmask, mask, w32, scale, total, dlogit, dy1, dy_f32, 0, stream);
}
extern "C" int advh_unet_head_bwd_split(const float* dmask, const float* mask, const float* w32, float scale, int64_t total,
                                        float* dlogit, void* dy1, int64_t dy_lo, advh_stream_t stream) {
    if (dy_lo <= 0 || dy_lo % 8) return ADVH_EINVAL;
    return head_bwd_launch(dmask, mask, w32, scale, total, dlogit, dy1, 0, dy_lo, stream);
}

ADVH_SPLIT_FLAG_SETTER(advh_split_flag_unet_train)
