// Input-gradient (dgrad-only) kernels of the frozen embedder: what Captum's Saliency / InputXGradient /
// IntegratedGradients need from autograd (captum_saliency.py:116-118, 131-135).  Weights are frozen, so
// there is no wgrad anywhere; every dense backward product is a transposed-weight launch of the implicit
// GEMM (gemm.hip) and this file holds the row-wise / attention / waveform-end pieces.
//
// Gradients travel as fp16 between GEMMs (fp32 accumulate, fp32 residual-stream gradient) multiplied by a
// caller-chosen power-of-two loss scale, which is exact because the whole chain is linear in the gradient.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include "addvisor_hip.h"
#include "common.h"
#include "device_math.h"

namespace advh {

constexpr float LOG2E = 1.4426950408889634f;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// fp32 row, fp16 row (lo == 0) or a split-format plane pair (lo = distance to the lo plane, device_math.h)
template <bool F32>
__device__ __forceinline__ void load4(const void* p, long off, float (&v)[4], long lo = 0) {
    if (F32) {
        float4 t = *(const float4*)((const float*)p + off);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        load_h_rt<4>((const _Float16*)p, off, lo, v);
    }
}

// ---------------------------------------------------------------------------------------------- LayerNorm
// y = LN(x) * gamma + beta  [optionally followed by GELU]; given dy returns
//   dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = (dy [* GELU'(y_pre)]) * gamma
// then optionally  dx *= GELU'(dact_src)  and  dx += add.   One wavefront per row.
template <bool X32, bool DY32, int MAXV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const void* __restrict__ x, const void* __restrict__ dy,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            int gelu_fwd, const float* __restrict__ add,
                                                            const _Float16* __restrict__ dact_src, float* __restrict__ out_f,
                                                            _Float16* __restrict__ out_h, int M, int C, float eps,
                                                            int remap_T, int remap_P, long x_lo, long dy_lo, long dact_lo, long out_lo) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    // outputs may go to a per-clip padded layout: row (b*T + t) -> b*P + t
    const long orow = remap_P > 0 ? (row / remap_T) * remap_P + row % remap_T : row;
    float xv[MAXV][4], gv[MAXV][4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        int c = (i * 64 + lane) * 4;
        if (c < C) {
            load4<X32>(x, row * C + c, xv[i], x_lo);
            s += (xv[i][0] + xv[i][1]) + (xv[i][2] + xv[i][3]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) xv[i][r] = 0.f;
        }
    }
    const float mean = wsum(s) / C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        int c = (i * 64 + lane) * 4;
        if (c < C) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { float d = xv[i][r] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wsum(q) / C + eps);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        int c = (i * 64 + lane) * 4;
        if (c < C) {
            float d[4];
            load4<DY32>(dy, row * C + c, d, dy_lo);
            float4 g = *(const float4*)(gamma + c);
            float gm[4] = {g.x, g.y, g.z, g.w};
            float bt[4] = {0.f, 0.f, 0.f, 0.f};
            if (gelu_fwd) { float4 b = *(const float4*)(beta + c); bt[0] = b.x; bt[1] = b.y; bt[2] = b.z; bt[3] = b.w; }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float xh = (xv[i][r] - mean) * rstd;
                float gg = d[r];
                if (gelu_fwd) gg *= gelu_grad(xh * gm[r] + bt[r]);
                gg *= gm[r];
                gv[i][r] = gg;
                xv[i][r] = xh;
                sg += gg;
                sgx += gg * xh;
            }
        }
    }
    const float mg = wsum(sg) / C, mgx = wsum(sgx) / C;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        int c = (i * 64 + lane) * 4;
        if (c < C) {
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = rstd * (gv[i][r] - mg - xv[i][r] * mgx);
            if (dact_src) {
                float z[4];
                load_h_rt<4>(dact_src, row * C + c, dact_lo, z);
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] *= gelu_grad(z[r]);
            }
            if (add) {
                float4 a = *(const float4*)(add + row * C + c);
                o[0] += a.x; o[1] += a.y; o[2] += a.z; o[3] += a.w;
            }
            if (out_f) *(float4*)(out_f + orow * C + c) = make_float4(o[0], o[1], o[2], o[3]);
            if (out_h) store_h_rt<4>(out_h, orow * C + c, out_lo, o);
        }
    }
}

// ---------------------------------------------------------------------------------------------- attention
// dqkv from (qkv, dctx) for softmax(QK^T * scale) V, T <= 256.  One workgroup per (head, clip), two passes:
//   A (one wavefront per 16-query tile, scores transposed: key on the MFMA row): row max / sum / delta and
//     dQ^T = K^T dS^T (sums over the accumulator's ROW index: operands straight from registers);
//   B (one wavefront per 16-key tile, scores un-transposed: query on the MFMA row): recompute P and dS with the
//     saved row statistics; dV^T = dO^T P and dK^T = Q^T dS again sum over the accumulator's row index.
// Recomputing S and dP in both orientations costs 2x of a small op and needs no transposes or atomics.
// TR (head dim exactly 64): no transposed copies at all -- K, V (pass A) and Q, dO (pass B) are staged row-major by the
// LDS DMA and the three operands that need the transposed orientation (K^T for dQ, dO^T for dV, Q^T for dK) are read with
// ds_read_b64_tr_b16; the key / query order of its two 4-row reads (4g+q | 16+4g+q) is the order the dS / P registers
// already have.  LDS drops from 88 KB to 53 KB (+ statistics).
typedef __fp16 trvec __attribute__((__vector_size__(4 * sizeof(__fp16))));
#define ADVH_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define ADVH_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// 8 consecutive-in-k values of column (c0 + lane%16) from a row-major [rows][64] fp16 LDS tile whose 16-byte chunk c of
// row r sits at slot c ^ (r & 7); rows ra+q (elements 0-3) and rb+q (elements 4-7)
__device__ __forceinline__ f16x8 tr_frag64(const __attribute__((address_space(3))) char* base, int ra, int rb, int c0, int p) {
    const int ch = (c0 >> 3) + (p >> 1), sub = (p & 1) * 8;
    trvec lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) trvec*)(base + ((size_t)ra * 8 + (ch ^ (ra & 7))) * 16 + sub));
    trvec hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) trvec*)(base + ((size_t)rb * 8 + (ch ^ (rb & 7))) * 16 + sub));
    f16x8 r;
    __builtin_memcpy(&r, &lo, 8);
    __builtin_memcpy((char*)&r + 8, &hi, 8);
    return r;
}

template <int NT, int D, bool TR>
__global__ __launch_bounds__(256) void attention_bwd_kernel(const _Float16* __restrict__ qkv, const _Float16* __restrict__ dctx,
                                                            _Float16* __restrict__ dqkv, int T, int H, int dm, float scale) {
    constexpr int NKEY = NT * 16, CH = D / 8, VP = NKEY + 64, NS = (NT + 1) / 2, KK = D / 32, DT = D / 16;
    constexpr int ROWB = NKEY * D, TRB = TR ? 0 : D * VP;
    static_assert(!TR || D == 64, "the transposing-read path is for head dim 64");
    // LDS: pass A uses K | V (row-major, swizzled) | Kt ; pass B re-uses the space for Qt | dOt ; stats stay.
    // For D = 128 V is not staged (pass A reads its fragments from global memory) to stay inside 160 KiB.
    constexpr bool STAGE_V = D <= 64;
    constexpr int LDSH = TR ? 2 * ROWB
                            : ((STAGE_V ? 2 * ROWB + TRB : ROWB + TRB) > 2 * TRB ? (STAGE_V ? 2 * ROWB + TRB : ROWB + TRB) : 2 * TRB);
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    _Float16* Ks = (_Float16*)smem_raw;
    _Float16* Vs = Ks + ROWB;
    _Float16* Kt = STAGE_V ? Vs + ROWB : Vs;
    _Float16* Qt = (_Float16*)smem_raw;
    _Float16* dOt = Qt + TRB;
    float* rmax = (float*)(smem_raw + LDSH * 2);
    float* rinv = rmax + NKEY;
    float* rdel = rinv + NKEY;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int head = blockIdx.x, b = blockIdx.y;
    const long ld = 3L * H;
    const _Float16* base = qkv + (long)b * T * ld + head * dm;     // q at +0, k at +H, v at +2H
    const _Float16* dob = dctx + (long)b * T * H + head * dm;
    _Float16* dbase = dqkv + (long)b * T * ld + head * dm;
    const int chm = dm / 8;                                        // real 16-byte chunks per row (D - dm is zero padding)
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const int fr = lane & 15, g = lane >> 4;
    const int tq = (lane >> 2) & 3, tp = lane & 3;                 // this lane's (row, column quad) inside a transposing read
    const auto* smem3 = (const __attribute__((address_space(3))) char*)ADVH_LDS_PTR(smem_raw);

    if (TR) {
        for (int i = tid; i < NKEY * CH; i += 256) {               // rows >= T re-read row T-1: finite, and masked (p = 0) downstream
            const int key = i / CH, c = (i % CH) ^ (key & 7);
            const _Float16* src = base + (long)min(key, T - 1) * ld + c * 8;
            __builtin_amdgcn_global_load_lds(ADVH_GLOBAL_PTR(src + H), ADVH_LDS_PTR((char*)Ks + (size_t)(i - lane) * 16), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(ADVH_GLOBAL_PTR(src + 2 * H), ADVH_LDS_PTR((char*)Vs + (size_t)(i - lane) * 16), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else
    for (int i = tid; i < NKEY * CH; i += 256) {
        int key = i / CH, c = i % CH;
        f16x8 kv = {0, 0, 0, 0, 0, 0, 0, 0}, vv = kv;
        if (key < T && c < chm) {
            kv = *(const f16x8*)(base + (long)key * ld + H + c * 8);
            vv = *(const f16x8*)(base + (long)key * ld + 2 * H + c * 8);
        }
        int sw = ((c ^ (key & (CH - 1))) * 8);
        *(f16x8*)(Ks + key * D + sw) = kv;
        if (STAGE_V) *(f16x8*)(Vs + key * D + sw) = vv;
#pragma unroll
        for (int j = 0; j < 8; ++j) Kt[(c * 8 + j) * VP + (c & 7) * 8 + key] = kv[j];
    }
    __syncthreads();

    // ------------------------------------------------------------------ pass A: query tiles
    for (int qt = wv; qt * 16 < T; qt += 4) {
        int qrow = qt * 16 + fr;
        int qr = qrow < T ? qrow : T - 1;
        f16x8 qf[KK], of[KK];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            qf[kk] = (kk * 4 + g < chm) ? *(const f16x8*)(base + (long)qr * ld + kk * 32 + g * 8) : zero8;
            of[kk] = (kk * 4 + g < chm) ? *(const f16x8*)(dob + (long)qr * H + kk * 32 + g * 8) : zero8;
        }
        f32x4 s[NT], dp[NT];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
            dp[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                int key = kt * 16 + fr, c = kk * 4 + g;
                int sw = ((c ^ (key & (CH - 1))) * 8);
                f16x8 kf = *(const f16x8*)(Ks + key * D + sw);
                f16x8 vf;
                if (STAGE_V) vf = *(const f16x8*)(Vs + key * D + sw);
                else vf = (key < T && c < chm) ? *(const f16x8*)(base + (long)key * ld + 2 * H + c * 8) : zero8;
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[kk], s[kt], 0, 0, 0);      // S^T  [key][q]
                dp[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, of[kk], dp[kt], 0, 0, 0);    // dP^T [key][q]
            }
            __builtin_amdgcn_sched_barrier(0);            // keeps the fragment reads of all NT tiles from being hoisted en bloc (VGPRs)
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = (kt * 16 + g * 4 + r < T) ? s[kt][r] * (scale * LOG2E) : -INFINITY;   // log2 domain: exp(x) = v_exp_f32(x log2 e)
                s[kt][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { float e = __builtin_amdgcn_exp2f(s[kt][r] - mx); s[kt][r] = e; sum += e; }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.f / sum;
        float del = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { s[kt][r] *= inv; del += s[kt][r] * dp[kt][r]; }
        del += __shfl_xor(del, 16, 64);
        del += __shfl_xor(del, 32, 64);
        if (g == 0 && qrow < NKEY) { rmax[qrow] = mx; rinv[qrow] = inv; rdel[qrow] = del; }
        // dS^T as the B operand of dQ^T = K^T dS^T (k order inside a 32-key step permuted, as in the forward)
        f16x8 dsf[NS];
#pragma unroll
        for (int ss = 0; ss < NS; ++ss)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dsf[ss][r] = (_Float16)(s[2 * ss][r] * (dp[2 * ss][r] - del) * scale);
                dsf[ss][4 + r] = (2 * ss + 1 < NT) ? (_Float16)(s[(2 * ss + 1 < NT) ? 2 * ss + 1 : 0][r] *
                                                                (dp[(2 * ss + 1 < NT) ? 2 * ss + 1 : 0][r] - del) * scale)
                                                   : (_Float16)0.f;
            }
        f32x4 o[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ss = 0; ss < NS; ++ss) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                f16x8 kf;
                if (TR) {                                           // K^T from the row-major tile; a missing odd tile re-reads the even one (dS = 0)
                    const int ra = 32 * ss + 4 * g + tq;
                    kf = tr_frag64(smem3, ra, (2 * ss + 1 < NT) ? ra + 16 : ra, dt * 16, tp);
                } else {
                    const _Float16* kr = Kt + (dt * 16 + fr) * VP + (((dt * 16 + fr) >> 3) & 7) * 8 + ss * 32 + g * 4;
                    f16x4 lo = *(const f16x4*)kr;
                    f16x4 hi = (2 * ss + 1 < NT) ? *(const f16x4*)(kr + 16) : f16x4{0, 0, 0, 0};
                    kf = f16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, dsf[ss], o[dt], 0, 0, 0);     // dQ^T [d][q]
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (qrow < T) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                if (dt * 16 + g * 4 >= dm) continue;
                f16x4 hv = {(_Float16)o[dt][0], (_Float16)o[dt][1], (_Float16)o[dt][2], (_Float16)o[dt][3]};
                *(f16x4*)(dbase + (long)qrow * ld + dt * 16 + g * 4) = hv;
            }
        }
    }
    __syncthreads();

    // ------------------------------------------------------------------ pass B: key tiles
    if (TR) {
        for (int i = tid; i < NKEY * CH; i += 256) {               // Q | dO row-major over the K | V space
            const int row = i / CH, c = (i % CH) ^ (row & 7), rr = min(row, T - 1);
            __builtin_amdgcn_global_load_lds(ADVH_GLOBAL_PTR(base + (long)rr * ld + c * 8), ADVH_LDS_PTR((char*)Ks + (size_t)(i - lane) * 16), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(ADVH_GLOBAL_PTR(dob + (long)rr * H + c * 8), ADVH_LDS_PTR((char*)Vs + (size_t)(i - lane) * 16), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else
    for (int i = tid; i < NKEY * CH; i += 256) {
        int row = i / CH, c = i % CH;
        f16x8 qv = {0, 0, 0, 0, 0, 0, 0, 0}, ov = qv;
        if (row < T && c < chm) {
            qv = *(const f16x8*)(base + (long)row * ld + c * 8);
            ov = *(const f16x8*)(dob + (long)row * H + c * 8);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { Qt[(c * 8 + j) * VP + (c & 7) * 8 + row] = qv[j]; dOt[(c * 8 + j) * VP + (c & 7) * 8 + row] = ov[j]; }
    }
    __syncthreads();
    for (int kt = wv; kt * 16 < T; kt += 4) {
        int krow = kt * 16 + fr;
        int kr = krow < T ? krow : T - 1;
        f16x8 kf[KK], vf[KK];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            kf[kk] = (kk * 4 + g < chm) ? *(const f16x8*)(base + (long)kr * ld + H + kk * 32 + g * 8) : zero8;
            vf[kk] = (kk * 4 + g < chm) ? *(const f16x8*)(base + (long)kr * ld + 2 * H + kk * 32 + g * 8) : zero8;
        }
        f32x4 dvt[DT], dkt[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) { dvt[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dkt[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        // walk the queries two 16-row tiles (= one 32-deep MFMA k step) at a time
        for (int ss = 0; ss < NS; ++ss) {
            f16x8 pf, dsf;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                int qt = 2 * ss + half;
                f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                int qrow = qt * 16 + fr;
                int qr = qrow < T ? qrow : T - 1;
                if (qt < NT) {
#pragma unroll
                    for (int kk = 0; kk < KK; ++kk) {
                        f16x8 qf = (kk * 4 + g < chm) ? *(const f16x8*)(base + (long)qr * ld + kk * 32 + g * 8) : zero8;
                        f16x8 of = (kk * 4 + g < chm) ? *(const f16x8*)(dob + (long)qr * H + kk * 32 + g * 8) : zero8;
                        s = __builtin_amdgcn_mfma_f32_16x16x32_f16(qf, kf[kk], s, 0, 0, 0);      // S  [q][key]
                        dp = __builtin_amdgcn_mfma_f32_16x16x32_f16(of, vf[kk], dp, 0, 0, 0);    // dP [q][key]
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int q = qt * 16 + g * 4 + r;                   // this lane's query rows; its key column = krow
                    float p = 0.f, ds = 0.f;
                    if (qt < NT && q < T && krow < T) {
                        p = __builtin_amdgcn_exp2f(s[r] * (scale * LOG2E) - rmax[q]) * rinv[q];   // rmax is kept in the log2 domain
                        ds = p * (dp[r] - rdel[q]) * scale;
                    }
                    pf[half * 4 + r] = (_Float16)p;
                    dsf[half * 4 + r] = (_Float16)ds;
                }
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                f16x8 oa, qa;
                if (TR) {                                           // dO^T and Q^T from the row-major tiles (Q over K's space, dO over V's)
                    const int ra = 32 * ss + 4 * g + tq, rb = (2 * ss + 1 < NT) ? ra + 16 : ra;
                    qa = tr_frag64(smem3, ra, rb, dt * 16, tp);
                    oa = tr_frag64(smem3 + (size_t)ROWB * 2, ra, rb, dt * 16, tp);
                } else {
                    const _Float16* orow = dOt + (dt * 16 + fr) * VP + (((dt * 16 + fr) >> 3) & 7) * 8 + ss * 32 + g * 4;
                    const _Float16* qrow_ = Qt + (dt * 16 + fr) * VP + (((dt * 16 + fr) >> 3) & 7) * 8 + ss * 32 + g * 4;
                    f16x4 olo = *(const f16x4*)orow, qlo = *(const f16x4*)qrow_;
                    f16x4 ohi = (2 * ss + 1 < NT) ? *(const f16x4*)(orow + 16) : f16x4{0, 0, 0, 0};
                    f16x4 qhi = (2 * ss + 1 < NT) ? *(const f16x4*)(qrow_ + 16) : f16x4{0, 0, 0, 0};
                    oa = f16x8{olo[0], olo[1], olo[2], olo[3], ohi[0], ohi[1], ohi[2], ohi[3]};
                    qa = f16x8{qlo[0], qlo[1], qlo[2], qlo[3], qhi[0], qhi[1], qhi[2], qhi[3]};
                }
                dvt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(oa, pf, dvt[dt], 0, 0, 0);      // dV^T [d][key]
                dkt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qa, dsf, dkt[dt], 0, 0, 0);     // dK^T [d][key]
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (krow < T) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                if (dt * 16 + g * 4 >= dm) continue;
                f16x4 kv = {(_Float16)dkt[dt][0], (_Float16)dkt[dt][1], (_Float16)dkt[dt][2], (_Float16)dkt[dt][3]};
                f16x4 vv = {(_Float16)dvt[dt][0], (_Float16)dvt[dt][1], (_Float16)dvt[dt][2], (_Float16)dvt[dt][3]};
                *(f16x4*)(dbase + (long)krow * ld + H + dt * 16 + g * 4) = kv;
                *(f16x4*)(dbase + (long)krow * ld + 2 * H + dt * 16 + g * 4) = vv;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- head / misc
// dh[b][t][:] = coef[:] * (dlogit[b] / T)      (gradient of mean-pool + Linear(H,1))
__global__ __launch_bounds__(256) void pool_logreg_bwd_kernel(const float* __restrict__ coef, const float* __restrict__ dlogit,
                                                              float* __restrict__ dh, _Float16* __restrict__ dh16, int T, int H, long total4, long dh16_lo) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        int c = (int)(i % (H / 4)) * 4;
        long bt = i / (H / 4);
        int b = (int)(bt / T);
        float s = dlogit[b] / T;
        float4 w = *(const float4*)(coef + c);
        float o[4] = {w.x * s, w.y * s, w.z * s, w.w * s};
        if (dh) *(float4*)(dh + bt * H + c) = make_float4(o[0], o[1], o[2], o[3]);
        if (dh16) store_h_rt<4>(dh16, bt * H + c, dh16_lo, o);
    }
}

// y[i] = alpha[row] * x[i] (+ y[i] if accumulate);  row = i / n.  IG path scaling and step accumulation.
__global__ __launch_bounds__(256) void scale_rows_kernel(const float* __restrict__ x, const float* __restrict__ alpha,
                                                         float* __restrict__ y, long n, long total, int accumulate, int xrows) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        long row = i / n, col = i - row * n;
        float v = alpha[row] * x[(row % xrows) * n + col];
        y[i] = accumulate ? y[i] + v : v;
    }
}

// attr = |g| (mode 0: Saliency), x * g (mode 1: InputXGradient / IntegratedGradients with a zero baseline)
__global__ __launch_bounds__(256) void attr_finalize_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                            float* __restrict__ out, int mode, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256)
        out[i] = mode == 0 ? fabsf(g[i]) : x[i] * g[i];
}

// mask[b][:] = |attr[b][:]| / (max|attr[b][:]| + 1e-8)          (captum_saliency.py:136-139), one workgroup per clip
__global__ __launch_bounds__(1024) void time_mask_kernel(const float* __restrict__ attr, float* __restrict__ mask, long n) {
    __shared__ float red[16];
    const float* a = attr + (long)blockIdx.x * n;
    float m = 0.f;
    for (long i = threadIdx.x; i < n; i += 1024) m = fmaxf(m, fabsf(a[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = red[0];
    for (int i = 1; i < 16; ++i) m = fmaxf(m, red[i]);
    const float d = m + 1e-8f;
    for (long i = threadIdx.x; i < n; i += 1024) mask[(long)blockIdx.x * n + i] = fabsf(a[i]) / d;
}

// wave_in = wave * mask, wave_out = wave * (1 - mask)            (captum_saliency.py:141-143)
__global__ __launch_bounds__(256) void apply_time_mask_kernel(const float* __restrict__ wave, const float* __restrict__ mask,
                                                              float* __restrict__ win, float* __restrict__ wout, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        float w = wave[i], m = mask[i];
        win[i] = w * m;
        wout[i] = w * (1.f - m);
    }
}

}  // namespace advh

using namespace advh;

extern "C" int advh_attr_finalize(const float* g, const float* x, float* out, int mode, int64_t total, advh_stream_t stream) {
    if (!g || !out || (mode == 1 && !x) || (mode != 0 && mode != 1) || total <= 0) return ADVH_EINVAL;
    long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(attr_finalize_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, x, out, mode, (long)total);
    return ADVH_LAUNCH_CHECK();
}

extern "C" int advh_time_mask(const float* attr, float* mask, float* wave_in, float* wave_out, const float* wave, int B, int64_t n,
                              advh_stream_t stream) {
    if (!attr || !mask || B <= 0 || n <= 0 || ((wave_in || wave_out) && (!wave || !wave_in || !wave_out))) return ADVH_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(time_mask_kernel, dim3(B), dim3(1024), 0, s, attr, mask, (long)n);
    if (wave_in) {
        long total = (long)B * n, blocks = (total + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(apply_time_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, s, wave, (const float*)mask, wave_in, wave_out, total);
    }
    return ADVH_LAUNCH_CHECK();
}

static int layernorm_bwd_launch(const void* x, int x_is_f32, const void* dy, int dy_is_f32, const float* gamma,
                                const float* beta, int gelu_fwd, const float* add, const void* dact_src, float* out_f,
                                void* out_h, int M, int C, float eps, int remap_T, int remap_P, long x_lo, long dy_lo, long dact_lo,
                                long out_lo, advh_stream_t stream) {
    if (remap_P > 0 && (remap_T <= 0 || remap_P < remap_T)) return ADVH_EINVAL;
    if (!x || !dy || !gamma || (gelu_fwd && !beta) || (!out_f && !out_h) || M <= 0 || C <= 0 || C % 4) return ADVH_EINVAL;
    if (C > 64 * 4 * 8) return ADVH_EUNSUPPORTED;
    dim3 grid((M + 3) / 4), block(256);
    hipStream_t s = (hipStream_t)stream;
#define LNB(X32, D32, MV)                                                                                            \
    hipLaunchKernelGGL((layernorm_bwd_kernel<X32, D32, MV>), grid, block, 0, s, x, dy, gamma, beta, gelu_fwd, add,     \
                       (const _Float16*)dact_src, out_f, (_Float16*)out_h, M, C, eps, remap_T, remap_P, x_lo, dy_lo, dact_lo, out_lo)
#define LNB_MV(MV)                                                                                                   \
    do {                                                                                                             \
        if (x_is_f32 && dy_is_f32) LNB(true, true, MV); else if (x_is_f32) LNB(true, false, MV);                     \
        else if (dy_is_f32) LNB(false, true, MV); else LNB(false, false, MV);                                        \
    } while (0)
    if (C <= 64 * 4 * 2) LNB_MV(2); else if (C <= 64 * 4 * 4) LNB_MV(4); else LNB_MV(8);
#undef LNB_MV
#undef LNB
    return ADVH_LAUNCH_CHECK();
}

extern "C" int advh_layernorm_bwd(const void* x, int x_is_f32, const void* dy, int dy_is_f32, const float* gamma,
                                  const float* beta, int gelu_fwd, const float* add, const void* dact_src, float* out_f,
                                  void* out_h, int M, int C, float eps, int remap_T, int remap_P, advh_stream_t stream) {
    return layernorm_bwd_launch(x, x_is_f32, dy, dy_is_f32, gamma, beta, gelu_fwd, add, dact_src, out_f, out_h, M, C, eps, remap_T,
                                remap_P, 0, 0, 0, 0, stream);
}

extern "C" int advh_layernorm_bwd_split(const void* x, int x_is_f32, int64_t x_lo, const void* dy, int dy_is_f32, int64_t dy_lo,
                                        const float* gamma, const float* beta, int gelu_fwd, const float* add, const void* dact_src,
                                        int64_t dact_lo, float* out_f, void* out_h, int64_t out_lo, int M, int C, float eps,
                                        int remap_T, int remap_P, advh_stream_t stream) {
    if ((!x_is_f32 && x_lo <= 0) || (!dy_is_f32 && dy_lo <= 0) || (dact_src && dact_lo <= 0) || (out_h && out_lo <= 0)) return ADVH_EINVAL;
    if (x_lo % 4 || dy_lo % 4 || dact_lo % 4 || out_lo % 4) return ADVH_EINVAL;
    return layernorm_bwd_launch(x, x_is_f32, dy, dy_is_f32, gamma, beta, gelu_fwd, add, dact_src, out_f, out_h, M, C, eps, remap_T,
                                remap_P, x_is_f32 ? 0 : x_lo, dy_is_f32 ? 0 : dy_lo, dact_src ? dact_lo : 0, out_h ? out_lo : 0, stream);
}

template <int NT, int D, bool TR>
static int launch_att_bwd(const void* qkv, const void* dctx, void* dqkv, int B, int T, int H, int heads, int dm, float scale, hipStream_t s) {
    constexpr int NKEY = NT * 16, VP = NKEY + 64, ROWB = NKEY * D, TRB = TR ? 0 : D * VP;
    constexpr int PA = TR ? 2 * ROWB : (D <= 64 ? 2 * ROWB + TRB : ROWB + TRB);
    const size_t lds = (size_t)(PA > 2 * TRB ? PA : 2 * TRB) * 2 + 3 * NKEY * 4;
    if (lds > 160 * 1024) return ADVH_EUNSUPPORTED;
    if (advh_ensure_lds((const void*)attention_bwd_kernel<NT, D, TR>) != ADVH_OK) return ADVH_ELAUNCH;
    hipLaunchKernelGGL((attention_bwd_kernel<NT, D, TR>), dim3(heads, B), dim3(256), lds, s, (const _Float16*)qkv, (const _Float16*)dctx,
                       (_Float16*)dqkv, T, H, dm, scale);
    return ADVH_LAUNCH_CHECK();
}

extern "C" int advh_attention_bwd_f16(const void* qkv, const void* dctx, void* dqkv, int B, int T, int H, int heads,
                                      advh_stream_t stream) {
    if (!qkv || !dctx || !dqkv || B <= 0 || T <= 0 || heads <= 0 || H % heads) return ADVH_EINVAL;
    const int dm = H / heads;
    if (T > 256 || dm % 8 || dm > 128) return ADVH_EUNSUPPORTED;
    const int D = dm <= 32 ? 32 : (dm <= 64 ? 64 : 128);
    const float scale = 1.f / sqrtf((float)dm);
    hipStream_t s = (hipStream_t)stream;
    const int nt = (T + 15) / 16;
#define ATB(NT_, D_) return launch_att_bwd<NT_, D_, false>(qkv, dctx, dqkv, B, T, H, heads, dm, scale, s)
#define ATBT(NT_) return launch_att_bwd<NT_, 64, true>(qkv, dctx, dqkv, B, T, H, heads, dm, scale, s)
    if (dm == 64) { if (nt <= 4) ATBT(4); else if (nt <= 8) ATBT(8); else if (nt <= 13) ATBT(13); else ATBT(16); }
    else if (D == 64) { if (nt <= 4) ATB(4, 64); else if (nt <= 8) ATB(8, 64); else if (nt <= 13) ATB(13, 64); else ATB(16, 64); }
    else if (D == 32) { if (nt <= 4) ATB(4, 32); else if (nt <= 8) ATB(8, 32); else if (nt <= 13) ATB(13, 32); else ATB(16, 32); }
    else { if (nt <= 4) ATB(4, 128); else if (nt <= 13) ATB(13, 128); else ATB(16, 128); }
#undef ATB
#undef ATBT
}

static int pool_logreg_bwd_launch(const float* coef, const float* dlogit, float* dh, void* dh16, long dh16_lo, int B, int T, int H,
                                  advh_stream_t stream) {
    if (!coef || !dlogit || (!dh && !dh16) || B <= 0 || T <= 0 || H <= 0 || H % 4) return ADVH_EINVAL;
    long total4 = (long)B * T * (H / 4);
    long blocks = (total4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(pool_logreg_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, coef, dlogit, dh,
                       (_Float16*)dh16, T, H, total4, dh16_lo);
    return ADVH_LAUNCH_CHECK();
}

extern "C" int advh_pool_logreg_bwd(const float* coef, const float* dlogit, float* dh, void* dh16, int B, int T, int H,
                                    advh_stream_t stream) {
    return pool_logreg_bwd_launch(coef, dlogit, dh, dh16, 0, B, T, H, stream);
}

extern "C" int advh_pool_logreg_bwd_split(const float* coef, const float* dlogit, float* dh, void* dh16, int64_t dh16_lo, int B, int T,
                                          int H, advh_stream_t stream) {
    if (dh16 && (dh16_lo <= 0 || dh16_lo % 4)) return ADVH_EINVAL;
    return pool_logreg_bwd_launch(coef, dlogit, dh, dh16, dh16 ? dh16_lo : 0, B, T, H, stream);
}

extern "C" int advh_scale_rows(const float* x, int x_rows, const float* alpha, float* y, int rows, int64_t n, int accumulate,
                               advh_stream_t stream) {
    if (!x || !alpha || !y || rows <= 0 || n <= 0 || x_rows <= 0) return ADVH_EINVAL;
    long total = (long)rows * n;
    long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, alpha, y, (long)n, total,
                       accumulate, x_rows);
    return ADVH_LAUNCH_CHECK();
}

ADVH_SPLIT_FLAG_SETTER(advh_split_flag_backward)
