// Backward of the waveform front end (normaliser + feature-encoder layer 0), input gradient only.
//   forward (frontend.hip): xhat = (x - mu) / (sigma + 1e-7);  z0[t,c] = sum_j w0[c,j] xhat[5t+j];
//                           group mode: y0 = GELU(z0 * scale_c + shift_c)  (GroupNorm over time folded)
// Given dy0 [B][P0][C0] (fp16, loss-scaled):
//   1. gn0_bwd_stats : per (clip, channel) time sums of dn and dn*nhat, dn = dy0 * GELU'(u) * gamma
//                      (z0 is recomputed from the waveform: 10 MACs, nothing was saved)
//   2. gn0_bwd_dz    : dz0 = r_c (dn - mean_t dn - nhat mean_t(dn nhat))           -> fp16 [B][P0][C0]
//   3. implicit GEMM : g[t][j] = sum_c dz0[t,c] w0[c,j]                             (advh_gemm_f16, N = 10 -> 16)
//   4. wave_bwd      : dxhat[u] = g[q][phi] + g[q-1][phi+5], u = 5q + phi;  then the normaliser's Jacobian.
// The "layer" feature extractor uses advh_layernorm_bwd for steps 1-2 and shares 3-4.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include "addvisor_hip.h"
#include "common.h"
#include "device_math.h"

namespace advh {

constexpr int K0 = 10, S0 = 5, TT = 64;

__device__ __forceinline__ float ldn(const float* w, int i, int n, float mean, float rstd) {
    float x = i < n ? w[i] : 0.f;
    return (x - mean) * rstd;
}

// part[b][tile][c] = (sum_t dn, sum_t dn * nhat) over the tile's 64 frames
template <int PASS>
__global__ __launch_bounds__(256) void gn0_bwd_kernel(const float* __restrict__ wave, long stride, int n_in, int L,
                                                      const float2* __restrict__ stats, const float* __restrict__ w0,
                                                      const float2* __restrict__ norm, const float2* __restrict__ mr,
                                                      const float* __restrict__ gamma, const _Float16* __restrict__ dy,
                                                      float2* __restrict__ part, const float2* __restrict__ sums,
                                                      _Float16* __restrict__ dz, int T0, int P0, int C0, long dy_lo, long dz_lo) {
    __shared__ float xs[TT * S0 + K0];
    const int b = blockIdx.y, t0 = blockIdx.x * TT, tid = threadIdx.x, ntile = gridDim.x;
    const float* w = wave + (long)b * stride;
    const int n = n_in < L ? n_in : L;
    const float2 st = stats[b];
    for (int i = tid; i < TT * S0 + K0; i += 256) xs[i] = ldn(w, S0 * t0 + i, n, st.x, st.y);
    __syncthreads();
    const int c = 2 * tid;
    if (c >= C0) return;
    float wa[K0], wb[K0];
#pragma unroll
    for (int k = 0; k < K0; ++k) { wa[k] = w0[c * K0 + k]; wb[k] = w0[(c + 1) * K0 + k]; }
    const float2 na = norm[(long)b * C0 + c], nb = norm[(long)b * C0 + c + 1];     // (scale, shift) of u = z*scale+shift
    const float2 ma = mr[(long)b * C0 + c], mb = mr[(long)b * C0 + c + 1];         // (mean_c, rstd_c) of z over time
    const float ga = gamma[c], gb = gamma[c + 1];
    float s1a = 0.f, s2a = 0.f, s1b = 0.f, s2b = 0.f;
    float m1a = 0.f, m2a = 0.f, m1b = 0.f, m2b = 0.f;
    if (PASS == 1) {
        float2 sa = sums[(long)b * C0 + c], sb = sums[(long)b * C0 + c + 1];
        m1a = sa.x / T0; m2a = sa.y / T0; m1b = sb.x / T0; m2b = sb.y / T0;
    }
    const int tend = min(TT, P0 - t0);
    for (int t = 0; t < tend; ++t) {
        const long o = ((long)b * P0 + t0 + t) * C0 + c;
        float da = 0.f, db = 0.f;
        if (t0 + t < T0) {
            float za = 0.f, zb = 0.f;
#pragma unroll
            for (int k = 0; k < K0; ++k) { float x = xs[S0 * t + k]; za = fmaf(wa[k], x, za); zb = fmaf(wb[k], x, zb); }
            float d2[2];
            load_h_rt<2>(dy, o, dy_lo, d2);                   // fp16, or the split-format plane pair (dy_lo != 0)
            float dna = d2[0] * gelu_grad(za * na.x + na.y) * ga;
            float dnb = d2[1] * gelu_grad(zb * nb.x + nb.y) * gb;
            float ha = (za - ma.x) * ma.y, hb = (zb - mb.x) * mb.y;
            if (PASS == 0) { s1a += dna; s2a += dna * ha; s1b += dnb; s2b += dnb * hb; }
            else { da = ma.y * (dna - m1a - ha * m2a); db = mb.y * (dnb - m1b - hb * m2b); }
        }
        if (PASS == 1) { const float dd[2] = {da, db}; store_h_rt<2>(dz, o, dz_lo, dd); }
    }
    if (PASS == 0) {
        part[((long)b * ntile + blockIdx.x) * C0 + c] = make_float2(s1a, s2a);
        part[((long)b * ntile + blockIdx.x) * C0 + c + 1] = make_float2(s1b, s2b);
    }
}

__global__ __launch_bounds__(256) void gn0_reduce_kernel(const float2* __restrict__ part, float2* __restrict__ sums, int ntile, int C0) {
    const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C0) return;
    float a = 0.f, d = 0.f;
    for (int i = 0; i < ntile; ++i) { float2 v = part[((long)b * ntile + i) * C0 + c]; a += v.x; d += v.y; }
    sums[(long)b * C0 + c] = make_float2(a, d);
}

// dxhat[u] = g[q][phi] + g[q-1][phi + 5]  (g: [B][P0][16] fp32, rows >= T0 are zero); per-tile partial sums for
// the normaliser Jacobian: S1 = sum dxhat, S2 = sum dxhat * xhat.
constexpr int WT = 2048;
__global__ __launch_bounds__(256) void wave_bwd_gather_kernel(const float* __restrict__ g, const float* __restrict__ wave, long stride,
                                                              int n_in, int L, const float2* __restrict__ stats, float* __restrict__ dxh,
                                                              float2* __restrict__ part, int T0, int P0) {
    __shared__ float r1[4], r2[4];
    const int b = blockIdx.y, u0 = blockIdx.x * WT, tid = threadIdx.x;
    const int n = n_in < L ? n_in : L;
    const float2 st = stats[b];
    const float* w = wave + (long)b * stride;
    float s1 = 0.f, s2 = 0.f;
    for (int i = tid; i < WT; i += 256) {
        int u = u0 + i;
        if (u >= L) break;
        int q = u / S0, phi = u - q * S0;
        float v = 0.f;
        if (q < T0) v += g[((long)b * P0 + q) * 16 + phi];
        if (q >= 1 && q - 1 < T0) v += g[((long)b * P0 + q - 1) * 16 + phi + S0];
        dxh[(long)b * L + u] = v;
        float xh = ldn(w, u, n, st.x, st.y);
        s1 += v; s2 += v * xh;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    if ((tid & 63) == 0) { r1[tid >> 6] = s1; r2[tid >> 6] = s2; }
    __syncthreads();
    if (tid == 0) part[(long)b * gridDim.x + blockIdx.x] = make_float2((r1[0] + r1[1]) + (r1[2] + r1[3]), (r2[0] + r2[1]) + (r2[2] + r2[3]));
}

// dx[k] = rho (g_k - S1/L) - xhat_k S2 / ((L-1) sigma),  rho = 1/(sigma + 1e-7);   out = dx * out_scale (k < n_in)
__global__ __launch_bounds__(256) void wave_bwd_final_kernel(const float* __restrict__ dxh, const float* __restrict__ wave, long stride,
                                                             int n_in, int L, const float2* __restrict__ stats, const float2* __restrict__ part,
                                                             int npart, int normalize, float out_scale, float* __restrict__ dx, long dx_stride) {
    __shared__ float sh[2];
    const int b = blockIdx.y, tid = threadIdx.x;
    if (tid == 0) {
        float a = 0.f, d = 0.f;
        for (int i = 0; i < npart; ++i) { float2 v = part[(long)b * npart + i]; a += v.x; d += v.y; }
        sh[0] = a; sh[1] = d;
    }
    __syncthreads();
    const float2 st = stats[b];
    const float rho = st.y, sigma = 1.f / st.y - 1e-7f;
    const float S1 = sh[0], S2 = sh[1];
    const int n = n_in < L ? n_in : L;
    const float* w = wave + (long)b * stride;
    for (int i = tid; i < WT; i += 256) {
        int u = blockIdx.x * WT + i;
        if (u >= n_in) break;
        float v = 0.f;
        if (u < L) {
            float gk = dxh[(long)b * L + u];
            if (normalize) {
                float xh = (w[u] - st.x) * rho;
                v = rho * (gk - S1 / L) - xh * S2 / ((float)(L - 1) * sigma);
            } else {
                v = gk;
            }
        }
        dx[(long)b * dx_stride + u] = v * out_scale;
    }
    (void)n;
}

}  // namespace advh

using namespace advh;

static int frontend_bwd_group_launch(const float* wave, int64_t wave_stride, int n_in, int B, int L, const float* w0,
                                     const float* gamma, const float* stats_ws, const float* norm_ws, const float* mr_ws,
                                     const void* dy0, long dy_lo, float* part_ws, float* sums_ws, void* dz0, long dz_lo, int T0, int P0,
                                     int C0, advh_stream_t stream) {
    if (!wave || !w0 || !gamma || !stats_ws || !norm_ws || !mr_ws || !dy0 || !part_ws || !sums_ws || !dz0) return ADVH_EINVAL;
    if (B <= 0 || C0 <= 0 || C0 > 512 || (C0 & 1) || T0 != (L - K0) / S0 + 1 || P0 < T0) return ADVH_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((P0 + TT - 1) / TT, B);
    hipLaunchKernelGGL(gn0_bwd_kernel<0>, grid, dim3(256), 0, s, wave, (long)wave_stride, n_in, L, (const float2*)stats_ws, w0,
                       (const float2*)norm_ws, (const float2*)mr_ws, gamma, (const _Float16*)dy0, (float2*)part_ws, (const float2*)nullptr,
                       (_Float16*)nullptr, T0, P0, C0, dy_lo, dz_lo);
    hipLaunchKernelGGL(gn0_reduce_kernel, dim3((C0 + 255) / 256, B), dim3(256), 0, s, (const float2*)part_ws, (float2*)sums_ws, (int)grid.x, C0);
    hipLaunchKernelGGL(gn0_bwd_kernel<1>, grid, dim3(256), 0, s, wave, (long)wave_stride, n_in, L, (const float2*)stats_ws, w0,
                       (const float2*)norm_ws, (const float2*)mr_ws, gamma, (const _Float16*)dy0, (float2*)nullptr, (const float2*)sums_ws,
                       (_Float16*)dz0, T0, P0, C0, dy_lo, dz_lo);
    return ADVH_LAUNCH_CHECK();
}

extern "C" int advh_w2v2_frontend_bwd_group(const float* wave, int64_t wave_stride, int n_in, int B, int L, const float* w0,
                                            const float* gamma, const float* stats_ws, const float* norm_ws, const float* mr_ws,
                                            const void* dy0, float* part_ws, float* sums_ws, void* dz0, int T0, int P0, int C0,
                                            advh_stream_t stream) {
    return frontend_bwd_group_launch(wave, wave_stride, n_in, B, L, w0, gamma, stats_ws, norm_ws, mr_ws, dy0, 0, part_ws, sums_ws, dz0, 0,
                                     T0, P0, C0, stream);
}

extern "C" int advh_w2v2_frontend_bwd_group_split(const float* wave, int64_t wave_stride, int n_in, int B, int L, const float* w0,
                                                  const float* gamma, const float* stats_ws, const float* norm_ws, const float* mr_ws,
                                                  const void* dy0, int64_t dy_lo, float* part_ws, float* sums_ws, void* dz0, int64_t dz_lo,
                                                  int T0, int P0, int C0, advh_stream_t stream) {
    if (dy_lo <= 0 || dz_lo <= 0 || dy_lo % 2 || dz_lo % 2) return ADVH_EINVAL;
    return frontend_bwd_group_launch(wave, wave_stride, n_in, B, L, w0, gamma, stats_ws, norm_ws, mr_ws, dy0, dy_lo, part_ws, sums_ws, dz0,
                                     dz_lo, T0, P0, C0, stream);
}

extern "C" int advh_wave_bwd(const float* g, const float* wave, int64_t wave_stride, int n_in, int B, int L, const float* stats_ws,
                             float* dxhat_ws, float* part_ws, int normalize, float out_scale, float* dx, int64_t dx_stride, int T0,
                             int P0, advh_stream_t stream) {
    if (!g || !wave || !stats_ws || !dxhat_ws || !part_ws || !dx || B <= 0 || L <= 0 || n_in <= 0 || dx_stride < n_in) return ADVH_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int nt = (L + WT - 1) / WT, nto = ((n_in > L ? n_in : L) + WT - 1) / WT;
    hipLaunchKernelGGL(wave_bwd_gather_kernel, dim3(nt, B), dim3(256), 0, s, g, wave, (long)wave_stride, n_in, L, (const float2*)stats_ws,
                       dxhat_ws, (float2*)part_ws, T0, P0);
    hipLaunchKernelGGL(wave_bwd_final_kernel, dim3(nto, B), dim3(256), 0, s, (const float*)dxhat_ws, wave, (long)wave_stride, n_in, L,
                       (const float2*)stats_ws, (const float2*)part_ws, nt, normalize, out_scale, dx, (long)dx_stride);
    return ADVH_LAUNCH_CHECK();
}

ADVH_SPLIT_FLAG_SETTER(advh_split_flag_frontend_bwd)
