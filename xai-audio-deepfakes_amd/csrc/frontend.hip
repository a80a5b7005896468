// wav2vec2 waveform front end: clip normaliser + feature-encoder layer 0.
//
//   zero_mean_unit_var_norm           classifier_embedder.py:59-63 (unbiased std, eps added to std)
//   Conv1d(1 -> C0, k=10, s=5)        transformers/.../modeling_wav2vec2.py:254-323 (layer_id 0)
//   GroupNorm(C0 groups) + GELU       :302-323  ("group" feature extractor, wav2vec2-base)
//
// Layer 0 has one input channel, so it is 10 MACs per output and HBM-bound on its fp16 output
// ([B][P0][C0] channels-last, 13 MB per 4 s clip): no matrix cores.  GroupNorm with one group per
// channel normalises over TIME; because the layer is linear in the waveform its per-channel mean
// and variance follow exactly from the 10 x 10 Gram matrix of the strided input windows, so the
// statistics cost one pass over the 256 KB clip instead of two passes over the 13 MB activation:
//     mean_c = sum_k w[c,k] S1[k],   E[y_c^2] = sum_{k,k'} w[c,k] w[c,k'] S2[k,k'].
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include "addvisor_hip.h"
#include "common.h"
#include "device_math.h"

namespace advh {

constexpr int K0 = 10, S0 = 5;     // kernel / stride of feature-encoder layer 0 (every wav2vec2 config)

__device__ __forceinline__ double block_sum_d(double v, double* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
    return s;
}

// stats[b] = (mean, 1/(std_unbiased + 1e-7)) of the clip padded / cropped to L samples
__global__ __launch_bounds__(256) void wave_stats_kernel(const float* __restrict__ wave, long stride, int n_in, int L,
                                                         float2* __restrict__ stats, int normalize) {
    __shared__ double red[4];
    if (!normalize) {                                  // input_values already normalised by the caller
        if (threadIdx.x == 0) stats[blockIdx.x] = make_float2(0.f, 1.f);
        return;
    }
    const float* w = wave + (long)blockIdx.x * stride;
    const int n = n_in < L ? n_in : L;
    // 16-byte loads with four independent fp64 accumulators (the scalar loop was latency-bound: 129 us for 49 MB)
    const int n4 = (((unsigned long)w & 15) == 0) ? n / 4 : 0;
    const float4* w4 = (const float4*)w;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int i = threadIdx.x; i < n4; i += 256) { const float4 v = w4[i]; s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w; }
    for (int i = 4 * n4 + threadIdx.x; i < n; i += 256) s0 += w[i];
    const double mean = block_sum_d((s0 + s1) + (s2 + s3), red) / L;
    double q0 = 0, q1 = 0, q2 = 0, q3 = 0;
    for (int i = threadIdx.x; i < n4; i += 256) {
        const float4 v = w4[i];
        const double d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
        q0 += d0 * d0; q1 += d1 * d1; q2 += d2 * d2; q3 += d3 * d3;
    }
    for (int i = 4 * n4 + threadIdx.x; i < n; i += 256) { const double d = w[i] - mean; q0 += d * d; }
    double q = block_sum_d((q0 + q1) + (q2 + q3), red) + (double)(L - n) * mean * mean;       // zero-padded tail
    if (threadIdx.x == 0) {
        float sd = (float)sqrt(q / (L - 1));
        stats[blockIdx.x] = make_float2((float)mean, 1.f / (sd + 1e-7f));
    }
}

__device__ __forceinline__ float load_norm(const float* w, int i, int n, float mean, float rstd) {
    float x = i < n ? w[i] : 0.f;
    return (x - mean) * rstd;
}

// GroupNorm statistics from the Gram matrix; norm[b][c] = (scale, shift) with
// y_norm = conv(xhat)[c] * scale + shift  (gamma, beta and eps = 1e-5 folded in).
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ wave, long stride, int n_in, int L,
                                                       const float2* __restrict__ stats, const float* __restrict__ w0,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float2* __restrict__ norm, float2* __restrict__ mr, int T0, int C0) {
    __shared__ double S[K0 + K0 * K0];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* w = wave + (long)b * stride;
    const int n = n_in < L ? n_in : L;
    const float2 st = stats[b];
    double s1[K0], s2[K0 * (K0 + 1) / 2];
#pragma unroll
    for (int k = 0; k < K0; ++k) s1[k] = 0;
#pragma unroll
    for (int k = 0; k < K0 * (K0 + 1) / 2; ++k) s2[k] = 0;
    for (int t = tid; t < T0; t += 256) {
        float x[K0];
#pragma unroll
        for (int k = 0; k < K0; ++k) x[k] = load_norm(w, S0 * t + k, n, st.x, st.y);
        int idx = 0;
#pragma unroll
        for (int k = 0; k < K0; ++k) {
            s1[k] += x[k];
#pragma unroll
            for (int j = k; j < K0; ++j) s2[idx++] += (double)x[k] * x[j];
        }
    }
    {   // all 65 sums in ONE exchange: wavefront shuffles, partials to LDS, fixed-order add (was 65 block reductions)
        __shared__ double part[4][K0 + K0 * (K0 + 1) / 2];
        const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
        for (int k = 0; k < K0; ++k) {
            double v = s1[k];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) part[wv][k] = v;
        }
#pragma unroll
        for (int k = 0; k < K0 * (K0 + 1) / 2; ++k) {
            double v = s2[k];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) part[wv][K0 + k] = v;
        }
        __syncthreads();
        if (tid < K0) S[tid] = ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid])) / T0;
        if (tid < K0 * (K0 + 1) / 2) {
            int k = 0, rem = tid;                                  // tid -> (k, j >= k) in the row-major upper triangle
            while (rem >= K0 - k) { rem -= K0 - k; ++k; }
            const int j = k + rem, i = K0 + tid;
            const double v = ((part[0][i] + part[1][i]) + (part[2][i] + part[3][i])) / T0;
            S[K0 + k * K0 + j] = v;
            S[K0 + j * K0 + k] = v;
        }
    }
    __syncthreads();
    for (int c = tid; c < C0; c += 256) {
        double m = 0, e2 = 0;
        for (int k = 0; k < K0; ++k) {
            double wk = w0[c * K0 + k];
            m += wk * S[k];
            for (int j = 0; j < K0; ++j) e2 += wk * (double)w0[c * K0 + j] * S[K0 + k * K0 + j];
        }
        double var = e2 - m * m;
        if (var < 0) var = 0;
        float sc = gamma[c] * (float)(1.0 / sqrt(var + 1e-5));
        norm[(long)b * C0 + c] = make_float2(sc, beta[c] - (float)m * sc);
        if (mr) mr[(long)b * C0 + c] = make_float2((float)m, (float)(1.0 / sqrt(var + 1e-5)));     // saved for the backward
    }
}

// out[b][t][c] = GELU(conv * scale + shift)   (mode 0, "group")   or   conv + bias   (mode 1, raw: the
// LayerNorm + GELU of the "layer" feature extractor follow as a row kernel).  Rows t in [T0, P0) = 0.
// A thread owns 8 consecutive channels (their 80 taps live in registers) and walks the tile's frames, so every
// store is one 16-byte vector and a wavefront writes 1 KiB contiguous per frame (C0 = 512: 64 lanes x 8 ch).
constexpr int TT = 64;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void conv0_kernel(const float* __restrict__ wave, long stride, int n_in, int L,
                                                    const float2* __restrict__ stats, const float* __restrict__ w0,
                                                    const float* __restrict__ bias, const float2* __restrict__ norm,
                                                    _Float16* __restrict__ out, long out_lo, int T0, int P0, int C0, int mode) {
    __shared__ float xs[TT * S0 + K0];
    const int b = blockIdx.y, t0 = blockIdx.x * TT, tid = threadIdx.x;
    const float* w = wave + (long)b * stride;
    const int n = n_in < L ? n_in : L;
    const float2 st = stats[b];
    for (int i = tid; i < TT * S0 + K0; i += 256) xs[i] = load_norm(w, S0 * t0 + i, n, st.x, st.y);
    __syncthreads();
    const int ngrp = C0 / 8;                       // channel groups of 8
    const int grp = tid % ngrp, tsub = tid / ngrp, tstep = 256 / ngrp;     // C0 = 512: 64 groups x 4 frame lanes
    if (tsub >= tstep) return;
    const int c = grp * 8;
    float wr[8][K0], sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int k = 0; k < K0; ++k) wr[j][k] = w0[(c + j) * K0 + k];
        sc[j] = 1.f; sh[j] = 0.f;
        if (mode == 0) { float2 nn = norm[(long)b * C0 + c + j]; sc[j] = nn.x; sh[j] = nn.y; }
        else if (bias) sh[j] = bias[c + j];
    }
    const int tend = min(TT, P0 - t0);
    for (int t = tsub; t < tend; t += tstep) {
        float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (t0 + t < T0) {
            float x[K0];
#pragma unroll
            for (int k = 0; k < K0; ++k) x[k] = xs[S0 * t + k];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float y = 0.f;
#pragma unroll
                for (int k = 0; k < K0; ++k) y = fmaf(wr[j][k], x[k], y);
                y = y * sc[j] + sh[j];
                if (mode == 0) y = gelu_fast(y);
                o[j] = y;
            }
        }
        store_h_rt<8>(out, ((long)b * P0 + t0 + t) * C0 + c, out_lo, o);
    }
}

}  // namespace advh

using namespace advh;

static int frontend_launch(const float* wave, int64_t wave_stride, int n_in, int B, int L, const float* w0,
                           const float* bias0, const float* gamma, const float* beta, int mode, int normalize, float* stats_ws,
                           float* norm_ws, float* mr_ws, void* out, int64_t out_lo, int T0, int P0, int C0, advh_stream_t stream) {
    if (!wave || !w0 || !stats_ws || !out || B <= 0 || L < K0 || n_in <= 0 || C0 <= 0 || C0 > 2048 || (C0 % 8) || 256 % (C0 / 8 < 256 ? C0 / 8 : 256)) return ADVH_EINVAL;
    if (T0 != (L - K0) / S0 + 1 || P0 < T0 || wave_stride < (n_in < L ? n_in : L)) return ADVH_EINVAL;
    if (mode == 0 && (!gamma || !beta || !norm_ws)) return ADVH_EINVAL;
    if (mode != 0 && mode != 1) return ADVH_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(wave_stats_kernel, dim3(B), dim3(256), 0, s, wave, (long)wave_stride, n_in, L, (float2*)stats_ws, normalize);
    if (mode == 0)
        hipLaunchKernelGGL(gn_stats_kernel, dim3(B), dim3(256), 0, s, wave, (long)wave_stride, n_in, L,
                           (const float2*)stats_ws, w0, gamma, beta, (float2*)norm_ws, (float2*)mr_ws, T0, C0);
    hipLaunchKernelGGL(conv0_kernel, dim3((P0 + TT - 1) / TT, B), dim3(256), 0, s, wave, (long)wave_stride, n_in, L,
                       (const float2*)stats_ws, w0, bias0, (const float2*)norm_ws, (_Float16*)out, (long)out_lo, T0, P0, C0, mode);
    return ADVH_LAUNCH_CHECK();
}

extern "C" int advh_w2v2_frontend(const float* wave, int64_t wave_stride, int n_in, int B, int L, const float* w0,
                                  const float* bias0, const float* gamma, const float* beta, int mode, int normalize, float* stats_ws,
                                  float* norm_ws, float* mr_ws, void* out, int T0, int P0, int C0, advh_stream_t stream) {
    return frontend_launch(wave, wave_stride, n_in, B, L, w0, bias0, gamma, beta, mode, normalize, stats_ws, norm_ws, mr_ws, out, 0, T0, P0, C0, stream);
}

extern "C" int advh_w2v2_frontend_split(const float* wave, int64_t wave_stride, int n_in, int B, int L, const float* w0,
                                        const float* bias0, const float* gamma, const float* beta, int mode, int normalize, float* stats_ws,
                                        float* norm_ws, float* mr_ws, void* out, int64_t out_lo, int T0, int P0, int C0, advh_stream_t stream) {
    if (out_lo <= 0 || out_lo % 8) return ADVH_EINVAL;
    return frontend_launch(wave, wave_stride, n_in, B, L, w0, bias0, gamma, beta, mode, normalize, stats_ws, norm_ws, mr_ws, out, out_lo, T0, P0, C0, stream);
}

ADVH_SPLIT_FLAG_SETTER(advh_split_flag_frontend)
