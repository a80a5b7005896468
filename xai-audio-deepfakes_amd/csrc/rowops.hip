// Row-wise (HBM-bound) kernels of the embedder: LayerNorm (+GELU), the positional-conv operand
// gather, and the time-pool + logistic-regression head.  One wavefront owns one row, loads are
// 16-byte vectors, reductions are wave shuffles -- no LDS, no atomics.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include "addvisor_hip.h"
#include "common.h"
#include "device_math.h"

namespace advh {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}


// LayerNorm over the last dim C (C % 4 == 0, C <= 64*4*MAXV): torch.nn.functional.layer_norm semantics
// (biased variance, eps inside the sqrt); two-pass in registers for accuracy.
// in: fp32 or fp16 rows, optionally + an fp16 addend row (residual + branch); out_f (fp32) and/or out_h (fp16);
// optional GELU after the affine.
template <bool IN_F32, int MAXV>
__global__ __launch_bounds__(256) void layernorm_kernel(const void* __restrict__ in, long in_ld,
                                                        const _Float16* __restrict__ add_h, long add_ld,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ out_f, _Float16* __restrict__ out_h, long out_ld,
                                                        int M, int C, float eps, int gelu, long in_lo, long add_lo, long out_lo) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float v[MAXV][4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        int c = (i * 64 + lane) * 4;
        if (c < C) {
            if (IN_F32) {
                float4 t = *(const float4*)((const float*)in + row * in_ld + c);
                v[i][0] = t.x; v[i][1] = t.y; v[i][2] = t.z; v[i][3] = t.w;
            } else {
                load_h_rt<4>((const _Float16*)in, row * in_ld + c, in_lo, v[i]);
            }
            if (add_h) {                                  // x = in + branch (the fp16 output of the preceding projection)
                float t[4];
                load_h_rt<4>(add_h, row * add_ld + c, add_lo, t);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[i][r] += t[r];
            }
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        } else {
            v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.f;
        }
    }
    const float mean = wave_sum(s) / C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        int c = (i * 64 + lane) * 4;
        if (c < C) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { float d = v[i][r] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / C + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        int c = (i * 64 + lane) * 4;
        if (c < C) {
            float4 g = *(const float4*)(gamma + c), b = *(const float4*)(beta + c);
            float o[4] = {(v[i][0] - mean) * rstd * g.x + b.x, (v[i][1] - mean) * rstd * g.y + b.y,
                          (v[i][2] - mean) * rstd * g.z + b.z, (v[i][3] - mean) * rstd * g.w + b.w};
            if (gelu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = gelu_fast(o[r]);
            }
            if (out_f) *(float4*)(out_f + row * out_ld + c) = make_float4(o[0], o[1], o[2], o[3]);
            if (out_h) store_h_rt<4>(out_h, row * out_ld + c, out_lo, o);
        }
    }
}

// h [B][T][H] fp32 -> xg [G][B][P][Cg] fp16, P = T + K, data rows [K/2, K/2 + T), zeros elsewhere
// (the zero padding of the positional Conv1d, modeling_wav2vec2.py:326-379).
__global__ __launch_bounds__(256) void posconv_gather_kernel(const float* __restrict__ h, _Float16* __restrict__ xg,
                                                             int B, int T, int H, int G, int K, int pad_left,
                                                             const _Float16* __restrict__ dact_src, long xg_lo, long dact_lo) {
    const int Cg = H / G, P = T + K, c4 = Cg / 4;
    const long total = (long)G * B * P * c4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % c4) * 4;
        long r = i / c4;
        int p = (int)(r % P);
        long gb = r / P;
        int b = (int)(gb % B), g = (int)(gb / B);
        int t = p - pad_left;
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        if (t >= 0 && t < T) {
            const long src = ((long)b * T + t) * H + g * Cg + c;
            float4 v = *(const float4*)(h + src);
            if (dact_src) {                              // backward: gradient w.r.t. the pre-GELU conv output
                float z[4];
                load_h_rt<4>(dact_src, src, dact_lo, z);
                v.x *= gelu_grad(z[0]); v.y *= gelu_grad(z[1]);
                v.z *= gelu_grad(z[2]); v.w *= gelu_grad(z[3]);
            }
            o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
        }
        store_h_rt<4>(xg, r * Cg + c, xg_lo, o);
    }
}

// logit[b] = mean_t(h[b,t,:]) . coef + intercept;  prob = sigmoid(logit)
// (LMAC_metrics.py:130,146,156 pooling + classifier_embedder.py:34-38).  One workgroup per clip.
// Fast path (H % 4 == 0, H <= 2048): wavefront w sums the frames t = w, w+4, ... with 16-byte loads (NV independent
// accumulators per lane, two frames in flight), the four partial rows meet in LDS in a fixed order -> deterministic.
template <int NV>
__global__ __launch_bounds__(256) void pool_logreg_rows_kernel(const float* __restrict__ h, const float* __restrict__ coef,
                                                               float intercept, float* __restrict__ logit,
                                                               float* __restrict__ prob, float* __restrict__ pooled,
                                                               int T, int H) {
    extern __shared__ float part[];                       // [4][H]
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* hb = h + (long)b * T * H;
    float4 acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = wv; t < T; t += 8) {
        const bool two = t + 4 < T;
        float4 a[NV], c2[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = 4 * (lane + 64 * i);
            a[i] = c < H ? *(const float4*)(hb + (long)t * H + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            c2[i] = (two && c < H) ? *(const float4*)(hb + (long)(t + 4) * H + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            acc[i].x += a[i].x; acc[i].y += a[i].y; acc[i].z += a[i].z; acc[i].w += a[i].w;
            acc[i].x += c2[i].x; acc[i].y += c2[i].y; acc[i].z += c2[i].z; acc[i].w += c2[i].w;
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = 4 * (lane + 64 * i);
        if (c < H) *(float4*)(part + wv * H + c) = acc[i];
    }
    __syncthreads();
    float dot = 0.f;
    for (int c = tid; c < H; c += 256) {
        const float s = ((part[c] + part[H + c]) + (part[2 * H + c] + part[3 * H + c])) / T;
        if (pooled) pooled[(long)b * H + c] = s;
        dot += s * coef[c];
    }
    dot = wave_sum(dot);
    if (lane == 0) red[wv] = dot;
    __syncthreads();
    if (tid == 0) {
        float z = (red[0] + red[1]) + (red[2] + red[3]) + intercept;
        logit[b] = z;
        prob[b] = 1.f / (1.f + expf(-z));
    }
}

// any H: one thread per channel, frames in order
__global__ __launch_bounds__(256) void pool_logreg_kernel(const float* __restrict__ h, const float* __restrict__ coef,
                                                          float intercept, float* __restrict__ logit,
                                                          float* __restrict__ prob, float* __restrict__ pooled,
                                                          int T, int H) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* hb = h + (long)b * T * H;
    float dot = 0.f;
    for (int c = tid; c < H; c += 256) {
        float s = 0.f;
        for (int t = 0; t < T; ++t) s += hb[(long)t * H + c];
        s /= T;
        if (pooled) pooled[(long)b * H + c] = s;
        dot += s * coef[c];
    }
    dot = wave_sum(dot);
    if ((tid & 63) == 0) red[tid >> 6] = dot;
    __syncthreads();
    if (tid == 0) {
        float z = (red[0] + red[1]) + (red[2] + red[3]) + intercept;
        logit[b] = z;
        prob[b] = 1.f / (1.f + expf(-z));
    }
}

// fp32 -> split format on the device (the training path's per-step weight refresh: unet_train.py re-packs the live fp32
// parameters; the torch formulation was a dozen fp64 element-wise launches and a host synchronisation per tensor).
__global__ __launch_bounds__(256) void split_f32_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, long lo, long n) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 4 <= n) {
        const float4 t = *(const float4*)(src + i);
        const float v[4] = {t.x, t.y, t.z, t.w};
        store_h_rt<4>(dst, i, lo, v);
    } else {
        for (long j = i; j < n; ++j) {
            _Float16 h, l;
            split_f32(src[j], h, l);
            dst[j] = h; dst[j + lo] = l;
        }
    }
}

}  // namespace advh

using namespace advh;

static int layernorm_launch(const void* in, int in_is_f32, int64_t in_ld, const void* add_h, int64_t add_ld,
                            const float* gamma, const float* beta, float* out_f, void* out_h, int64_t out_ld, int M, int C,
                            float eps, int gelu, int64_t in_lo, int64_t add_lo, int64_t out_lo, advh_stream_t stream) {
    if (!in || !gamma || !beta || (!out_f && !out_h) || M <= 0 || C <= 0 || C % 4 || in_ld % 4 || out_ld % 4 || (add_h && add_ld % 4))
        return ADVH_EINVAL;
    if (C > 64 * 4 * 8) return ADVH_EUNSUPPORTED;
    dim3 grid((M + 3) / 4), block(256);
    hipStream_t s = (hipStream_t)stream;
#define LN_LAUNCH(F32, MV)                                                                                         \
    hipLaunchKernelGGL((layernorm_kernel<F32, MV>), grid, block, 0, s, in, (long)in_ld, (const _Float16*)add_h,   \
                       (long)add_ld, gamma, beta, out_f, (_Float16*)out_h, (long)out_ld, M, C, eps, gelu, (long)in_lo, (long)add_lo, (long)out_lo)
    if (C <= 64 * 4 * 2) { if (in_is_f32) LN_LAUNCH(true, 2); else LN_LAUNCH(false, 2); }
    else if (C <= 64 * 4 * 4) { if (in_is_f32) LN_LAUNCH(true, 4); else LN_LAUNCH(false, 4); }
    else { if (in_is_f32) LN_LAUNCH(true, 8); else LN_LAUNCH(false, 8); }
#undef LN_LAUNCH
    return ADVH_LAUNCH_CHECK();
}

extern "C" int advh_layernorm_add(const void* in, int in_is_f32, int64_t in_ld, const void* add_h, int64_t add_ld,
                                  const float* gamma, const float* beta, float* out_f, void* out_h, int64_t out_ld, int M, int C,
                                  float eps, int gelu, advh_stream_t stream) {
    return layernorm_launch(in, in_is_f32, in_ld, add_h, add_ld, gamma, beta, out_f, out_h, out_ld, M, C, eps, gelu, 0, 0, 0, stream);
}

extern "C" int advh_layernorm_split(const void* in, int in_is_f32, int64_t in_ld, int64_t in_lo, const void* add_h, int64_t add_ld,
                                    int64_t add_lo, const float* gamma, const float* beta, float* out_f, void* out_h, int64_t out_ld,
                                    int64_t out_lo, int M, int C, float eps, int gelu, advh_stream_t stream) {
    if ((!in_is_f32 && in_lo <= 0) || (add_h && add_lo <= 0) || (out_h && out_lo <= 0) || in_lo % 4 || add_lo % 4 || out_lo % 4) return ADVH_EINVAL;
    return layernorm_launch(in, in_is_f32, in_ld, add_h, add_ld, gamma, beta, out_f, out_h, out_ld, M, C, eps, gelu,
                            in_is_f32 ? 0 : in_lo, add_h ? add_lo : 0, out_h ? out_lo : 0, stream);
}

extern "C" int advh_layernorm(const void* in, int in_is_f32, int64_t in_ld, const float* gamma, const float* beta,
                              float* out_f, void* out_h, int64_t out_ld, int M, int C, float eps, int gelu,
                              advh_stream_t stream) {
    return advh_layernorm_add(in, in_is_f32, in_ld, nullptr, 0, gamma, beta, out_f, out_h, out_ld, M, C, eps, gelu, stream);
}

static int posconv_gather_launch(const float* h, void* xg, int64_t xg_lo, int B, int T, int H, int G, int K, int pad_left,
                                 const void* dact_src, int64_t dact_lo, advh_stream_t stream) {
    if (!h || !xg || B <= 0 || T <= 0 || H <= 0 || G <= 0 || H % G || (H / G) % 8 || K <= 0 || K % 2 || pad_left < 0 || pad_left > K) return ADVH_EINVAL;
    long total = (long)G * B * (T + K) * (H / G / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(posconv_gather_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, h, (_Float16*)xg, B, T, H, G, K, pad_left,
                       (const _Float16*)dact_src, (long)xg_lo, (long)dact_lo);
    return ADVH_LAUNCH_CHECK();
}

extern "C" int advh_posconv_gather(const float* h, void* xg, int B, int T, int H, int G, int K, int pad_left,
                                   const void* dact_src, advh_stream_t stream) {
    return posconv_gather_launch(h, xg, 0, B, T, H, G, K, pad_left, dact_src, 0, stream);
}

extern "C" int advh_posconv_gather_split(const float* h, void* xg, int64_t xg_lo, int B, int T, int H, int G, int K, int pad_left,
                                         advh_stream_t stream) {
    if (xg_lo <= 0 || xg_lo % 4) return ADVH_EINVAL;
    return posconv_gather_launch(h, xg, xg_lo, B, T, H, G, K, pad_left, nullptr, 0, stream);
}

extern "C" int advh_posconv_gather_bwd_split(const float* dh, void* xg, int64_t xg_lo, int B, int T, int H, int G, int K, int pad_left,
                                             const void* dact_src, int64_t dact_lo, advh_stream_t stream) {
    if (xg_lo <= 0 || xg_lo % 4 || !dact_src || dact_lo <= 0 || dact_lo % 4) return ADVH_EINVAL;
    return posconv_gather_launch(dh, xg, xg_lo, B, T, H, G, K, pad_left, dact_src, dact_lo, stream);
}

extern "C" int advh_pool_logreg(const float* h, const float* coef, float intercept, float* logit, float* prob,
                                float* pooled, int B, int T, int H, advh_stream_t stream) {
    if (!h || !coef || !logit || !prob || B <= 0 || T <= 0 || H <= 0) return ADVH_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = 4 * (size_t)H * sizeof(float);
    if (H % 4 == 0 && H <= 1024)
        hipLaunchKernelGGL(pool_logreg_rows_kernel<4>, dim3(B), dim3(256), lds, s, h, coef, intercept, logit, prob, pooled, T, H);
    else if (H % 4 == 0 && H <= 2048)
        hipLaunchKernelGGL(pool_logreg_rows_kernel<8>, dim3(B), dim3(256), lds, s, h, coef, intercept, logit, prob, pooled, T, H);
    else
        hipLaunchKernelGGL(pool_logreg_kernel, dim3(B), dim3(256), 0, s, h, coef, intercept, logit, prob, pooled, T, H);
    return ADVH_LAUNCH_CHECK();
}

ADVH_SPLIT_FLAG_SETTER(advh_split_flag_rowops)

extern "C" int advh_split_f32(const float* src, void* dst, int64_t dst_lo, int64_t n, advh_stream_t stream) {
    if (!src || !dst || n <= 0 || dst_lo < n || dst_lo % 4 || ((uintptr_t)src & 15) || ((uintptr_t)dst & 7)) return ADVH_EINVAL;
    hipLaunchKernelGGL(split_f32_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, src, (_Float16*)dst, (long)dst_lo, (long)n);
    return ADVH_LAUNCH_CHECK();
}
