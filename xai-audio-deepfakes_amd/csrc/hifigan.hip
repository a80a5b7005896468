// HiFi-GAN V1 generator: the pieces that are not GEMM-shaped.  (The Conv1d / ConvTranspose1d layers run
// on the implicit-GEMM kernel: gemm.plan_conv1d_same / plan_convT1d.)  Reference handle: hifigan.py:106-110,
// 163-180 (SpeechBrain HIFIGAN.decode_batch + mel_spectogram); architecture per Kong et al. 2020, config V1.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include "addvisor_hip.h"
#include "common.h"
#include "device_math.h"

namespace advh {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// mel [B][C][T] fp32 (torch layout) -> zero-haloed channels-last fp16 [B][T+2*halo][C]   (interior only)
__global__ __launch_bounds__(256) void pack_mel_kernel(const float* __restrict__ mel, _Float16* __restrict__ out,
                                                       int C, int T, int halo, long total, long out_lo) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    int c = (int)(i % C);
    long r = i / C;
    int t = (int)(r % T), b = (int)(r / T);
    const float v[1] = {mel[((long)b * C + c) * T + t]};
    store_h_rt<1>(out, ((long)b * (T + 2 * halo) + t + halo) * C + c, out_lo, v);
}

// Same with `pad` replicated frames in front and behind (SpeechBrain / Coqui `inference_padding`: F.pad(mel, (p, p),
// "replicate") before the generator); the map holds T + 2 pad interior rows.
__global__ __launch_bounds__(256) void pack_mel_pad_kernel(const float* __restrict__ mel, _Float16* __restrict__ out,
                                                           int C, int T, int pad, int halo, long total, long out_lo) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int Tp = T + 2 * pad;
    int c = (int)(i % C);
    long r = i / C;
    int t = (int)(r % Tp), b = (int)(r / Tp);
    int ts = min(max(t - pad, 0), T - 1);
    const float v[1] = {mel[((long)b * C + c) * T + ts]};
    store_h_rt<1>(out, ((long)b * (Tp + 2 * halo) + t + halo) * C + c, out_lo, v);
}

// Halo fill of a channels-last map [B][T + 2 halo][C]: mode 0 = zeros, mode 1 = reflection about the first / last
// sample (x[-j] = x[j], x[T-1+j] = x[T-1-j]; torch "reflect" padding), for j = 1 .. min(halo, T - 1); rows beyond stay zero.
// Lets the "same" Conv1d layers read a reflect-padded input (SpeechBrain's Conv1d default padding_mode) from the map itself.
__global__ __launch_bounds__(256) void halo_fill_kernel(_Float16* __restrict__ x, int T, int C, int halo, int mode, long total) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c8 = C / 8;
    int c = (int)(i % c8);
    long r = i / c8;
    int j = (int)(r % (2 * halo)), b = (int)(r / (2 * halo));
    const long base = (long)b * (T + 2 * halo);
    int dst, src;
    if (j < halo) { dst = halo - 1 - j; src = halo + 1 + j; }               // x[-(j+1)] <- x[j+1]
    else { const int jj = j - halo; dst = halo + T + jj; src = halo + T - 2 - jj; }
    f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    const int dist = j < halo ? j + 1 : j - halo + 1;
    if (mode == 1 && dist <= T - 1) v = *(const f16x8*)(x + (base + src) * C + c * 8);
    *(f16x8*)(x + (base + dst) * C + c * 8) = v;
}

// MRF mix: y = LeakyReLU_slope((a + b + c) / 3), whole padded maps (zero halo stays zero)
__global__ __launch_bounds__(256) void mrf_mix_kernel(const _Float16* __restrict__ a, const _Float16* __restrict__ b,
                                                      const _Float16* __restrict__ c, _Float16* __restrict__ y, float slope, long n8,
                                                      long lo) {     // lo != 0: split-format maps (plane pairs, lo plane `lo` elements behind)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        float va[8], vb[8], vc[8], o[8];
        load_h_rt<8>(a, i * 8, lo, va);
        load_h_rt<8>(b, i * 8, lo, vb);
        load_h_rt<8>(c, i * 8, lo, vc);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = (va[j] + vb[j] + vc[j]) * (1.f / 3.f);
            o[j] = v > 0.f ? v : slope * v;
        }
        store_h_rt<8>(y, i * 8, lo, o);
    }
}

// conv_post: Conv1d(C -> 1, k, "same") + tanh on a pre-activated zero-haloed map; wav [B][1][T] fp32.
// One workgroup = 256 consecutive output samples of one clip: the 256 + k - 1 rows they touch are staged ONCE in LDS as fp32
// (plain fp16 rows or joined split planes), pitch C + 4 words so that the per-thread float4 reads of consecutive rows are
// conflict-free; every thread then runs its k * C multiply-adds out of LDS with the weights in scalar registers.  (Round 2's
// form -- every thread loading its own k rows from global memory, both planes in the fp32-class mode -- took 6.75 ms per 256-clip
// batch, 13x its HBM time: profiles/r03_hifigan_f32_kernel_summary.txt.)
constexpr int CP_TO = 256, CP_MAXK = 15;
__global__ __launch_bounds__(256) void conv_post_kernel(const _Float16* __restrict__ x, const float* __restrict__ w /*[k][C]*/,
                                                        float bias, float* __restrict__ wav, int C, int T, int halo, int k, long x_lo) {
    extern __shared__ __attribute__((aligned(16))) float xs[];            // [CP_TO + k - 1][C + 4]
    const int tid = threadIdx.x, b = blockIdx.y, t0 = blockIdx.x * CP_TO;
    const int pitch = C + 4, rows = CP_TO + k - 1, ch = C / 8;
    const long P = T + 2 * halo;
    const long r0 = (long)b * P + t0 + halo - (k - 1) / 2;                // map row under tap 0 of output t0 (always inside the clip's padded rows)
    for (int i = tid; i < rows * ch; i += 256) {
        const int row = i / ch, c = i % ch;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (t0 + row < T + (k - 1)) load_h_rt<8>(x, (r0 + row) * C + c * 8, x_lo, v);     // rows past the clip's halo only feed outputs t >= T
        *(float4*)(xs + row * pitch + c * 8) = make_float4(v[0], v[1], v[2], v[3]);
        *(float4*)(xs + row * pitch + c * 8 + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
    __syncthreads();
    const int t = t0 + tid;
    if (t >= T) return;
    float acc = bias;
    for (int j = 0; j < k; ++j) {
        const float* xr = xs + (tid + j) * pitch;
        const float* wr = w + j * C;                                       // uniform: scalar loads
        for (int c = 0; c < C; c += 4) {
            const float4 xv = *(const float4*)(xr + c);
            acc = fmaf(xv.x, wr[c], acc); acc = fmaf(xv.y, wr[c + 1], acc); acc = fmaf(xv.z, wr[c + 2], acc); acc = fmaf(xv.w, wr[c + 3], acc);
        }
    }
    wav[(long)b * T + t] = tanhf(acc);
}

// log-mel: out[b][m][t] = log(max(sum_f fb[f][m] * mag[b][f][t], 1e-5))   (hifigan.py:163-178)
__global__ __launch_bounds__(256) void mel_log_kernel(const float* __restrict__ mag, const float* __restrict__ fb,
                                                      float* __restrict__ out, int F, int T, int NM, long total) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    int t = (int)(i % T);
    long r = i / T;
    int m = (int)(r % NM), b = (int)(r / NM);
    const float* mg = mag + (long)b * F * T + t;
    float acc = 0.f;
    for (int f = 0; f < F; ++f) acc = fmaf(fb[(long)f * NM + m], mg[(long)f * T], acc);
    out[i] = logf(fmaxf(acc, 1e-5f));
}

}  // namespace advh

using namespace advh;

static int pack_mel_launch(const float* mel, void* out, int64_t out_lo, int B, int C, int T, int pad, int halo, advh_stream_t stream) {
    if (!mel || !out || B <= 0 || C <= 0 || T <= 0 || halo < 0 || pad < 0 || out_lo < 0) return ADVH_EINVAL;
    if (pad == 0) {
        long total = (long)B * C * T;
        hipLaunchKernelGGL(pack_mel_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mel, (_Float16*)out, C, T, halo, total,
                           (long)out_lo);
    } else {
        long total = (long)B * C * (T + 2 * pad);
        hipLaunchKernelGGL(pack_mel_pad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mel, (_Float16*)out, C, T, pad,
                           halo, total, (long)out_lo);
    }
    return ADVH_LAUNCH_CHECK();
}

extern "C" int advh_hifigan_pack_mel(const float* mel, void* out, int B, int C, int T, int halo, advh_stream_t stream) {
    return pack_mel_launch(mel, out, 0, B, C, T, 0, halo, stream);
}

extern "C" int advh_hifigan_pack_mel_split(const float* mel, void* out, int64_t out_lo, int B, int C, int T, int pad, int halo,
                                           advh_stream_t stream) {
    if (out_lo <= 0) return ADVH_EINVAL;
    return pack_mel_launch(mel, out, out_lo, B, C, T, pad, halo, stream);
}

extern "C" int advh_hifigan_pack_mel_pad(const float* mel, void* out, int B, int C, int T, int pad, int halo, advh_stream_t stream) {
    return pack_mel_launch(mel, out, 0, B, C, T, pad, halo, stream);
}

extern "C" int advh_halo_fill_f16(void* x, int B, int T, int C, int halo, int mode, advh_stream_t stream) {
    if (!x || B <= 0 || T <= 0 || C <= 0 || C % 8 || halo < 0 || (mode != 0 && mode != 1)) return ADVH_EINVAL;
    if (halo == 0) return ADVH_OK;
    long total = (long)B * 2 * halo * (C / 8);
    hipLaunchKernelGGL(halo_fill_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (_Float16*)x, T, C, halo, mode, total);
    return ADVH_LAUNCH_CHECK();
}

static int mrf_mix_launch(const void* a, const void* b, const void* c, void* y, float slope, int64_t numel, int64_t lo, advh_stream_t stream) {
    if (!a || !b || !c || !y || numel <= 0 || numel % 8 || lo < 0 || lo % 8) return ADVH_EINVAL;
    long n8 = numel / 8;
    long blocks = (n8 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(mrf_mix_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const _Float16*)a, (const _Float16*)b,
                       (const _Float16*)c, (_Float16*)y, slope, n8, (long)lo);
    return ADVH_LAUNCH_CHECK();
}

extern "C" int advh_hifigan_mrf_mix(const void* a, const void* b, const void* c, void* y, float slope, int64_t numel, advh_stream_t stream) {
    return mrf_mix_launch(a, b, c, y, slope, numel, 0, stream);
}

extern "C" int advh_hifigan_mrf_mix_split(const void* a, const void* b, const void* c, void* y, float slope, int64_t numel, int64_t lo,
                                          advh_stream_t stream) {
    if (lo <= 0) return ADVH_EINVAL;
    return mrf_mix_launch(a, b, c, y, slope, numel, lo, stream);
}

static int conv_post_launch(const void* x, int64_t x_lo, const float* w, float bias, float* wav, int B, int C, int T, int halo, int k,
                            advh_stream_t stream) {
    if (!x || !w || !wav || B <= 0 || C <= 0 || C % 8 || T <= 0 || k <= 0 || !(k & 1) || halo < (k - 1) / 2 || x_lo < 0) return ADVH_EINVAL;
    if (k > CP_MAXK || C > 64) return ADVH_EUNSUPPORTED;
    const size_t lds = (size_t)(CP_TO + k - 1) * (C + 4) * sizeof(float);
    if (lds > 64 * 1024 && advh_ensure_lds((const void*)conv_post_kernel) != ADVH_OK) return ADVH_ELAUNCH;
    hipLaunchKernelGGL(conv_post_kernel, dim3((unsigned)((T + CP_TO - 1) / CP_TO), B), dim3(256), lds, (hipStream_t)stream, (const _Float16*)x, w,
                       bias, wav, C, T, halo, k, (long)x_lo);
    return ADVH_LAUNCH_CHECK();
}

extern "C" int advh_hifigan_conv_post(const void* x, const float* w, float bias, float* wav, int B, int C, int T, int halo, int k,
                                      advh_stream_t stream) {
    return conv_post_launch(x, 0, w, bias, wav, B, C, T, halo, k, stream);
}

extern "C" int advh_hifigan_conv_post_split(const void* x, int64_t x_lo, const float* w, float bias, float* wav, int B, int C, int T, int halo,
                                            int k, advh_stream_t stream) {
    if (x_lo <= 0) return ADVH_EINVAL;
    return conv_post_launch(x, x_lo, w, bias, wav, B, C, T, halo, k, stream);
}

extern "C" int advh_mel_log(const float* mag, const float* fb, float* out, int B, int F, int T, int n_mels, advh_stream_t stream) {
    if (!mag || !fb || !out || B <= 0 || F <= 0 || T <= 0 || n_mels <= 0) return ADVH_EINVAL;
    long total = (long)B * n_mels * T;
    hipLaunchKernelGGL(mel_log_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mag, fb, out, F, T, n_mels, total);
    return ADVH_LAUNCH_CHECK();
}

ADVH_SPLIT_FLAG_SETTER(advh_split_flag_hifigan)
