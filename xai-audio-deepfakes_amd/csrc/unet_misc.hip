// The three U-Net layers that are not GEMM-shaped (addvisor.py:27-84): the 1-channel stem
// convolution, the copy of the input magnitude into the last skip-concat buffer, and the
// 1x1 mask head + sigmoid.  All HBM-bound, one thread per spatial position, 16-byte vectors on the
// channels-last side.  H = frequency bins, W = frames; the magnitude arrives as torch's
// [B][F][T] fp32 (t fastest), cropped to H x W by indexing, never copied.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include "addvisor_hip.h"
#include "common.h"
#include "device_math.h"

namespace advh {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// e1.block.0: Conv2d(1, 32, (5,3), stride (2,1), padding (2,1)) with BatchNorm folded, LeakyReLU(0.2)
// (addvisor.py:31, 15-17).  out: zero-haloed NHWC fp16 [B][Ho+2PH][W+2PW][32], interior written.
__global__ __launch_bounds__(256) void unet_stem_kernel(const float* __restrict__ mag, int Fq, int Tq, int H, int W,
                                                        const float* __restrict__ wgt /*[32][15]*/,
                                                        const float* __restrict__ bias, _Float16* __restrict__ out,
                                                        int PH, int PW, float slope, long total, long out_lo) {
    __shared__ float ws[32 * 15 + 32];
    for (int i = threadIdx.x; i < 32 * 15 + 32; i += 256) ws[i] = i < 480 ? wgt[i] : bias[i - 480];
    __syncthreads();
    const int Ho = H / 2;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    int w = (int)(i % W);
    long r = i / W;
    int ho = (int)(r % Ho), b = (int)(r / Ho);
    float x[15];
#pragma unroll
    for (int kh = 0; kh < 5; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            int h = 2 * ho + kh - 2, ww = w + kw - 1;
            x[kh * 3 + kw] = (h >= 0 && h < H && ww >= 0 && ww < W) ? mag[((long)b * Fq + h) * Tq + ww] : 0.f;
        }
    const long o = (((long)b * (Ho + 2 * PH) + ho + PH) * (W + 2 * PW) + w + PW) * 32;
#pragma unroll
    for (int c8 = 0; c8 < 4; ++c8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int c = c8 * 8 + j;
            float y = ws[480 + c];
#pragma unroll
            for (int k = 0; k < 15; ++k) y = fmaf(ws[c * 15 + k], x[k], y);
            v[j] = y > 0.f ? y : slope * y;
        }
        store_h_rt<8>(out, o + c8 * 8, out_lo, v);
    }
}

// channels [c0, c0+8) of the d1 skip-concat buffer <- (x, 1, 0, 0, 0, 0, 0, 0)   (torch.cat([y1, x]), addvisor.py:79).
// The 1 is the in-image indicator of the fused up-convolution (zero in the halo, which is never written); the unfused
// path pairs that channel with zero weights.
__global__ __launch_bounds__(256) void unet_pack_x_kernel(const float* __restrict__ mag, int Fq, int Tq, int H, int W,
                                                          _Float16* __restrict__ cat, int C, int c0, int PH, int PW, long total, long cat_lo) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    int w = (int)(i % W);
    long r = i / W;
    int h = (int)(r % H), b = (int)(r / H);
    const float v[8] = {mag[((long)b * Fq + h) * Tq + w], 1.0f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    store_h_rt<8>(cat, (((long)b * (H + 2 * PH) + h + PH) * (W + 2 * PW) + w + PW) * C + c0, cat_lo, v);
}

// mask[b][h][w] = sigmoid(sum_c y[b,h,w,c] * w[c] + bias)   (mask_head, addvisor.py:57-60); fp32 out, w fastest.
// Four lanes per position, 8 channels each: a wavefront's loads are 16 positions x 64 contiguous bytes per plane (one thread
// per position read 64 bytes at a 64-byte lane stride: 0.3 TB/s); the four partial sums meet in two xor-shuffles.
__global__ __launch_bounds__(256) void unet_head_kernel(const _Float16* __restrict__ y, int H, int W, int PH, int PW,
                                                        const float* __restrict__ wgt, float bias,
                                                        float* __restrict__ mask, float* __restrict__ logits, long total, long y_lo) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const int c8 = (int)(t & 3);
    const long i = min(t >> 2, total - 1);                 // the tail lanes repeat the last position (shuffles need every lane)
    int w = (int)(i % W);
    long r = i / W;
    int h = (int)(r % H), b = (int)(r / H);
    const long p = (((long)b * (H + 2 * PH) + h + PH) * (W + 2 * PW) + w + PW) * 32;
    float v[8];
    load_h_rt<8>(y, p + c8 * 8, y_lo, v);
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = fmaf(v[j], wgt[c8 * 8 + j], acc);
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += bias;
    if (c8 == 0 && (t >> 2) < total) {
        if (logits) logits[i] = acc;
        mask[i] = 1.f / (1.f + expf(-acc));
    }
}

}  // namespace advh

using namespace advh;

static int stem_launch(const float* mag, int Fq, int Tq, int B, int H, int W, const float* wgt, const float* bias,
                       void* out, int64_t out_lo, int PH, int PW, float slope, advh_stream_t stream) {
    if (!mag || !wgt || !bias || !out || B <= 0 || H <= 0 || (H & 1) || W <= 0 || H > Fq || W > Tq || PH < 0 || PW < 0) return ADVH_EINVAL;
    long total = (long)B * (H / 2) * W;
    hipLaunchKernelGGL(unet_stem_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mag, Fq, Tq, H, W,
                       wgt, bias, (_Float16*)out, PH, PW, slope, total, (long)out_lo);
    return ADVH_LAUNCH_CHECK();
}
extern "C" int advh_unet_stem(const float* mag, int Fq, int Tq, int B, int H, int W, const float* wgt, const float* bias,
                              void* out, int PH, int PW, float slope, advh_stream_t stream) {
    return stem_launch(mag, Fq, Tq, B, H, W, wgt, bias, out, 0, PH, PW, slope, stream);
}
extern "C" int advh_unet_stem_split(const float* mag, int Fq, int Tq, int B, int H, int W, const float* wgt, const float* bias,
                                    void* out, int64_t out_lo, int PH, int PW, float slope, advh_stream_t stream) {
    if (out_lo <= 0 || out_lo % 8) return ADVH_EINVAL;
    return stem_launch(mag, Fq, Tq, B, H, W, wgt, bias, out, out_lo, PH, PW, slope, stream);
}

static int pack_x_launch(const float* mag, int Fq, int Tq, int B, int H, int W, void* cat, int64_t cat_lo, int C, int c0, int PH, int PW,
                         advh_stream_t stream) {
    if (!mag || !cat || B <= 0 || H <= 0 || W <= 0 || H > Fq || W > Tq || C % 8 || c0 % 8 || c0 + 8 > C) return ADVH_EINVAL;
    long total = (long)B * H * W;
    hipLaunchKernelGGL(unet_pack_x_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mag, Fq, Tq, H, W,
                       (_Float16*)cat, C, c0, PH, PW, total, (long)cat_lo);
    return ADVH_LAUNCH_CHECK();
}
extern "C" int advh_unet_pack_x(const float* mag, int Fq, int Tq, int B, int H, int W, void* cat, int C, int c0, int PH, int PW,
                                advh_stream_t stream) {
    return pack_x_launch(mag, Fq, Tq, B, H, W, cat, 0, C, c0, PH, PW, stream);
}
extern "C" int advh_unet_pack_x_split(const float* mag, int Fq, int Tq, int B, int H, int W, void* cat, int64_t cat_lo, int C, int c0,
                                      int PH, int PW, advh_stream_t stream) {
    if (cat_lo <= 0 || cat_lo % 8) return ADVH_EINVAL;
    return pack_x_launch(mag, Fq, Tq, B, H, W, cat, cat_lo, C, c0, PH, PW, stream);
}

static int head_launch(const void* y, int64_t y_lo, int B, int H, int W, int PH, int PW, const float* wgt, float bias, float* mask,
                       float* logits, advh_stream_t stream) {
    if (!y || !wgt || !mask || B <= 0 || H <= 0 || W <= 0) return ADVH_EINVAL;
    long total = (long)B * H * W;
    hipLaunchKernelGGL(unet_head_kernel, dim3((unsigned)((4 * total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)y, H, W,
                       PH, PW, wgt, bias, mask, logits, total, (long)y_lo);
    return ADVH_LAUNCH_CHECK();
}
extern "C" int advh_unet_head(const void* y, int B, int H, int W, int PH, int PW, const float* wgt, float bias, float* mask,
                              float* logits, advh_stream_t stream) {
    return head_launch(y, 0, B, H, W, PH, PW, wgt, bias, mask, logits, stream);
}
extern "C" int advh_unet_head_split(const void* y, int64_t y_lo, int B, int H, int W, int PH, int PW, const float* wgt, float bias,
                                    float* mask, float* logits, advh_stream_t stream) {
    if (y_lo <= 0 || y_lo % 8) return ADVH_EINVAL;
    return head_launch(y, y_lo, B, H, W, PH, PW, wgt, bias, mask, logits, stream);
}

ADVH_SPLIT_FLAG_SETTER(advh_split_flag_unet_misc)
