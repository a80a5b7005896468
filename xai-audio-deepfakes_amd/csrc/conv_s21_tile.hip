// LDS line-tile kernel for e2.block.0 of the U-Net: Conv2d(32, 64, (5, 3), stride (2, 1), padding (2, 1)) + folded BatchNorm +
// LeakyReLU (addvisor.py:32 with ConvBlock :12-25).  K = 15 taps x 32 channels = 480 and 64 output channels: as an implicit
// GEMM on the 256 x 64 tile every output re-reads its operand row once per tap through L2 (258 us, 390 TFLOP/s, 2.8x its
// HBM time).  Here the 15 x 64 x 32 fp16 weights (60 KB) stay resident in LDS and a persistent workgroup streams 8 x 16
// output tiles through a double-buffered 19 x 18 input patch (21.5 KB), so HBM traffic = input once (+ halo) + output once.
// Weights are the MFMA A operand (rows = output channels, host-permuted so a lane's two accumulator tiles of a pair are 8
// consecutive channels), positions the B operand; wavefront w owns output rows 2w, 2w+1 of the tile.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include "addvisor_hip.h"
#include "common.h"

namespace advh {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define DS_READ128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define LGKM_WAIT(n) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory")

constexpr int S21_CI = 32, S21_CO = 64, S21_KH = 5, S21_KW = 3, S21_NT = S21_KH * S21_KW;   // 15 taps = 15 k-steps of 32
constexpr int S21_TY = 8, S21_TX = 16;                                                        // output tile
constexpr int S21_PRW = 2 * (S21_TY - 1) + S21_KH, S21_PCL = S21_TX + S21_KW - 1;             // patch 19 x 18 pixels
constexpr int S21_SLOTS = 4, S21_PITCH = S21_SLOTS * 16;                                      // 64-byte pixels, chunk c at slot c ^ ((pixel >> 1) & 2)
constexpr int S21_WBYTES = S21_NT * S21_CO * 64;                                              // 61 440
constexpr int S21_PCH = (S21_PRW * S21_PCL * S21_SLOTS + 63) & ~63;                           // patch chunks (whole-wave loads)

__global__ __launch_bounds__(256) void conv53s21_tile_kernel(const advh_convs21_desc p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int fr = lane & 15, g = lane >> 4;
    char* Wl = lds;
    char* Xl = lds + S21_WBYTES;
    // weights [15][64 rows (host-permuted)][32]: 64-byte LDS rows, chunk c of row r at slot c ^ ((r >> 1) & 2)
    const _Float16* Wg = (const _Float16*)p.W;
    for (int i = tid; i < S21_WBYTES / 16; i += 256) {
        const int row = i >> 2, pos = i & 3;
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(Wg + (long)row * 32 + ((pos ^ ((row >> 1) & 2)) * 8)),
                                         LDS_PTR(Wl + (size_t)(i - lane) * 16), 16, 0, 0);
    }
    const _Float16* X = (const _Float16*)p.X;
    const int Hi = 2 * p.Ho;                                         // input rows (stride 2, "same"-style padding 2)
    const int Hpi = Hi + 2 * p.PHi, Wpi = p.W_ + 2 * p.PWi, Hpo = p.Ho + 2 * p.PHo, Wpo = p.W_ + 2 * p.PWo;
    const int tx = (p.W_ + S21_TX - 1) / S21_TX, ty = (p.Ho + S21_TY - 1) / S21_TY, ntiles = p.B * ty * tx;
    auto origin = [&](int tile, int& b, int& y0, int& x0) {
        x0 = (tile % tx) * S21_TX;
        const int r = tile / tx;
        y0 = (r % ty) * S21_TY;
        b = r / ty;
    };
    auto load_patch = [&](int tile, int buf) {
        int b, y0, x0;
        origin(tile, b, y0, x0);
        char* dst = Xl + (size_t)buf * S21_PCH * 16;
        for (int i = tid; i < S21_PCH; i += 256) {
            int pix = i / S21_SLOTS;
            const int slot = i - pix * S21_SLOTS;
            if (pix >= S21_PRW * S21_PCL) pix = 0;
            // patch row 0 = input row 2 y0 - 2, column 0 = x0 - 1 (padded coordinates, clamped: clamped pixels only feed skipped outputs)
            const int gy = min(2 * y0 + p.PHi - 2 + pix / S21_PCL, Hpi - 1), gx = min(x0 + p.PWi - 1 + pix % S21_PCL, Wpi - 1);
            const _Float16* src = X + (((long)b * Hpi + gy) * Wpi + gx) * S21_CI + ((slot ^ (((i / S21_SLOTS) >> 1) & 2)) * 8);
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(dst + (size_t)(i - lane) * 16), 16, 0, 0);
        }
    };
    float4 bias[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        bias[i] = p.bias ? *(const float4*)(p.bias + (i >> 1) * 32 + g * 8 + (i & 1) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned lds0 = (unsigned)(unsigned long)LDS_PTR(lds);
    const unsigned wbase = lds0 + (unsigned)((fr * 4 + (g ^ ((fr >> 1) & 2))) * 16);
    int buf = 0;
    if ((int)blockIdx.x < ntiles) load_patch(blockIdx.x, 0);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, buf ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) load_patch(tile + gridDim.x, buf ^ 1);
        const unsigned xl = lds0 + S21_WBYTES + (unsigned)buf * S21_PCH * 16;
        f32x4 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto xaddr = [&](int s, int j) {                             // tap s = kh * 3 + kw, output row 2 wv + j of the tile
            const int kh = s / S21_KW, kw = s - kh * S21_KW;
            const int pix = (2 * (2 * wv + j) + kh) * S21_PCL + kw + fr;
            return xl + (unsigned)(pix * S21_PITCH + ((g ^ ((pix >> 1) & 2)) * 16));
        };
        f16x8 wf[2][4], xf[2][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) DS_READ128(wf[0][i], wbase, i * 16 * 64);
#pragma unroll
        for (int j = 0; j < 2; ++j) { const unsigned a = xaddr(0, j); DS_READ128(xf[0][j], a, 0); }
#pragma unroll
        for (int s = 0; s < S21_NT; ++s) {                          // fully unrolled: static register double buffer
            const int cur = s & 1, nxt = cur ^ 1;
            const unsigned wn = wbase + (unsigned)((s + 1) * S21_CO * 64);
            LGKM_WAIT(0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int i = m >> 1, j = m & 1;
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[cur][i], xf[cur][j], acc[i][j], 0, 0, 0);
                if (s + 1 < S21_NT) {
                    if (m < 4) DS_READ128(wf[nxt][m < 4 ? m : 0], wn, (m < 4 ? m : 0) * 16 * 64);
                    else if (m < 6) { const unsigned a = xaddr(s + 1, m - 4 < 2 ? m - 4 : 0); DS_READ128(xf[nxt][m - 4 < 2 ? m - 4 : 0], a, 0); }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        int b, y0, x0;
        origin(tile, b, y0, x0);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gy = y0 + 2 * wv + j, gx = x0 + fr;
            if (gy >= p.Ho || gx >= p.W_) continue;
            const long pos = ((long)b * Hpo + gy + p.PHo) * Wpo + gx + p.PWo;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const long o = pos * S21_CO + q * 32 + g * 8;
                float v[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[r] = acc[2 * q][j][r]; v[4 + r] = acc[2 * q + 1][j][r]; }
                v[0] += bias[2 * q].x; v[1] += bias[2 * q].y; v[2] += bias[2 * q].z; v[3] += bias[2 * q].w;
                v[4] += bias[2 * q + 1].x; v[5] += bias[2 * q + 1].y; v[6] += bias[2 * q + 1].z; v[7] += bias[2 * q + 1].w;
                f16x8 hv;
#pragma unroll
                for (int r = 0; r < 8; ++r) hv[r] = (_Float16)((p.act == ADVH_ACT_LEAKY && v[r] < 0.f) ? p.slope * v[r] : v[r]);
                *(f16x8*)((_Float16*)p.out_h + o) = hv;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace advh

using namespace advh;

extern "C" int advh_conv53s21_tile_lds_bytes(void) { return S21_WBYTES + 2 * S21_PCH * 16; }

extern "C" int advh_conv53s21_tile_f16(const advh_convs21_desc* d, advh_stream_t stream) {
    if (!d || !d->X || !d->W || !d->out_h || d->B <= 0 || d->Ho <= 0 || d->W_ <= 0) return ADVH_EINVAL;
    if (d->PHi < 2 || d->PWi < 1 || d->PHo < 0 || d->PWo < 0) return ADVH_EINVAL;
    if (d->act != ADVH_ACT_NONE && d->act != ADVH_ACT_LEAKY) return ADVH_EINVAL;
    if (advh_ensure_lds((const void*)conv53s21_tile_kernel) != ADVH_OK) return ADVH_ELAUNCH;
    const long ntiles = (long)d->B * ((d->Ho + S21_TY - 1) / S21_TY) * ((d->W_ + S21_TX - 1) / S21_TX);
    const long grid = ntiles < 256 ? ntiles : 256;
    hipLaunchKernelGGL(conv53s21_tile_kernel, dim3((unsigned)grid), dim3(256), S21_WBYTES + 2 * S21_PCH * 16, (hipStream_t)stream, *d);
    return ADVH_LAUNCH_CHECK();
}
