// LDS line-tile kernel for the LAST decoder stage of the U-Net: up1 = ConvTranspose2d(64, 32, (2,1), stride (2,1)) folded
// into d1.block.0 = Conv2d(33, 32, 3, padding 1) + BatchNorm + LeakyReLU (addvisor.py:53-54,78-80; the composition is
// gemm.plan_upconv2d's).  With 32 output channels the implicit GEMM re-reads its operand rows once per
// tap through L2 (34 KB per MFLOP); here a persistent workgroup keeps BOTH row parities' composed weights (2 x 15 k-steps
// x 32 x 32 fp16 = 60 KB) in LDS, streams 16 x 16 output tiles through a double-buffered pair of patches -- 10 x 18
// pixels of the 64-channel coarse map and 18 x 18 pixels of the 8-channel map (x, in-image indicator, 0...) -- and feeds
// the matrix cores from LDS only: HBM traffic = coarse map once (+ halo), skip map once, output once.
//
// K layout per parity (the same order as gemm.plan_upconv2d, zero-padded from 456 to 480):
//   k-steps 0..11 : coarse tap t = s / 2 = 3 ti + tj (rows ((y+1)>>1) + ti of the patch, columns x + tj), channels 32 (s & 1) ..
//   k-steps 12..14: fine taps 4 (s - 12) + g of the 8-channel map (lane group g = one tap; taps 9..11 have zero weights)
// Wavefronts 0,1 compute the even output rows of the tile, 2,3 the odd ones, so a wavefront reads one parity's weights.
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

constexpr int UC_N = 32, UC_C0 = 64, UC_NS = 15;                   // output channels, coarse channels, k-steps per parity
constexpr int UC_WBYTES = 2 * UC_NS * UC_N * 64;                   // 61 440
constexpr int UC_PC = 10 * 18, UC_PS = 18 * 18;                    // patch positions: coarse, skip
constexpr int UC_CCH = (UC_PC * 8 + 63) & ~63, UC_SCH = (UC_PS + 63) & ~63;   // 16-byte chunks per patch (whole-wave loads)
constexpr int UC_BUF = (UC_CCH + UC_SCH) * 16;                     // one patch pair: 29 696 bytes

__global__ __launch_bounds__(256) void upconv21_tile_kernel(const advh_upconv_desc p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int fr = lane & 15, g = lane >> 4;
    char* Wl = lds;
    char* Xl = lds + UC_WBYTES;
    // ---- weights: [2][15][32 rows (host-permuted)][32 k], 64-byte LDS rows, chunk c of row r at slot c ^ ((r >> 1) & 2)
    const _Float16* Wg = (const _Float16*)p.W;
    for (int i = tid; i < UC_WBYTES / 16; i += 256) {
        const int row = i >> 2, pos = i & 3;
        const _Float16* src = Wg + (long)row * 32 + ((pos ^ ((row >> 1) & 2)) * 8);
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(Wl + (size_t)(i - lane) * 16), 16, 0, 0);
    }
    const _Float16* Xc = (const _Float16*)p.Xc;
    const _Float16* Xs = (const _Float16*)p.Xs;
    const int Hf = 2 * p.Hc;
    const int Hpc = p.Hc + 2 * p.PHc, Wpc = p.W_ + 2 * p.PWc, Hps = Hf + 2 * p.PHs, Wps = p.W_ + 2 * p.PWs;
    const int Hpo = Hf + 2 * p.PHo, Wpo = p.W_ + 2 * p.PWo;
    const int tx = (p.W_ + 15) / 16, ty = (Hf + 15) / 16, ntiles = p.B * ty * tx;
    auto origin = [&](int tile, int& b, int& y0, int& x0) {
        x0 = (tile % tx) * 16;
        const int r = tile / tx;
        y0 = (r % ty) * 16;
        b = r / ty;
    };
    auto load_patches = [&](int tile, int buf) {
        int b, y0, x0;
        origin(tile, b, y0, x0);
        char* dc = Xl + (size_t)buf * UC_BUF;
        char* ds = dc + UC_CCH * 16;
        for (int i = tid; i < UC_CCH; i += 256) {                  // coarse: rows y0/2 - 1 .. + 9, columns x0 - 1 .. + 17 (clamped: finite filler)
            int row = i >> 3;
            const int pos = i & 7;
            if (row >= UC_PC) row = 0;
            const int gy = min((y0 >> 1) + p.PHc - 1 + row / 18, Hpc - 1), gx = min(x0 + p.PWc - 1 + row % 18, Wpc - 1);
            const _Float16* src = Xc + (((long)b * Hpc + gy) * Wpc + gx) * UC_C0 + ((pos ^ ((i >> 3) & 7)) * 8);
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(dc + (size_t)(i - lane) * 16), 16, 0, 0);
        }
        for (int i = tid; i < UC_SCH; i += 256) {                  // skip (8 channels = one chunk per pixel): rows y0 - 1 .. + 17
            const int row = i < UC_PS ? i : 0;
            const int gy = min(y0 + p.PHs - 1 + row / 18, Hps - 1), gx = min(x0 + p.PWs - 1 + row % 18, Wps - 1);
            const _Float16* src = Xs + (((long)b * Hps + gy) * Wps + gx) * 8;
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(ds + (size_t)(i - lane) * 16), 16, 0, 0);
        }
    };
    const int ph = wv >> 1, a0 = 4 * (wv & 1);                     // this wavefront: output rows y = 2 (a0 + j) + ph, j = 0..3
    const float4 b0 = p.bias ? *(const float4*)(p.bias + g * 8) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 b1 = p.bias ? *(const float4*)(p.bias + g * 8 + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned lds0 = (unsigned)(unsigned long)LDS_PTR(lds);
    const unsigned wbase = lds0 + (unsigned)ph * (UC_NS * UC_N * 64) + (fr * 4 + (g ^ ((fr >> 1) & 2))) * 16;
    int buf = 0;
    if ((int)blockIdx.x < ntiles) load_patches(blockIdx.x, 0);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, buf ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) load_patches(tile + gridDim.x, buf ^ 1);
        const unsigned xc = lds0 + UC_WBYTES + (unsigned)buf * UC_BUF, xs = xc + UC_CCH * 16;
        f32x4 acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto addr = [&](int s, unsigned (&xa)[4]) {
            if (s < 12) {
                const int t = s >> 1, ti = t / 3, tj = t - ti * 3, c = (s & 1) * 4 + g;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = (a0 + j + ph + ti) * 18 + tj + fr;
                    xa[j] = xc + (row * 8 + (c ^ (row & 7))) * 16;
                }
            } else {
                const int tap = min(4 * (s - 12) + g, 8), kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
                for (int j = 0; j < 4; ++j) xa[j] = xs + ((2 * (a0 + j) + ph + kh) * 18 + kw + fr) * 16;
            }
        };
        unsigned xa_[2][4];
        f16x8 wf[2][2], xf[2][4];
        addr(0, xa_[0]);
        DS_READ128(wf[0][0], wbase, 0);
        DS_READ128(wf[0][1], wbase, 16 * 64);
#pragma unroll
        for (int j = 0; j < 4; ++j) DS_READ128(xf[0][j], xa_[0][j], 0);
#pragma unroll
        for (int s = 0; s < UC_NS; ++s) {                          // fully unrolled: static register double buffer
            const int cur = s & 1, nxt = cur ^ 1;
            if (s + 1 < UC_NS) addr(s + 1, xa_[nxt]);
            const unsigned wa = wbase + (unsigned)(s + 1) * (UC_N * 64);
            LGKM_WAIT(0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int i = m >> 2, j = m & 3;
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[cur][i], xf[cur][j], acc[i][j], 0, 0, 0);
                if (s + 1 < UC_NS) {
                    if (m == 0) DS_READ128(wf[nxt][0], wa, 0);
                    else if (m == 1) DS_READ128(wf[nxt][1], wa, 16 * 64);
                    else if (m < 6) DS_READ128(xf[nxt][m - 2 < 4 ? m - 2 : 0], xa_[nxt][m - 2 < 4 ? m - 2 : 0], 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        int b, y0, x0;
        origin(tile, b, y0, x0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gy = y0 + 2 * (a0 + j) + ph, gx = x0 + fr;
            if (gy >= Hf || gx >= p.W_) continue;
            const long o = ((((long)b * Hpo + gy + p.PHo) * Wpo + gx + p.PWo)) * UC_N + g * 8;
            float v[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] = acc[0][j][r]; v[4 + r] = acc[1][j][r]; }
            v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
            v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
            f16x8 hv;
#pragma unroll
            for (int r = 0; r < 8; ++r) hv[r] = (_Float16)((p.act == ADVH_ACT_LEAKY && v[r] < 0.f) ? p.slope * v[r] : v[r]);
            *(f16x8*)((_Float16*)p.out_h + o) = hv;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace advh

using namespace advh;

extern "C" int advh_upconv21_tile_lds_bytes(void) { return UC_WBYTES + 2 * UC_BUF; }

extern "C" int advh_upconv21_tile_f16(const advh_upconv_desc* d, advh_stream_t stream) {
    if (!d || !d->Xc || !d->Xs || !d->W || !d->out_h || d->B <= 0 || d->Hc <= 0 || d->W_ <= 0) return ADVH_EINVAL;
    if (d->PHc < 1 || d->PWc < 1 || d->PHs < 1 || d->PWs < 1 || d->PHo < 0 || d->PWo < 0) return ADVH_EINVAL;
    if (d->Hc % 8) return ADVH_EUNSUPPORTED;                        // tiles are 16 output rows = 8 coarse rows
    if (d->act != ADVH_ACT_NONE && d->act != ADVH_ACT_LEAKY) return ADVH_EINVAL;
    if (advh_ensure_lds((const void*)upconv21_tile_kernel) != ADVH_OK) return ADVH_ELAUNCH;
    const long ntiles = (long)d->B * ((2 * d->Hc + 15) / 16) * ((d->W_ + 15) / 16);
    const long grid = ntiles < 256 ? ntiles : 256;
    hipLaunchKernelGGL(upconv21_tile_kernel, dim3((unsigned)grid), dim3(256), UC_WBYTES + 2 * UC_BUF, (hipStream_t)stream, *d);
    return ADVH_LAUNCH_CHECK();
}
