// fp32-class form of the fused HiFi-GAN ResBlock1 step (resblock_pair.hip) for the 32-channel stage:
//     x <- x + conv2( lrelu( conv1( lrelu(x) ) ) ),   conv1: k taps, dilation d; conv2: k taps, dilation 1
// (speechbrain HifiganGenerator via hifigan.py:106-110, 180; Kong et al. 2020, ResBlock1.forward; the reference runs it in fp32).
//
// Every tensor is a split-format plane pair (hi + lo * 2^-11, csrc/device_math.h) and every product costs three fp16 MFMAs
// with fp32 accumulation -- acc += Wh Xh; accx += Wh Xl + Wl Xh; result = acc + accx * 2^-11 -- exactly the arithmetic and the
// K order (tap-major, 32 channels per step) of the x3 implicit GEMM the unfused path runs, so the two paths agree to the last
// bits of the split representation.  As in the fp16 kernel the step moves ONE map in and ONE map out instead of five: both
// weight tensors (two planes each: k * 8 KiB) stay resident in LDS, the raw line buffer (both planes) arrives by LDS DMA, the
// residual is kept in registers as fp32, LeakyReLU is applied to the joined value in place, conv1 leaves lrelu(conv1 + b1)
// (zero outside the clip = conv2's zero padding) as a split tile over the same buffer, conv2 adds bias and residual.
// In the fp32-class mode the unfused 32-channel layers are HBM-bound on 4-byte elements (five maps per step); the fused step is
// bound by its 3x MFMA work instead.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include "addvisor_hip.h"
#include "common.h"
#include "device_math.h"

namespace advh {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define DS_READ128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define LGKM_WAIT(n) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory")

namespace x3rb {
constexpr int C = 32, CH = 4, CT = 2, NJ = 4, NTH = 256, WP = 64;          // channels, 16-byte chunks per row, 16-row weight tiles, column tiles per wavefront
constexpr int TC = 256, TMROWS = TC + 16;                                  // conv1 positions per tile; rows of the intermediate tile
__device__ __forceinline__ int swz(int r) { return (r >> 1) & 2; }         // 64-byte rows: conflict-free ds_read_b128 from any starting row
__device__ __forceinline__ int cout_of(int R) { return ((R >> 5) << 5) + (((R >> 2) & 3) << 3) + (((R >> 4) & 1) << 2) + (R & 3); }

// acc / accx += sum over taps t and channels of W[t][..][..] * rows[row0 + 16 j + fr + t * dil][..] in split arithmetic.
// wbase: hi plane of the weight tensor (lo plane wlo bytes behind); xbase: hi plane of the rows (lo plane xlo bytes behind).
__device__ __forceinline__ void taps_mma_x3(unsigned wbase, unsigned wlo, unsigned xbase, unsigned xlo, int row0, int ntap, int dil, int fr, int g,
                                            f32x4 (&acc)[CT][NJ], f32x4 (&accx)[CT][NJ]) {
    auto addr = [&](int t, unsigned& wa, unsigned& xa) {
        wa = wbase + (unsigned)t * (C * C * 2) + (fr * CH + (g ^ swz(fr))) * 16;
        const int row = row0 + fr + t * dil;
        xa = xbase + (row * CH + (g ^ swz(row))) * 16;
    };
    auto issue = [&](f16x8 (&wh)[CT], f16x8 (&wl)[CT], f16x8 (&xh)[NJ], f16x8 (&xl)[NJ], unsigned wa, unsigned xa) {
#pragma unroll
        for (int i = 0; i < CT; ++i) { DS_READ128(wh[i], wa, i * 16 * C * 2); DS_READ128(wl[i], wa + wlo, i * 16 * C * 2); }
#pragma unroll
        for (int j = 0; j < NJ; ++j) { DS_READ128(xh[j], xa, j * 16 * C * 2); DS_READ128(xl[j], xa + xlo, j * 16 * C * 2); }
    };
    // one 32-deep k-step per tap (C = 32); the fragments of tap t+1 are requested before the 24 MFMAs of tap t issue
    f16x8 wh[2][CT], wl[2][CT], xh[2][NJ], xl[2][NJ];
    {
        unsigned wa, xa;
        addr(0, wa, xa);
        issue(wh[0], wl[0], xh[0], xl[0], wa, xa);
    }
    for (int t = 0; t < ntap; t += 2) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (t + half >= ntap) break;
            const int cur = half, nxt = half ^ 1;
            unsigned wa, xa;
            addr(min(t + half + 1, ntap - 1), wa, xa);
            LGKM_WAIT(0);
            __builtin_amdgcn_sched_barrier(0);
            issue(wh[nxt], wl[nxt], xh[nxt], xl[nxt], wa, xa);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
#pragma unroll
                for (int i = 0; i < CT; ++i) accx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[cur][i], xl[cur][j], accx[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < CT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[cur][i], xh[cur][j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < CT; ++i) accx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[cur][i], xh[cur][j], accx[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    LGKM_WAIT(0);                                                          // the last (redundant) prefetch must land before its registers are reused
    __builtin_amdgcn_sched_barrier(0);
}
}  // namespace x3rb

__global__ __launch_bounds__(256, 2) void resblock_pair_x3_kernel(const advh_resblock_x3_desc p, int nbuf) {
    using namespace x3rb;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int fr = lane & 15, g = lane >> 4;
    const int k = p.k, d = p.dil, h1 = (k - 1) * d / 2, h2 = (k - 1) / 2;
    const int TO = TC - 2 * h2;                                            // outputs per tile
    const int SR = TC + 2 * h1;                                            // line-buffer rows
    const int SRC = (max(SR, TMROWS) * CH + 63) & ~63;                     // chunks per plane of a buffer (it later holds the intermediate tile too)
    const unsigned WB = (unsigned)k * C * C * 2;                           // bytes of one plane of one weight tensor
    // LDS: W1 hi | W1 lo | W2 hi | W2 lo | nbuf x (lines hi | lines lo)
    char* XR = lds + 4 * (size_t)WB;
    const unsigned PB = (unsigned)SRC * 16;                                // bytes of one plane of a line buffer
    const _Float16* X = (const _Float16*)p.X;

    for (int which = 0; which < 4; ++which) {                              // (W1, W2) x (hi, lo): once per workgroup, rows permuted (cout_of)
        const _Float16* Wg = (const _Float16*)((which >> 1) ? p.W2 : p.W1) + (size_t)(which & 1) * p.w_lo;
        char* Wl = lds + (size_t)which * WB;
        for (int i = tid; i < k * C * CH; i += NTH) {
            const int row = i / CH, pos = i % CH;
            const _Float16* src = Wg + ((long)(row / C) * C + cout_of(row % C)) * C + ((pos ^ swz(row)) * 8);
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(Wl + (size_t)(i - lane) * 16), 16, 0, 0);
        }
    }
    const int ntiles = (p.M + TO - 1) / TO;
    auto load_lines = [&](int tile, int buf) {                             // rows tile*TO - h2 - h1 .. + SR of both planes, clamped into the map
        const long r0 = (long)tile * TO - h2 - h1;
        char* dst = XR + (size_t)buf * 2 * PB;
        for (int i = tid; i < SRC; i += NTH) {
            const int row = i / CH, pos = i % CH;
            long r = r0 + row;
            r = r < 0 ? 0 : (r >= p.M ? p.M - 1 : r);
            const _Float16* src = X + r * C + ((pos ^ swz(row)) * 8);
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(dst + (size_t)(i - lane) * 16), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src + p.x_lo), LDS_PTR(dst + PB + (size_t)(i - lane) * 16), 16, 0, 0);
        }
    };
    float b1[8], b2[8];                                                    // this lane's 8 consecutive output channels 8 g .. 8 g + 7
#pragma unroll
    for (int e = 0; e < 8; ++e) { b1[e] = p.b1[g * 8 + e]; b2[e] = p.b2[g * 8 + e]; }
    const unsigned lds0 = (unsigned)(unsigned long)LDS_PTR(lds);
    const unsigned w1a = lds0, w2a = lds0 + 2 * WB, xra = lds0 + 4 * WB;
    auto in_clip = [&](long m) {                                           // row m of the map is a real sample of its clip
        if (m < 0 || m >= p.M) return false;
        const int w = (int)(m % p.Wg);
        return w >= p.w0 && w < p.w1;
    };
    int buf = 0;
    if (nbuf == 2 && (int)blockIdx.x < ntiles) load_lines(blockIdx.x, 0);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, buf = nbuf == 2 ? buf ^ 1 : 0) {
        const long base = (long)tile * TO;
        if (nbuf == 1) {
            __syncthreads();                                               // everyone is done with the previous tile's intermediate rows
            load_lines(tile, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                                   // the raw lines landed; everyone is done with the previous tile
        if (nbuf == 2 && tile + (int)gridDim.x < ntiles) load_lines(tile + gridDim.x, buf ^ 1);
        char* xh = XR + (size_t)buf * 2 * PB;
        char* xl = xh + PB;
        const unsigned xa0 = xra + (unsigned)buf * 2 * PB;
        // residual of this lane's outputs as fp32, taken before the buffer is activated in place
        float res[NJ][8];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int row = wv * WP + j * 16 + fr + h1 + h2;               // output o sits at line-buffer row o + h1 + h2
            const size_t o = ((size_t)row * CH + (g ^ swz(row))) * 16;
            const f16x8 hv = *(const f16x8*)(xh + o), lv = *(const f16x8*)(xl + o);
#pragma unroll
            for (int e = 0; e < 8; ++e) res[j][e] = join_f32(hv[e], lv[e]);
        }
        __syncthreads();
        for (int i = tid; i < SRC; i += NTH) {                             // LeakyReLU on the joined value, re-split in place
            f16x8 hv = *(f16x8*)(xh + (size_t)i * 16), lv = *(f16x8*)(xl + (size_t)i * 16);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = join_f32(hv[e], lv[e]);
                v = v > 0.f ? v : p.slope * v;
                _Float16 h, l;
                split_f32(v, h, l);
                hv[e] = h; lv[e] = l;
            }
            *(f16x8*)(xh + (size_t)i * 16) = hv;
            *(f16x8*)(xl + (size_t)i * 16) = lv;
        }
        __syncthreads();
        // ---- conv1 on positions c = 0 .. 255 (map row base - h2 + c); tap t reads line-buffer row c + t*d
        f32x4 acc[CT][NJ], accx[CT][NJ];
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) { acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; accx[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        taps_mma_x3(w1a, WB, xa0, PB, wv * WP, k, d, fr, g, acc, accx);
        __syncthreads();                                                   // every wavefront has read its lines: the buffer becomes the intermediate tile
        if (tid < (TMROWS - TC) * CH) {                                    // rows past 256 only feed masked outputs, but must be finite
            *(f16x8*)(xh + ((size_t)TC * CH + tid) * 16) = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            *(f16x8*)(xl + ((size_t)TC * CH + tid) * 16) = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = wv * WP + j * 16 + fr;
            const bool ok = in_clip(base - h2 + c);
            f16x8 hv, lv;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = fmaf(accx[e >> 2][j][e & 3], SPLIT_LO_INV, acc[e >> 2][j][e & 3]) + b1[e];
                v = ok ? (v > 0.f ? v : p.slope * v) : 0.f;
                _Float16 h, l;
                split_f32(v, h, l);
                hv[e] = h; lv[e] = l;
            }
            const size_t o = ((size_t)c * CH + (g ^ swz(c))) * 16;
            *(f16x8*)(xh + o) = hv;
            *(f16x8*)(xl + o) = lv;
        }
        __syncthreads();
        // ---- conv2 on outputs o = 0 .. TO-1 (map row base + o); tap t reads intermediate row o + t
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) { acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; accx[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        taps_mma_x3(w2a, WB, xa0, PB, wv * WP, k, 1, fr, g, acc, accx);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int o = wv * WP + j * 16 + fr;
            const long m = base + o;
            if (o >= TO || m >= p.M) continue;
            const bool ok = in_clip(m);
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = ok ? fmaf(accx[e >> 2][j][e & 3], SPLIT_LO_INV, acc[e >> 2][j][e & 3]) + b2[e] + res[j][e] : 0.f;
            store_h_rt<8>((_Float16*)p.out_h, m * C + g * 8, p.o_lo, v);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

static int rbx3_lds(int k, int dil, int nbuf) {
    const int h1 = (k - 1) * dil / 2, sr = x3rb::TC + 2 * h1, rows = sr > x3rb::TMROWS ? sr : x3rb::TMROWS;
    const int src = (rows * x3rb::CH + 63) / 64 * 64;
    return 4 * k * 32 * 32 * 2 + nbuf * 2 * src * 16;
}
// two workgroups per CU with one line buffer each where that fits (<= 80 KiB), else one workgroup with two buffers, else one with one
static int rbx3_nbuf(int k, int dil) {
    if (rbx3_lds(k, dil, 1) <= 80 * 1024) return 1;
    return rbx3_lds(k, dil, 2) <= 160 * 1024 ? 2 : 1;
}

}  // namespace advh

using namespace advh;

extern "C" int advh_resblock_pair_x3_lds_bytes(int C, int k, int dil) {
    if (C != 32 || k < 1 || !(k & 1) || k > 15 || dil < 1) return -1;
    const int lds = rbx3_lds(k, dil, rbx3_nbuf(k, dil));
    return lds <= 160 * 1024 ? lds : -1;
}

extern "C" int advh_resblock_pair_x3(const advh_resblock_x3_desc* d, int C, advh_stream_t stream) {
    if (!d || !d->X || !d->W1 || !d->W2 || !d->b1 || !d->b2 || !d->out_h || d->M <= 0 || d->Wg <= 0 || d->k < 1 || !(d->k & 1) ||
        d->k > 15 || d->dil < 1 || d->X == d->out_h || d->x_lo <= 0 || d->o_lo <= 0 || d->w_lo <= 0 || (d->x_lo & 7) || (d->o_lo & 7) || (d->w_lo & 7))
        return ADVH_EINVAL;
    if (C != 32) return ADVH_EUNSUPPORTED;
    const int nbuf = rbx3_nbuf(d->k, d->dil);
    const int lds = rbx3_lds(d->k, d->dil, nbuf);
    if (lds > 160 * 1024 || (d->k - 1) / 2 * 2 >= x3rb::TC / 2) return ADVH_EUNSUPPORTED;
    if (advh_ensure_lds((const void*)resblock_pair_x3_kernel) != ADVH_OK) return ADVH_ELAUNCH;
    const int TO = x3rb::TC - (d->k - 1);
    const int ntiles = (d->M + TO - 1) / TO;
    int grid = 256 * (lds <= 80 * 1024 ? 2 : 1);
    if (grid > ntiles) grid = ntiles;
    hipLaunchKernelGGL(resblock_pair_x3_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, *d, nbuf);
    return ADVH_LAUNCH_CHECK();
}

ADVH_SPLIT_FLAG_SETTER(advh_split_flag_resblock_pair_x3)
