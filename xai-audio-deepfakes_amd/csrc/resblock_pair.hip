// One HiFi-GAN ResBlock1 step fused into one kernel (speechbrain HifiganGenerator via hifigan.py:106-110, 180; Kong et
// al. 2020, ResBlock1.forward):      x <- x + conv2( lrelu( conv1( lrelu(x) ) ) ),   conv1: k taps, dilation d; conv2: k taps.
// For the 32- and 64-channel stages both weight tensors fit in LDS next to the line buffer, so the intermediate map never
// leaves the chip and the step moves one map in and one map out instead of five (the unfused line-tile launches are
// HBM-bound at 3.5-5 TB/s).  Per tile: the raw line buffer arrives by LDS DMA (double-buffered); every wavefront keeps the
// residual fragments of its outputs in registers, then the buffer is LeakyReLU'd in place; conv1 runs on 256 positions
// and leaves lrelu(conv1 + b1) -- zeroed outside the clip, which is conv2's zero padding -- in an LDS tile; conv2 runs on
// the 256 - (k-1) positions whose taps lie inside that tile and adds bias and residual.  MFMA loops as conv_taps.hip.
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

template <int C> __device__ __forceinline__ int rswz(int r) { return C == 64 ? (r & 7) : ((r >> 1) & 2); }
__device__ __forceinline__ int rcout_of(int R) { return ((R >> 5) << 5) + (((R >> 2) & 3) << 3) + (((R >> 4) & 1) << 2) + (R & 3); }

constexpr int RB_TC = 256;                 // conv1 positions per tile
constexpr int RB_TMROWS = RB_TC + 16;      // rows of the intermediate tile (conv2 taps of masked outputs may run past 256)

// acc[i][j] += sum over taps t and channels of W[t][16 i + ..][..] * rows[row0 + 16 j + fr + t * dil][..]
template <int C, int NJ>
__device__ __forceinline__ void taps_mma(unsigned wbase, unsigned xbase, int row0, int ntap, int dil, int fr, int g,
                                         f32x4 (&acc)[C / 16][NJ]) {
    constexpr int CH = C / 8, CT = C / 16, KS = C / 32;
    const int NS = ntap * KS;
    auto addr = [&](int s, unsigned& wa, unsigned& xa) {
        const int t = s / KS, c = (s % KS) * 4 + g;
        wa = wbase + (unsigned)t * (C * C * 2) + (fr * CH + (c ^ rswz<C>(fr))) * 16;
        const int row = row0 + fr + t * dil;
        xa = xbase + (row * CH + (c ^ rswz<C>(row))) * 16;
    };
    auto issue = [&](int m, f16x8 (&wf)[CT], f16x8 (&xf)[NJ], unsigned wa, unsigned xa) {
#pragma unroll
        for (int i = 0; i < CT; ++i)
            if (m == i) DS_READ128(wf[i], wa, i * 16 * C * 2);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            if (m == CT + j) DS_READ128(xf[j], xa, j * 16 * C * 2);
    };
    auto step = [&](int sn, const f16x8 (&wc)[CT], const f16x8 (&xc)[NJ], f16x8 (&wn)[CT], f16x8 (&xn)[NJ]) {
        unsigned wa, xa;
        addr(sn, wa, xa);
        LGKM_WAIT(0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < CT * NJ; ++m) {
            const int i = m / NJ, j = m % NJ;
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wc[i], xc[j], acc[i][j], 0, 0, 0);
            if (m < CT + NJ) issue(m, wn, xn, wa, xa);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    f16x8 wa_[CT], xa_[NJ], wb_[CT], xb_[NJ];
    {
        unsigned wa, xa;
        addr(0, wa, xa);
#pragma unroll
        for (int m = 0; m < CT + NJ; ++m) issue(m, wa_, xa_, wa, xa);
    }
    for (int s = 0; s < NS; s += 2) {
        step(min(s + 1, NS - 1), wa_, xa_, wb_, xb_);
        if (s + 1 < NS) step(min(s + 2, NS - 1), wb_, xb_, wa_, xa_);
    }
    LGKM_WAIT(0);
    __builtin_amdgcn_sched_barrier(0);
}

// NJ = 16-position column tiles per wavefront; 256 / (16 NJ) wavefronts per workgroup (NJ = 2: eight wavefronts for the 64-channel
// instance, which mostly runs one workgroup per CU)
template <int C, int NJ = 4>
__global__ __launch_bounds__(64 * (16 / NJ)) void resblock_pair_kernel(const advh_resblock_desc p, int nbuf) {
    constexpr int CH = C / 8, CT = C / 16, NTH = 64 * (16 / NJ), WP = 16 * NJ;      // threads; positions per wavefront
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int fr = lane & 15, g = lane >> 4;
    const int k = p.k, d = p.dil, h1 = (k - 1) * d / 2, h2 = (k - 1) / 2;
    const int TO = RB_TC - 2 * h2;                                  // outputs per tile
    const int SR = RB_TC + 2 * h1;                                  // line-buffer rows
    const int SRC = (max(SR, RB_TMROWS) * CH + 63) & ~63;           // chunks per buffer (it later holds the intermediate tile too)
    char* W1 = lds;
    char* W2 = W1 + (size_t)k * C * C * 2;
    char* XR = W2 + (size_t)k * C * C * 2;                          // nbuf x [SRC] chunks: raw lines -> (residual taken) -> activated in place
                                                                    // -> (conv1 done) -> overwritten by the intermediate tile
    // nbuf = 2: the next tile's lines arrive during this tile (one workgroup per CU); nbuf = 1: smaller footprint so that
    // TWO workgroups fit a CU and hide each other's DMA
    const _Float16* X = (const _Float16*)p.X;

    for (int which = 0; which < 2; ++which) {                       // both weight tensors: once per workgroup, rows permuted (rcout_of)
        const _Float16* Wg = (const _Float16*)(which ? p.W2 : p.W1);
        char* Wl = which ? W2 : W1;
        for (int i = tid; i < k * C * CH; i += NTH) {
            const int row = i / CH, pos = i % CH;
            const _Float16* src = Wg + ((long)(row / C) * C + rcout_of(row % C)) * C + ((pos ^ rswz<C>(row)) * 8);
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(Wl + (size_t)(i - lane) * 16), 16, 0, 0);
        }
    }
    const int ntiles = (p.M + TO - 1) / TO;
    auto load_lines = [&](int tile, int buf) {                      // rows tile*TO - h2 - h1 .. + SR, clamped into the map
        const long r0 = (long)tile * TO - h2 - h1;
        char* dst = XR + (size_t)buf * SRC * 16;
        for (int i = tid; i < SRC; i += NTH) {
            const int row = i / CH, pos = i % CH;
            long r = r0 + row;
            r = r < 0 ? 0 : (r >= p.M ? p.M - 1 : r);
            const _Float16* src = X + r * C + ((pos ^ rswz<C>(row)) * 8);
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(dst + (size_t)(i - lane) * 16), 16, 0, 0);
        }
    };
    float4 b1[CT], b2[CT];                                          // bias[2q + e] = channels 32 q + 8 g + 4 e .. + 3
#pragma unroll
    for (int i = 0; i < CT; ++i) {
        const int o = (i >> 1) * 32 + g * 8 + (i & 1) * 4;
        b1[i] = *(const float4*)(p.b1 + o);
        b2[i] = *(const float4*)(p.b2 + o);
    }
    const unsigned lds0 = (unsigned)(unsigned long)LDS_PTR(lds);
    const unsigned w1a = lds0, w2a = lds0 + (unsigned)k * (C * C * 2), xra = w2a + (unsigned)k * (C * C * 2);
    auto in_clip = [&](long m) {                                    // row m of the map is a real sample of its clip
        if (m < 0 || m >= p.M) return false;
        const int w = (int)(m % p.Wg);
        return w >= p.w0 && w < p.w1;
    };
    int buf = 0;
    if (nbuf == 2 && (int)blockIdx.x < ntiles) load_lines(blockIdx.x, 0);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, buf = nbuf == 2 ? buf ^ 1 : 0) {
        const long base = (long)tile * TO;
        if (nbuf == 1) {
            __syncthreads();                                        // everyone is done with the previous tile's intermediate rows
            load_lines(tile, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                            // the raw lines landed; everyone is done with the previous tile
        if (nbuf == 2 && tile + (int)gridDim.x < ntiles) load_lines(tile + gridDim.x, buf ^ 1);
        char* xb = XR + (size_t)buf * SRC * 16;
        char* TM = xb;
        const unsigned tma = xra + (unsigned)buf * SRC * 16;
        // residual fragments of this lane's outputs, taken before the buffer is activated in place
        f16x8 res[CT / 2][NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int row = wv * WP + j * 16 + fr + h1 + h2;        // output o = wv*64 + j*16 + fr sits at line-buffer row o + h1 + h2
#pragma unroll
            for (int q = 0; q < CT / 2; ++q)
                res[q][j] = *(const f16x8*)(xb + ((size_t)row * CH + ((4 * q + g) ^ rswz<C>(row))) * 16);
        }
        __syncthreads();
        {
            const _Float16 sl = (_Float16)p.slope;
            for (int i = tid; i < SRC; i += NTH) {
                f16x8 v = *(f16x8*)(xb + (size_t)i * 16);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] > (_Float16)0 ? v[e] : v[e] * sl;
                *(f16x8*)(xb + (size_t)i * 16) = v;
            }
        }
        __syncthreads();
        // ---- conv1 on positions c = 0 .. 255 (map row base - h2 + c); tap t reads line-buffer row c + t*d
        f32x4 acc[CT][NJ];
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        taps_mma<C, NJ>(w1a, xra + (unsigned)buf * SRC * 16, wv * WP, k, d, fr, g, acc);
        __syncthreads();                                            // every wavefront has read its lines: the buffer becomes the intermediate tile
        if (tid < (RB_TMROWS - RB_TC) * CH)                         // rows past 256 only feed masked outputs, but must be finite
            *(f16x8*)(TM + ((size_t)RB_TC * CH + tid) * 16) = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = wv * WP + j * 16 + fr;
            const bool ok = in_clip(base - h2 + c);
#pragma unroll
            for (int q = 0; q < CT / 2; ++q) {
                float v[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[r] = acc[2 * q][j][r]; v[4 + r] = acc[2 * q + 1][j][r]; }
                v[0] += b1[2 * q].x; v[1] += b1[2 * q].y; v[2] += b1[2 * q].z; v[3] += b1[2 * q].w;
                v[4] += b1[2 * q + 1].x; v[5] += b1[2 * q + 1].y; v[6] += b1[2 * q + 1].z; v[7] += b1[2 * q + 1].w;
                f16x8 hv;
#pragma unroll
                for (int r = 0; r < 8; ++r) hv[r] = ok ? (_Float16)(v[r] > 0.f ? v[r] : p.slope * v[r]) : (_Float16)0.f;
                *(f16x8*)(TM + ((size_t)c * CH + ((4 * q + g) ^ rswz<C>(c))) * 16) = hv;
            }
        }
        __syncthreads();
        // ---- conv2 on outputs o = 0 .. TO-1 (map row base + o); tap t reads intermediate row o + t
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        taps_mma<C, NJ>(w2a, tma, wv * WP, k, 1, fr, g, acc);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int o = wv * WP + j * 16 + fr;
            const long m = base + o;
            if (o >= TO || m >= p.M) continue;
            const bool ok = in_clip(m);
#pragma unroll
            for (int q = 0; q < CT / 2; ++q) {
                float v[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[r] = acc[2 * q][j][r]; v[4 + r] = acc[2 * q + 1][j][r]; }
                v[0] += b2[2 * q].x; v[1] += b2[2 * q].y; v[2] += b2[2 * q].z; v[3] += b2[2 * q].w;
                v[4] += b2[2 * q + 1].x; v[5] += b2[2 * q + 1].y; v[6] += b2[2 * q + 1].z; v[7] += b2[2 * q + 1].w;
                f16x8 hv;
#pragma unroll
                for (int r = 0; r < 8; ++r) hv[r] = ok ? (_Float16)(v[r] + (float)res[q][j][r]) : (_Float16)0.f;
                *(f16x8*)((_Float16*)p.out_h + m * C + q * 32 + g * 8) = hv;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

static int rb_lds(int C, int k, int dil, int nbuf) {
    const int h1 = (k - 1) * dil / 2, sr = RB_TC + 2 * h1, rows = sr > RB_TMROWS ? sr : RB_TMROWS;
    const int src = (rows * (C / 8) + 63) / 64 * 64;
    return 2 * k * C * C * 2 + nbuf * src * 16;
}

// two workgroups per CU with one line buffer each where that fits (<= 80 KiB), else one workgroup with two buffers
// (a single buffer with one workgroup per CU -- DMA exposed -- is the last resort: 64 channels, k = 7)
static int rb_nbuf(int C, int k, int dil) {
    if (rb_lds(C, k, dil, 1) <= 80 * 1024) return 1;
    return rb_lds(C, k, dil, 2) <= 160 * 1024 ? 2 : 1;
}

}  // namespace advh

using namespace advh;

extern "C" int advh_resblock_pair_lds_bytes(int C, int k, int dil) { return rb_lds(C, k, dil, rb_nbuf(C, k, dil)); }

extern "C" int advh_resblock_pair_f16(const advh_resblock_desc* d, int C, advh_stream_t stream) {
    if (!d || !d->X || !d->W1 || !d->W2 || !d->b1 || !d->b2 || !d->out_h || d->M <= 0 || d->Wg <= 0 || d->k < 1 || !(d->k & 1) ||
        d->k > 15 || d->dil < 1 || d->X == d->out_h)
        return ADVH_EINVAL;
    if (C != 32 && C != 64) return ADVH_EUNSUPPORTED;
    const int nbuf = rb_nbuf(C, d->k, d->dil);
    const int lds = rb_lds(C, d->k, d->dil, nbuf);
    if (lds > 160 * 1024 || (d->k - 1) / 2 * 2 >= RB_TC / 2) return ADVH_EUNSUPPORTED;
    const int ci = C == 64;
    const void* fn = ci ? (const void*)resblock_pair_kernel<64, 2> : (const void*)resblock_pair_kernel<32>;
    if (advh_ensure_lds(fn) != ADVH_OK) return ADVH_ELAUNCH;
    const int TO = RB_TC - (d->k - 1);
    const int ntiles = (d->M + TO - 1) / TO;
    int grid = 256 * (lds <= 80 * 1024 ? 2 : 1);
    if (grid > ntiles) grid = ntiles;
    if (ci) hipLaunchKernelGGL((resblock_pair_kernel<64, 2>), dim3(grid), dim3(512), lds, (hipStream_t)stream, *d, nbuf);
    else hipLaunchKernelGGL(resblock_pair_kernel<32>, dim3(grid), dim3(256), lds, (hipStream_t)stream, *d, nbuf);
    return ADVH_LAUNCH_CHECK();
}
