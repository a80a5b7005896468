// Weight gradient of a 3x3 stride-1 "same" Conv2d with C_in = C_out = C in {32, 64} without materialised transposes
// (training step of the U-Net, addvisor.py:20-24 under train_addvisor.py:376):
//     dW[kh][kw][co][ci] = sum over positions p of dz[p][co] * x[p + (kh-1, kw-1)][ci].
// The reduction runs over POSITIONS, i.e. over the slow index of both channels-last operands.  Instead of transposing
// them in HBM (advh_transpose_gather + split-K GEMM, 4 C x the map size of extra traffic), a persistent workgroup streams
// 16 x 16 position tiles through LDS -- the dz tile and the 18 x 18 input patch, exactly the line buffer of
// conv_taps2d_kernel -- and reads BOTH MFMA operands with the transposing LDS load ds_read_b64_tr_b16: lane i of a
// 16-lane group receives channel c0+i of 4 consecutive positions, which is the 16x16x32 operand layout with k = position.
// Accumulators stay in registers across all tiles of the workgroup (18 units = 9 taps x 2 input-channel groups dealt to
// the 4 wavefronts); each workgroup writes one fp32 partial, a second kernel adds the partials in a fixed order.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include "addvisor_hip.h"
#include "common.h"

namespace advh {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __fp16 trvec __attribute__((__vector_size__(4 * sizeof(__fp16))));

#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int C> __device__ __forceinline__ int wswz(int r) { return C == 64 ? (r & 7) : ((r >> 1) & 2); }

// 8 k-values (positions) of one channel per lane: two transposing reads of 4 rows each
template <int C>
__device__ __forceinline__ f16x8 tr_frag(const __attribute__((address_space(3))) char* base, int row0, int row1, int c0, int q, int p) {
    // lane 4q+p of its 16-lane group supplies row (rowX + q), columns c0 + 4p .. + 3 (8 bytes)
    const int ra = row0 + q, rb = row1 + q;
    const int ch = (c0 >> 3) + (p >> 1), sub = (p & 1) * 8;
    const auto* pa = (const __attribute__((address_space(3))) trvec*)(base + ((size_t)ra * (C / 8) + (ch ^ wswz<C>(ra))) * 16 + sub);
    const auto* pb = (const __attribute__((address_space(3))) trvec*)(base + ((size_t)rb * (C / 8) + (ch ^ wswz<C>(rb))) * 16 + sub);
    trvec lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) trvec*)pa);
    trvec hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) trvec*)pb);
    f16x8 r;
    __builtin_memcpy(&r, &lo, 8);
    __builtin_memcpy((char*)&r + 8, &hi, 8);
    return r;
}

template <int C>
__global__ __launch_bounds__(256) void conv_wgrad2d_kernel(const advh_wgrad2d_desc p) {
    constexpr int CH = C / 8, CT = C / 16, CIG = CT / 2, PR = 18, SRX = PR * PR, SRZ = 256, MAXU = 5;
    constexpr int NX = (SRX * CH + 63) & ~63, NZ = SRZ * CH;      // 16-byte chunks per buffer
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3, fr = lane & 15;
    const _Float16* X = (const _Float16*)p.X;
    const _Float16* Z = (const _Float16*)p.DZ;
    const int Hx = p.H + 2 * p.PHx, Wx = p.W_ + 2 * p.PWx, Hz = p.H + 2 * p.PHz, Wz = p.W_ + 2 * p.PWz;
    const int tx = (p.W_ + 15) / 16, ty = (p.H + 15) / 16, ntiles = p.B * ty * tx;
    auto origin = [&](int tile, int& b, int& y0, int& x0) {
        x0 = (tile % tx) * 16;
        const int r = tile / tx;
        y0 = (r % ty) * 16;
        b = r / ty;
    };
    auto load_tile = [&](int tile, int buf) {
        int b, y0, x0;
        origin(tile, b, y0, x0);
        char* xd = lds + (size_t)buf * (NX + NZ) * 16;
        char* zd = xd + (size_t)NX * 16;
        for (int i = tid; i < NX; i += 256) {                      // 18 x 18 input patch; rows outside the map are clamped (finite)
            int row = i / CH, pos = i % CH;
            if (row >= SRX) row = 0;
            int gy = min(y0 + p.PHx - 1 + row / PR, Hx - 1), gx = min(x0 + p.PWx - 1 + row % PR, Wx - 1);
            const _Float16* src = X + (((long)b * Hx + gy) * Wx + gx) * C + ((pos ^ wswz<C>(i / CH)) * 8);
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(xd + (size_t)(i - lane) * 16), 16, 0, 0);
        }
        for (int i = tid; i < NZ; i += 256) {                      // 16 x 16 dz tile; positions outside the image read a halo zero
            const int row = i / CH, pos = i % CH;
            const int ly = row >> 4, lx = row & 15;
            const bool in = y0 + ly < p.H && x0 + lx < p.W_;
            const int gy = in ? y0 + ly + p.PHz : 0, gx = in ? x0 + lx + p.PWz : 0;
            const _Float16* src = Z + (((long)b * Hz + gy) * Wz + gx) * C + ((pos ^ wswz<C>(row)) * 8);
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(zd + (size_t)(i - lane) * 16), 16, 0, 0);
        }
    };
    f32x4 acc[MAXU][CT][CIG];
#pragma unroll
    for (int u = 0; u < MAXU; ++u)
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
            for (int j = 0; j < CIG; ++j) acc[u][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const auto* lds3 = (const __attribute__((address_space(3))) char*)LDS_PTR(lds);
    int buf = 0;
    if ((int)blockIdx.x < ntiles) load_tile(blockIdx.x, 0);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, buf ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x, buf ^ 1);
        const auto* xb = lds3 + (size_t)buf * (NX + NZ) * 16;
        const auto* zb = xb + (size_t)NX * 16;
        for (int ks = 0; ks < 8; ++ks) {                           // 32 positions per step: tile rows 2ks (elements 0-3) and 2ks+1 (4-7)
            f16x8 af[CT];
#pragma unroll
            for (int i = 0; i < CT; ++i) af[i] = tr_frag<C>(zb, 32 * ks + 4 * g, 32 * ks + 16 + 4 * g, i * 16, q, pp);
#pragma unroll
            for (int u = 0; u < MAXU; ++u) {
                const int unit = wv + 4 * u;                       // wave-uniform
                if (unit >= 18) break;
                const int t = unit >> 1, cig = unit & 1, kh = t / 3, kw = t - kh * 3;
                const int r0 = (2 * ks + kh) * PR + kw + 4 * g, r1 = r0 + PR;
#pragma unroll
                for (int j = 0; j < CIG; ++j) {
                    const f16x8 bf = tr_frag<C>(xb, r0, r1, (cig * CIG + j) * 16, q, pp);
#pragma unroll
                    for (int i = 0; i < CT; ++i) acc[u][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf, acc[u][i][j], 0, 0, 0);
                }
            }
        }
    }
    // partial[blk][t][co][ci]: D row = co (4g + r), column = ci (fr)
    float* out = p.partial + (size_t)blockIdx.x * 9 * C * C;
#pragma unroll
    for (int u = 0; u < MAXU; ++u) {
        const int unit = wv + 4 * u;
        if (unit >= 18) break;
        const int t = unit >> 1, cig = unit & 1;
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
            for (int j = 0; j < CIG; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    out[((size_t)t * C + i * 16 + 4 * g + r) * C + (cig * CIG + j) * 16 + fr] = acc[u][i][j][r];
    }
}

// fp32-class form (round 3): both maps are split-format plane pairs (hi plane, lo plane x_lo / z_lo elements behind), every fragment pair
// costs three MFMAs (acc += Zh Xh; accx += Zh Xl + Zl Xh; result acc + accx * 2^-11, the arithmetic of gemm_x3_kernel), and the two planes of a
// tile are staged by the same LDS DMA.  The split-K GEMM it replaces for these layers needed four operand transposes per layer (both planes of
// x with its three horizontal taps and of dz: 2.2 ms for the 512 x 196 x 32-channel map of a 64-clip batch) and then ran at 58 TFLOP/s because
// its A operand is re-read once per vertical tap (profiles/r03_train_f32_kernel_summary.txt).  Tile shape, buffering and wavefront count per
// (CI, CO): advh_conv_wgrad2d_split below.
// CI input channels (a slice [cx0, cx0 + CI) of a map with Cx channels) x CO output channels (slice [cz0, cz0 + CO) of the Cz-channel dz
// map): wider layers and concatenated sources are covered slice pair by slice pair (addvisor_hip/unet_train.py), each launch streaming its two
// slices once -- (CI + CO) x 4 bytes per position for 9 x CI x CO x 3 MFMA-MACs, against (128 + 128) x 4 bytes per 128 x 128 MACs of a split-K
// GEMM tile whose operands had to be transposed first.
// NW wavefronts: 4 (a tap's input-channel tiles in two units, 18 units) or -- CI = 64 only -- 8 (one unit per tap and 16-channel input tile, 36 units:
// half the accumulators per wavefront, so two wavefronts share a SIMD and cover each other's LDS latency).
template <int CI, int CO, int TR, int NBUF, int NW>
__global__ __launch_bounds__(64 * NW) void conv_wgrad2d_x3_kernel(const advh_wgrad2d_desc p, int Cx, int cx0, int Cz, int cz0, long x_lo, long z_lo) {
    constexpr int CHX = CI / 8, CHZ = CO / 8, CT = CO / 16, NTI = CI / 16, UPT = NW == 4 ? 2 : NTI, CIG = NTI / UPT, NUNIT = 9 * UPT;
    constexpr int PR = 18, SRX = (TR + 2) * PR, SRZ = TR * 16, MAXU = (NUNIT + NW - 1) / NW, NTH = 64 * NW;
    constexpr int NX = (SRX * CHX + 63) & ~63, NZ = SRZ * CHZ;    // 16-byte chunks per plane and buffer
    constexpr int PLN = (NX + NZ) * 16, BUF = 2 * PLN;            // bytes of one plane / of one buffer (hi plane, lo plane)
    static_assert(NZ % 64 == 0, "whole wavefronts of DMA pieces");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3, fr = lane & 15;
    const _Float16* X = (const _Float16*)p.X + cx0;
    const _Float16* Z = (const _Float16*)p.DZ + cz0;
    const int Hx = p.H + 2 * p.PHx, Wx = p.W_ + 2 * p.PWx, Hz = p.H + 2 * p.PHz, Wz = p.W_ + 2 * p.PWz;
    const int tx = (p.W_ + 15) / 16, ty = (p.H + TR - 1) / TR, ntiles = p.B * ty * tx;
    auto load_tile = [&](int tile, int buf) {
        const int x0 = (tile % tx) * 16, r_ = tile / tx, y0 = (r_ % ty) * TR, b = r_ / ty;
        char* xd = lds + (size_t)buf * BUF;
        char* zd = xd + (size_t)NX * 16;
        for (int i = tid; i < NX; i += NTH) {                      // (TR + 2) x 18 input patch; rows outside the map are clamped (finite)
            int row = i / CHX, pos = i % CHX;
            if (row >= SRX) row = 0;
            int gy = min(y0 + p.PHx - 1 + row / PR, Hx - 1), gx = min(x0 + p.PWx - 1 + row % PR, Wx - 1);
            const _Float16* src = X + (((long)b * Hx + gy) * Wx + gx) * Cx + ((pos ^ wswz<CI>(i / CHX)) * 8);
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(xd + (size_t)(i - lane) * 16), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src + x_lo), LDS_PTR(xd + PLN + (size_t)(i - lane) * 16), 16, 0, 0);
        }
        for (int i = tid; i < NZ; i += NTH) {                      // TR x 16 dz tile; positions outside the image read a halo zero
            const int row = i / CHZ, pos = i % CHZ;
            const int ly = row >> 4, lx = row & 15;
            const bool in = y0 + ly < p.H && x0 + lx < p.W_;
            const int gy = in ? y0 + ly + p.PHz : 0, gx = in ? x0 + lx + p.PWz : 0;
            const _Float16* src = Z + (((long)b * Hz + gy) * Wz + gx) * Cz + ((pos ^ wswz<CO>(row)) * 8);
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(zd + (size_t)(i - lane) * 16), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src + z_lo), LDS_PTR(zd + PLN + (size_t)(i - lane) * 16), 16, 0, 0);
        }
    };
    f32x4 acc[MAXU][CT][CIG], accx[MAXU][CT][CIG];
#pragma unroll
    for (int u = 0; u < MAXU; ++u)
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
            for (int j = 0; j < CIG; ++j) { acc[u][i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; accx[u][i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const auto* lds3 = (const __attribute__((address_space(3))) char*)LDS_PTR(lds);
    int buf = 0;
    if (NBUF == 2 && (int)blockIdx.x < ntiles) load_tile(blockIdx.x, 0);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (NBUF == 1) {
            __syncthreads();                                       // every wavefront is done with the previous tile
            load_tile(tile, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (NBUF == 2 && tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x, buf ^ 1);
        const auto* xh = lds3 + (size_t)buf * BUF;
        const auto* zh = xh + (size_t)NX * 16;
#pragma unroll 1
        for (int ks = 0; ks < TR / 2; ++ks) {                      // 32 positions per step: tile rows 2ks (elements 0-3) and 2ks+1 (4-7)
            f16x8 ah[CT], al[CT];
#pragma unroll
            for (int i = 0; i < CT; ++i) {
                ah[i] = tr_frag<CO>(zh, 32 * ks + 4 * g, 32 * ks + 16 + 4 * g, i * 16, q, pp);
                al[i] = tr_frag<CO>(zh + PLN, 32 * ks + 4 * g, 32 * ks + 16 + 4 * g, i * 16, q, pp);
            }
#pragma unroll
            for (int u = 0; u < MAXU; ++u) {
                const int unit = wv + NW * u;                      // wave-uniform
                if (unit >= NUNIT) break;
                const int t = unit / UPT, cig = unit % UPT, kh = t / 3, kw = t - kh * 3;
                const int r0 = (2 * ks + kh) * PR + kw + 4 * g, r1 = r0 + PR;
#pragma unroll
                for (int j = 0; j < CIG; ++j) {
                    const f16x8 bh = tr_frag<CI>(xh, r0, r1, (cig * CIG + j) * 16, q, pp);
                    const f16x8 bl = tr_frag<CI>(xh + PLN, r0, r1, (cig * CIG + j) * 16, q, pp);
#pragma unroll
                    for (int i = 0; i < CT; ++i) {
                        accx[u][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl, accx[u][i][j], 0, 0, 0);
                        acc[u][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh, acc[u][i][j], 0, 0, 0);
                        accx[u][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh, accx[u][i][j], 0, 0, 0);
                    }
                }
            }
        }
        if (NBUF == 2) buf ^= 1;
    }
    // partial[blk][t][co][ci]: D row = co (4g + r), column = ci (fr)
    float* out = p.partial + (size_t)blockIdx.x * 9 * CO * CI;
#pragma unroll
    for (int u = 0; u < MAXU; ++u) {
        const int unit = wv + NW * u;
        if (unit >= NUNIT) break;
        const int t = unit / UPT, cig = unit % UPT;
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
            for (int j = 0; j < CIG; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    out[((size_t)t * CO + i * 16 + 4 * g + r) * CI + (cig * CIG + j) * 16 + fr] = fmaf(accx[u][i][j][r], 1.f / 2048.f, acc[u][i][j][r]);
    }
}

// partial [nparts][n] -> out[n]: one thread per output, the parts added in index order in fp64 (deterministic); consecutive threads read
// consecutive outputs of one part, so every load instruction covers whole cache lines (the first form gave a wavefront to each output and read
// the parts with a stride of n floats: 46 us for 256 x 36 864 floats, 3.5 ms of the training step; profiles/r03_train_f32_kernel_summary.txt)
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial, int nparts, int n, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double a = 0.0;
    int pi = 0;
    for (; pi + 8 <= nparts; pi += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = partial[(size_t)(pi + j) * n + i];
#pragma unroll
        for (int j = 0; j < 8; ++j) a += (double)v[j];
    }
    for (; pi < nparts; ++pi) a += (double)partial[(size_t)pi * n + i];
    out[i] = (float)a;
}

}  // namespace advh

using namespace advh;

extern "C" int advh_conv_wgrad2d_parts(int C, int B, int H, int W) {
    const long ntiles = (long)B * ((H + 15) / 16) * ((W + 15) / 16);
    const long grid = 256L * (C == 32 ? 2 : 1);
    return (int)(ntiles < grid ? ntiles : grid);
}

extern "C" int advh_conv_wgrad2d_f16(const advh_wgrad2d_desc* d, int C, float* dw, advh_stream_t stream) {
    if (!d || !d->X || !d->DZ || !d->partial || !dw || d->B <= 0 || d->H <= 0 || d->W_ <= 0 || d->PHx < 1 || d->PWx < 1 ||
        d->PHz < 1 || d->PWz < 1)
        return ADVH_EINVAL;
    if (C != 32 && C != 64) return ADVH_EUNSUPPORTED;
    const int lds = 2 * ((((18 * 18 * (C / 8) + 63) & ~63) + 256 * (C / 8)) * 16);
    const int ci = C == 64;
    const void* fn = ci ? (const void*)conv_wgrad2d_kernel<64> : (const void*)conv_wgrad2d_kernel<32>;
    if (advh_ensure_lds(fn) != ADVH_OK) return ADVH_ELAUNCH;
    const int grid = advh_conv_wgrad2d_parts(C, d->B, d->H, d->W_);
    hipStream_t s = (hipStream_t)stream;
    if (ci) hipLaunchKernelGGL(conv_wgrad2d_kernel<64>, dim3(grid), dim3(256), lds, s, *d);
    else hipLaunchKernelGGL(conv_wgrad2d_kernel<32>, dim3(grid), dim3(256), lds, s, *d);
    const int n = 9 * C * C;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d->partial, grid, n, dw);
    return ADVH_LAUNCH_CHECK();
}

template <int CI, int CO, int TR, int NBUF, int NW>
static int launch_wgrad2d_x3(const advh_wgrad2d_desc& d, int Cx, int cx0, int Cz, int cz0, long x_lo, long z_lo, int grid, hipStream_t s) {
    constexpr int lds = NBUF * 2 * (((((TR + 2) * 18 * (CI / 8) + 63) & ~63) + TR * 16 * (CO / 8)) * 16);
    static_assert(lds <= 160 * 1024, "tile buffers");
    if (advh_ensure_lds((const void*)conv_wgrad2d_x3_kernel<CI, CO, TR, NBUF, NW>) != ADVH_OK) return ADVH_ELAUNCH;
    hipLaunchKernelGGL((conv_wgrad2d_x3_kernel<CI, CO, TR, NBUF, NW>), dim3(grid), dim3(64 * NW), lds, s, d, Cx, cx0, Cz, cz0, x_lo, z_lo);
    return ADVH_OK;
}

extern "C" int advh_conv_wgrad2d_split_parts(int CI, int CO, int B, int H, int W) {
    const int TR = (CI == 32 && CO == 32) ? 16 : 8;
    const long ntiles = (long)B * ((H + TR - 1) / TR) * ((W + 15) / 16);
    const long grid = 256L * ((CI == 32 && CO == 32) ? 2 : 1);
    return (int)(ntiles < grid ? ntiles : grid);
}

extern "C" int advh_conv_wgrad2d_split(const advh_wgrad2d_desc* d, int CI, int CO, int Cx, int cx0, int Cz, int cz0, int64_t x_lo,
                                       int64_t dz_lo, float* dw, advh_stream_t stream) {
    if (!d || !d->X || !d->DZ || !d->partial || !dw || d->B <= 0 || d->H <= 0 || d->W_ <= 0 || d->PHx < 1 || d->PWx < 1 ||
        d->PHz < 1 || d->PWz < 1 || x_lo <= 0 || dz_lo <= 0 || x_lo % 8 || dz_lo % 8)
        return ADVH_EINVAL;
    if ((CI != 32 && CI != 64) || (CO != 32 && CO != 64)) return ADVH_EUNSUPPORTED;
    if (Cx % 8 || Cz % 8 || cx0 % 8 || cz0 % 8 || cx0 < 0 || cz0 < 0 || cx0 + CI > Cx || cz0 + CO > Cz) return ADVH_EINVAL;
    const int grid = advh_conv_wgrad2d_split_parts(CI, CO, d->B, d->H, d->W_);
    hipStream_t s = (hipStream_t)stream;
    int rc;
    // 32 x 32: 16 x 16-position tiles in one buffer (74 KiB), two workgroups per CU hide each other's loads; the wider forms run 8 x 16-position
    // tiles through a two-slot ring (the next tile's DMA under the MFMAs), CI = 64 with eight wavefronts (64 x 64 at 64 x 128 x 196: 455 us
    // against 554 us with four, 64 x 64 x 98: 112 against 156; profiles/r03_wgrad_x3.txt)
    if (CI == 32 && CO == 32) rc = launch_wgrad2d_x3<32, 32, 16, 1, 4>(*d, Cx, cx0, Cz, cz0, x_lo, dz_lo, grid, s);
    else if (CI == 64 && CO == 64) rc = launch_wgrad2d_x3<64, 64, 8, 2, 8>(*d, Cx, cx0, Cz, cz0, x_lo, dz_lo, grid, s);
    else if (CI == 32) rc = launch_wgrad2d_x3<32, 64, 8, 2, 4>(*d, Cx, cx0, Cz, cz0, x_lo, dz_lo, grid, s);
    else rc = launch_wgrad2d_x3<64, 32, 8, 2, 8>(*d, Cx, cx0, Cz, cz0, x_lo, dz_lo, grid, s);
    if (rc != ADVH_OK) return rc;
    const int n = 9 * CO * CI;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d->partial, grid, n, dw);
    return ADVH_LAUNCH_CHECK();
}
