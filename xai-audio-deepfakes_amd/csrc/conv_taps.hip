// LDS line-tile convolution for narrow layers (C = 32 or 64 channels in and out): the HiFi-GAN V1 MRF
// ResBlock Conv1d layers of the two late stages (k = 3 / 7 / 11, dilation 1 / 3 / 5) -- hifigan.py:106-110, 180 via
// SpeechBrain's generator -- and any other "same" convolution that is a sum of taps at constant row offsets of a
// zero-haloed channels-last map (a 3x3 Conv2d on a padded NHWC map is 9 such taps).
//
// With 32 / 64 channels the implicit GEMM re-reads the input once per tap through L2 and is bound by the
// global->LDS path; here a persistent workgroup keeps the WHOLE weight tensor (k*C*C fp16 <= 90 KiB) resident in
// LDS, streams position tiles through a line buffer (tile + halo rows, loaded once, contiguous in HBM) and feeds
// the matrix cores from LDS only: HBM traffic = input once + output once.
// MFMA orientation as in gemm.hip: weights are the A operand (rows = output channels), positions the B operand,
// so a lane holds 4 consecutive output channels of one position.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include "addvisor_hip.h"
#include "common.h"
#include "device_math.h"

namespace advh {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define DS_READ128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define LGKM_WAIT(n) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory")

// 16-byte chunk c of LDS row r is stored at slot c ^ swz(r): conflict-free ds_read_b128 for 16 consecutive rows
// starting anywhere (enumerated in tests/test_gemm_plan.py)
template <int C> __device__ __forceinline__ int swz(int r) { return C == 64 ? (r & 7) : ((r >> 1) & 2); }

// MFMA row R = 16 i + 4 g + r of the weight operand carries output channel cout_of(R): the accumulators of the tile
// pair (2q, 2q+1) of one lane are then the 8 CONSECUTIVE channels 32 q + 8 g .. + 7 -> 16-byte stores, 64 contiguous
// bytes per position and store instruction (the row permutation is applied on the source side of the weight DMA).
__device__ __forceinline__ int cout_of(int R) { return ((R >> 5) << 5) + (((R >> 2) & 3) << 3) + (((R >> 4) & 1) << 2) + (R & 3); }

// NJ = 16-position column tiles per wavefront; a workgroup (4 wavefronts) owns TT = 64 * NJ positions per tile.
// The line buffer is double-buffered: the DMA of tile i+1 runs under the MFMAs and the stores of tile i.
template <int C, int NJ, int NW = 4>
__global__ __launch_bounds__(64 * NW) void conv_taps_kernel(const advh_taps_desc p) {
    constexpr int CH = C / 8, CT = C / 16, KS = C / 32, TT = 16 * NJ * NW, NTH = 64 * NW;   // NW wavefronts x NJ column tiles of 16 positions
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int fr = lane & 15, g = lane >> 4;
    int lo = 0, hi = 0;
    for (int t = 0; t < p.ntap; ++t) { lo = min(lo, p.toff[t]); hi = max(hi, p.toff[t]); }
    const int SR = TT + hi - lo;                                   // line-buffer rows
    const int SRC = (SR * CH + 63) & ~63;                          // chunks per buffer (whole wave loads)
    char* Wl = lds;                                                // [ntap*C][C] halfs
    char* Xl = lds + (size_t)p.ntap * C * C * 2;                   // 2 x [SR][C] halfs

    // ---- weights: once per workgroup
    const _Float16* Wg = (const _Float16*)p.W;
    for (int i = tid; i < p.ntap * C * CH; i += NTH) {             // i = lds chunk index (wave-linear)
        int row = i / CH, pos = i % CH;
        const _Float16* src = Wg + ((long)(row / C) * C + cout_of(row % C)) * C + ((pos ^ swz<C>(row)) * 8);
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(Wl + (size_t)(i - lane) * 16), 16, 0, 0);
    }
    const _Float16* X = (const _Float16*)p.X;
    const int ntiles = (p.M + TT - 1) / TT;
    // rows p0+lo .. p0+TT+hi, clamped: rows outside the map only ever feed halo outputs (written as zeros)
    auto load_lines = [&](int tile, int buf) {
        const long p0 = (long)tile * TT + lo;
        char* dst = Xl + (size_t)buf * SRC * 16;
        for (int i = tid; i < SRC; i += NTH) {
            int row = i / CH, pos = i % CH;
            long r = p0 + row;
            r = r < 0 ? 0 : (r >= p.M ? p.M - 1 : r);
            const _Float16* src = X + r * C + ((pos ^ swz<C>(row)) * 8);
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(dst + (size_t)(i - lane) * 16), 16, 0, 0);
        }
    };
    float4 bias[CT];                                               // bias[2q + e] = channels 32 q + 8 g + 4 e .. + 3
#pragma unroll
    for (int i = 0; i < CT; ++i)
        bias[i] = p.bias ? *(const float4*)(p.bias + (i >> 1) * 32 + g * 8 + (i & 1) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);

    const int NS = p.ntap * KS;                                    // 32-deep k steps
    const unsigned lds0 = (unsigned)(unsigned long)LDS_PTR(lds);
    int toffv = 0;                                                 // lane t holds toff[t]: v_readlane in the loop, no memory op
#pragma unroll
    for (int t = 0; t < 16; ++t) toffv = (lane == t) ? p.toff[t] : toffv;
    int buf = 0;
    if ((int)blockIdx.x < ntiles) load_lines(blockIdx.x, 0);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, buf ^= 1) {
        const long p0 = (long)tile * TT;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                           // buffer `buf` landed; everyone left buffer buf^1
        if (tile + (int)gridDim.x < ntiles) load_lines(tile + gridDim.x, buf ^ 1);
        const unsigned xl = lds0 + (unsigned)p.ntap * (C * C * 2) + (unsigned)buf * SRC * 16;
        if (p.pre_act) {
            // LeakyReLU applied to the line buffer in place: the producer then stores only the raw map (the residual
            // path needs it anyway) instead of a second, pre-activated copy -- one map write and one map read less
            char* xb = Xl + (size_t)buf * SRC * 16;
            const _Float16 sl = (_Float16)p.pre_slope;
            for (int i = tid; i < SRC; i += NTH) {
                f16x8 v = *(f16x8*)(xb + (size_t)i * 16);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = v[j] > (_Float16)0 ? v[j] : v[j] * sl;
                *(f16x8*)(xb + (size_t)i * 16) = v;
            }
            __syncthreads();
        }

        f32x4 acc[CT][NJ];
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // Fragment pipeline: the CT + NJ ds_reads of k-step s+1 are issued one per MFMA gap of step s.  The reads are
        // inline asm (hipcc would wait lgkmcnt(0) at the first use AND after the scalar tap-offset lookup), so each
        // step starts with the one wait that covers exactly the reads of the previous gap sequence.
        auto issue = [&](int s, int m, f16x8 (&wf)[CT], f16x8 (&xf)[NJ], unsigned wa_, unsigned xa_) {
            (void)s;
#pragma unroll
            for (int i = 0; i < CT; ++i)
                if (m == i) DS_READ128(wf[i], wa_, i * 16 * C * 2);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                if (m == CT + j) DS_READ128(xf[j], xa_, j * 16 * C * 2);
        };
        auto addr = [&](int s, unsigned& wa_, unsigned& xa_) {
            const int t = s / KS, c = (s % KS) * 4 + g;
            wa_ = lds0 + (unsigned)t * (C * C * 2) + (fr * CH + (c ^ swz<C>(fr))) * 16;
            const int row = wv * (16 * NJ) + fr + __builtin_amdgcn_readlane(toffv, t) - lo;
            xa_ = xl + (row * CH + (c ^ swz<C>(row))) * 16;
        };
        auto step = [&](int sn, const f16x8 (&wc)[CT], const f16x8 (&xc)[NJ], f16x8 (&wn)[CT], f16x8 (&xn)[NJ]) {
            unsigned wa_, xa_;
            addr(sn, wa_, xa_);
            LGKM_WAIT(0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < CT * NJ; ++m) {
                const int i = m / NJ, j = m % NJ;
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wc[i], xc[j], acc[i][j], 0, 0, 0);
                if (m < CT + NJ) issue(sn, m, wn, xn, wa_, xa_);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        f16x8 wa[CT], xa[NJ], wb[CT], xb[NJ];
        {
            unsigned wa_, xa_;
            addr(0, wa_, xa_);
#pragma unroll
            for (int m = 0; m < CT + NJ; ++m) issue(0, m, wa, xa, wa_, xa_);
        }
        for (int s = 0; s < NS; s += 2) {
            step(min(s + 1, NS - 1), wa, xa, wb, xb);
            if (s + 1 < NS) step(min(s + 2, NS - 1), wb, xb, wa, xa);
        }
        LGKM_WAIT(0);                                              // the last (redundant) prefetch must land before its registers are reused
        __builtin_amdgcn_sched_barrier(0);
        // ---- epilogue: bias, LeakyReLU, residual, fp16 stores (and the pre-activated copy)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const long pos = p0 + wv * (16 * NJ) + j * 16 + fr;
            if (pos >= p.M) continue;
            unsigned w = (unsigned)(pos % p.Wg), tq = (unsigned)(pos / p.Wg);
            unsigned h = tq % (unsigned)p.Hg;
            const bool ok = (int)h >= p.h0 && (int)h < p.h1 && (int)w >= p.w0 && (int)w < p.w1;
#pragma unroll
            for (int q = 0; q < CT / 2; ++q) {
                const long o = pos * C + q * 32 + g * 8;
                float v[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[r] = acc[2 * q][j][r]; v[4 + r] = acc[2 * q + 1][j][r]; }
                if (ok) {
                    v[0] += bias[2 * q].x; v[1] += bias[2 * q].y; v[2] += bias[2 * q].z; v[3] += bias[2 * q].w;
                    v[4] += bias[2 * q + 1].x; v[5] += bias[2 * q + 1].y; v[6] += bias[2 * q + 1].z; v[7] += bias[2 * q + 1].w;
                    if (p.act == ADVH_ACT_LEAKY) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] = v[r] > 0.f ? v[r] : p.slope * v[r];
                    }
                    if (p.resid) {
                        f16x8 rr = *(const f16x8*)((const _Float16*)p.resid + o);
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] += (float)rr[r];
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] = 0.f;
                }
                f16x8 hv;
#pragma unroll
                for (int r = 0; r < 8; ++r) hv[r] = (_Float16)v[r];
                *(f16x8*)((_Float16*)p.out_h + o) = hv;
                if (p.out_h2) {
                    f16x8 h2;
#pragma unroll
                    for (int r = 0; r < 8; ++r) h2[r] = (_Float16)(v[r] > 0.f ? v[r] : p.slope2 * v[r]);
                    *(f16x8*)((_Float16*)p.out_h2 + o) = h2;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------------------------------
// 2-D variant: 3x3 stride-1 "same" Conv2d on a zero-haloed NHWC map (the 32- and 64-channel ConvBlock second
// convolutions of the U-Net, addvisor.py:12-25), C_in = C_out = C.  A workgroup owns 16 x 16 output positions; the line
// buffer is the 18 x 18 input patch (1.27 x over-read instead of the 2.5 x of a 1-D line tile on a 198-wide map), so a tap
// (kh, kw) of output (ly, lx) is the patch row (ly + kh) * 18 + lx + kw: again a constant row offset.  Wavefront w owns
// tile rows 4w .. 4w+3 (one 16-position column tile each).  Only interior positions are written; the destination's halo
// stays as allocated (zero).
// NW wavefronts per workgroup: each owns 16 / NW rows of the 16 x 16 tile (NW = 8 for the 64-channel instance, whose 156 KB of
// LDS allow one workgroup per CU: eight wavefronts hide the LDS / store latency four could not)
template <int C, int NW = 4>
__global__ __launch_bounds__(64 * NW) void conv_taps2d_kernel(const advh_taps2d_desc p) {
    constexpr int CH = C / 8, CT = C / 16, KS = C / 32, NJ = 16 / NW, NTH = 64 * NW, PR = 18, SR = PR * PR;
    constexpr int SRC = (SR * CH + 63) & ~63;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int fr = lane & 15, g = lane >> 4;
    char* Wl = lds;
    char* Xl = lds + 9 * C * C * 2;
    const _Float16* Wg = (const _Float16*)p.W;
    for (int i = tid; i < 9 * C * CH; i += NTH) {
        int row = i / CH, pos = i % CH;
        const _Float16* src = Wg + ((long)(row / C) * C + cout_of(row % C)) * C + ((pos ^ swz<C>(row)) * 8);
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(Wl + (size_t)(i - lane) * 16), 16, 0, 0);
    }
    const _Float16* X = (const _Float16*)p.X;
    const int Hp = p.H + 2 * p.PH, Wp = p.W_ + 2 * p.PW;
    const int tx = (p.W_ + 15) / 16, ty = (p.H + 15) / 16, ntiles = p.B * ty * tx;
    auto origin = [&](int tile, int& b, int& y0, int& x0) {
        x0 = (tile % tx) * 16;
        const int r = tile / tx;
        y0 = (r % ty) * 16;
        b = r / ty;
    };
    auto load_patch = [&](int tile, int buf) {
        int b, y0, x0;
        origin(tile, b, y0, x0);
        char* dst = Xl + (size_t)buf * SRC * 16;
        for (int i = tid; i < SRC; i += NTH) {
            int row = i / CH, pos = i % CH;
            if (row >= SR) row = 0;
            // padded coordinates of patch row `row`, clamped into the map (clamped rows only feed skipped outputs)
            int gy = min(y0 + p.PH - 1 + row / PR, Hp - 1), gx = min(x0 + p.PW - 1 + row % PR, Wp - 1);
            const _Float16* src = X + (((long)b * Hp + gy) * Wp + gx) * C + ((pos ^ swz<C>(i / CH)) * 8);
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(dst + (size_t)(i - lane) * 16), 16, 0, 0);
        }
    };
    float4 bias[CT];
#pragma unroll
    for (int i = 0; i < CT; ++i)
        bias[i] = p.bias ? *(const float4*)(p.bias + (i >> 1) * 32 + g * 8 + (i & 1) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned lds0 = (unsigned)(unsigned long)LDS_PTR(lds);
    constexpr int NS = 9 * KS;
    int buf = 0;
    if ((int)blockIdx.x < ntiles) load_patch(blockIdx.x, 0);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, buf ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) load_patch(tile + gridDim.x, buf ^ 1);
        const unsigned xl = lds0 + 9 * C * C * 2 + (unsigned)buf * SRC * 16;
        f32x4 acc[CT][NJ];
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        unsigned wa_[2], xa_[2][NJ];
        auto addr = [&](int s, unsigned& wa, unsigned (&xa)[NJ]) {
            const int t = s / KS, c = (s % KS) * 4 + g, kh = t / 3, kw = t - kh * 3;
            wa = lds0 + (unsigned)t * (C * C * 2) + (fr * CH + (c ^ swz<C>(fr))) * 16;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int row = (wv * NJ + j + kh) * PR + kw + fr;
                xa[j] = xl + (row * CH + (c ^ swz<C>(row))) * 16;
            }
        };
        f16x8 wf[2][CT], xf[2][NJ];
        addr(0, wa_[0], xa_[0]);
#pragma unroll
        for (int i = 0; i < CT; ++i) DS_READ128(wf[0][i], wa_[0], i * 16 * C * 2);
#pragma unroll
        for (int j = 0; j < NJ; ++j) DS_READ128(xf[0][j], xa_[0][j], 0);
#pragma unroll
        for (int s = 0; s < NS; ++s) {                              // 9 taps x KS k-steps, fully unrolled: static register double buffer
            const int cur = s & 1, nxt = cur ^ 1;
            if (s + 1 < NS) addr(s + 1, wa_[nxt], xa_[nxt]);
            LGKM_WAIT(0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < CT * NJ; ++m) {
                const int i = m / NJ, j = m % NJ;
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[cur][i], xf[cur][j], acc[i][j], 0, 0, 0);
                if (s + 1 < NS) {
                    if (m < CT) DS_READ128(wf[nxt][m < CT ? m : 0], wa_[nxt], (m < CT ? m : 0) * 16 * C * 2);
                    else if (m < CT + NJ) DS_READ128(xf[nxt][m - CT < NJ ? m - CT : 0], xa_[nxt][m - CT < NJ ? m - CT : 0], 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        int b, y0, x0;
        origin(tile, b, y0, x0);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int gy = y0 + wv * NJ + j, gx = x0 + fr;
            if (gy >= p.H || gx >= p.W_) continue;
            const long pos = ((long)b * Hp + gy + p.PH) * Wp + gx + p.PW;
#pragma unroll
            for (int q = 0; q < CT / 2; ++q) {
                const long o = pos * C + q * 32 + g * 8;
                float v[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[r] = acc[2 * q][j][r]; v[4 + r] = acc[2 * q + 1][j][r]; }
                v[0] += bias[2 * q].x; v[1] += bias[2 * q].y; v[2] += bias[2 * q].z; v[3] += bias[2 * q].w;
                v[4] += bias[2 * q + 1].x; v[5] += bias[2 * q + 1].y; v[6] += bias[2 * q + 1].z; v[7] += bias[2 * q + 1].w;
                f16x8 hv;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float a = (p.act == ADVH_ACT_LEAKY && v[r] < 0.f) ? p.slope * v[r] : v[r];
                    hv[r] = (_Float16)a;
                }
                *(f16x8*)((_Float16*)p.out_h + o) = hv;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------------------------------
// fp32-class form for the 64-channel stage (round 3): X, W, resid, out_h, out_h2 are split-format plane pairs (hi plane, lo plane a fixed
// distance behind: advh_conv_taps_split's *_lo), every fragment pair costs three MFMAs in the order of gemm_x3_kernel (accx += Wh Xl;
// acc += Wh Xh; accx += Wl Xh; result acc + accx * 2^-11) and the K order is the GEMM's (tap-major, two 32-deep steps per tap): the outputs
// are bit-identical to the x3 implicit GEMM's (tests/test_gpu_hifigan.py).  Two planes of an 11-tap 64 x 64 weight tensor are 180 KiB, so the
// weights are NOT resident: they stream tap by tap (16 KiB per tap, both planes) through a four-slot LDS ring, tap n + 3 requested when tap n
// starts (an LDS DMA takes ~1 700 cycles from issue to landing); ONE line buffer (256 positions + halo, both planes): the next tile's lines are
// requested after the last tap and land under the epilogue.  Eight wavefronts of 64 channels x 32 positions (64 accumulator registers, two
// wavefronts per SIMD): the kernel is LDS-bandwidth-bound by construction -- 12 KiB of fragments per 24 MFMAs and wavefront = 125 B/clk per CU
// at full matrix rate -- and reaches 280 - 327 TFLOP/s on the k = 7 / 11 ResBlock convolutions of HiFi-GAN's 64-channel stage against 240 - 287
// for the x3 implicit GEMM on 256 x 64 tiles (which re-reads the input once per tap through the L2 -> LDS path); with four wavefronts of
// 64 x 64 (one per SIMD, nothing to cover LDS time and the per-tap barrier with) it ran level with the GEMM: profiles/r03_conv_taps_x3_experiment.txt.
template <int NJ, int NW>
__global__ __launch_bounds__(64 * NW) void conv_taps_x3_kernel(const advh_taps_desc p, long x_lo, long w_lo, long r_lo, long o_lo) {
    constexpr int C = 64, CH = 8, CT = 4, KS = 2, TT = 16 * NJ * NW, NTH = 64 * NW, WTAP = C * C * 2, WPT = 2 * (C * CH / NTH);   // WPT: DMA instructions per thread and tap     // WTAP: bytes of one plane of one tap
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int fr = lane & 15, g = lane >> 4;
    int lo = 0, hi = 0;
    for (int t = 0; t < p.ntap; ++t) { lo = min(lo, p.toff[t]); hi = max(hi, p.toff[t]); }
    const int SR = TT + hi - lo;                                   // line-buffer rows
    const int SRC = (SR * CH + 63) & ~63;                          // chunks per plane and buffer (whole wave loads)
    constexpr int NSLOT = 4, AHEAD = NSLOT - 1;                    // weight ring: tap n + 3 is requested when tap n starts
    char* Wl = lds;                                                // NSLOT slots x [hi | lo] x [C][C] halfs
    char* Xl = lds + NSLOT * 2 * WTAP;                             // [hi | lo] x [SR][C] halfs (one buffer: the next tile's lines are requested
                                                                   // when the last tap's MFMAs are done and arrive under the epilogue)
    const _Float16* Wg = (const _Float16*)p.W;
    const _Float16* X = (const _Float16*)p.X;
    const int ntiles = (p.M + TT - 1) / TT;
    auto load_weights = [&](int t, int slot) {
        char* dst = Wl + (size_t)slot * 2 * WTAP;
        for (int i = tid; i < C * CH; i += NTH) {                  // i = lds chunk index (wave-linear): row = output-channel slot, pos = chunk
            const int row = i / CH, pos = i % CH;
            const _Float16* src = Wg + ((long)t * C + cout_of(row)) * C + ((pos ^ swz<C>(row)) * 8);
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(dst + (size_t)(i - lane) * 16), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src + w_lo), LDS_PTR(dst + WTAP + (size_t)(i - lane) * 16), 16, 0, 0);
        }
    };
    auto load_lines = [&](int tile) {                              // rows p0+lo .. p0+TT+hi, clamped: rows outside the map only feed halo outputs
        const long p0 = (long)tile * TT + lo;
        char* dst = Xl;
        for (int i = tid; i < SRC; i += NTH) {
            int row = i / CH, pos = i % CH;
            long r = p0 + row;
            r = r < 0 ? 0 : (r >= p.M ? p.M - 1 : r);
            const _Float16* src = X + r * C + ((pos ^ swz<C>(row)) * 8);
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(dst + (size_t)(i - lane) * 16), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src + x_lo), LDS_PTR(dst + (size_t)SRC * 16 + (size_t)(i - lane) * 16), 16, 0, 0);
        }
    };
    float4 bias[CT];                                               // bias[2q + e] = channels 32 q + 8 g + 4 e .. + 3
#pragma unroll
    for (int i = 0; i < CT; ++i)
        bias[i] = p.bias ? *(const float4*)(p.bias + (i >> 1) * 32 + g * 8 + (i & 1) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    int toffv = 0;                                                 // lane t holds toff[t]
#pragma unroll
    for (int t = 0; t < 16; ++t) toffv = (lane == t) ? p.toff[t] : toffv;

    // global tap counter n (tap n % ntap of the workgroup's n / ntap-th tile) lives in slot n % NSLOT; taps 0 .. AHEAD-1 and the first lines up front
    const int mytiles = (int)blockIdx.x < ntiles ? (ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int ntot = mytiles * p.ntap;
    for (int n = 0; n < AHEAD && n < ntot; ++n) load_weights(n % p.ntap, n % NSLOT);
    if (mytiles) load_lines(blockIdx.x);
    int n = 0;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long p0 = (long)tile * TT;
        f32x4 acc[CT][NJ], accx[CT][NJ];
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) { acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; accx[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        const char* xb = Xl;
        // Fragment pipeline (one wavefront per SIMD: nothing else hides LDS time): the 16 ds_read_b128 of the NEXT 32-deep step are issued before
        // the 48 MFMAs of the current one.  The step after a tap's second one belongs to the next tap, so the barrier at the top of tap n
        // already covers the weights of tap n + 1 (requested three taps ago); tap n + 3 is requested right after it.
        f16x8 fwh[2][CT], fwl[2][CT], fxh[2][NJ], fxl[2][NJ];
        auto fetch = [&](int set, int nn, int tt, int ks) {
            const char* wb = Wl + (size_t)(nn % NSLOT) * 2 * WTAP;
            const int rowb = wv * (16 * NJ) + fr + __builtin_amdgcn_readlane(toffv, tt) - lo;
            const int c = ks * 4 + g;
#pragma unroll
            for (int i = 0; i < CT; ++i) {
                const int wo = ((i * 16 + fr) * CH + (c ^ swz<C>(fr))) * 16;             // (16 i + fr) & 7 == fr & 7
                fwh[set][i] = *(const f16x8*)(wb + wo);
                fwl[set][i] = *(const f16x8*)(wb + WTAP + wo);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int row = rowb + j * 16;
                const int xo = (row * CH + (c ^ swz<C>(row))) * 16;
                fxh[set][j] = *(const f16x8*)(xb + xo);
                fxl[set][j] = *(const f16x8*)(xb + (size_t)SRC * 16 + xo);
            }
        };
        auto mma = [&](int set) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
#pragma unroll
                for (int i = 0; i < CT; ++i) accx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fwh[set][i], fxl[set][j], accx[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < CT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fwh[set][i], fxh[set][j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < CT; ++i) accx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fwl[set][i], fxh[set][j], accx[i][j], 0, 0, 0);
            }
        };
        for (int t = 0; t < p.ntap; ++t, ++n) {
            // t = 0: the tile's lines, the previous epilogue's stores and taps n, n + 1 must have landed (everything); later: all but the 4 DMA
            // instructions per thread of the youngest requested tap (n + 2)
            if (t == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPT) : "memory");
            __syncthreads();                                       // ... everyone's pieces; every wavefront is done with tap n - 1 (its slot is free).
            // (The fence of __syncthreads() makes the compiler wait vmcnt(0) here, i.e. for the younger taps' pieces too; a counted wait + bare
            // s_barrier keeps them in flight but measured no faster -- the pieces are L2 hits that land well inside a tap -- so the fenced form stays.)
            if (n + AHEAD < ntot) load_weights((n + AHEAD) % p.ntap, (n + AHEAD) % NSLOT);
            if (t == 0) fetch(0, n, 0, 0);
            fetch(1, n, t, 1);
            __builtin_amdgcn_sched_barrier(0);
            mma(0);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < p.ntap) fetch(0, n + 1, t + 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            mma(1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (tile + (int)gridDim.x < ntiles) {
            __syncthreads();                                       // every wavefront has read its last fragments of this tile's lines
            load_lines(tile + gridDim.x);
        }
        // ---- epilogue: bias, LeakyReLU, residual, split stores (and the pre-activated copy)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const long pos = p0 + wv * (16 * NJ) + j * 16 + fr;
            if (pos >= p.M) continue;
            unsigned w = (unsigned)(pos % p.Wg), tq = (unsigned)(pos / p.Wg);
            unsigned h = tq % (unsigned)p.Hg;
            const bool ok = (int)h >= p.h0 && (int)h < p.h1 && (int)w >= p.w0 && (int)w < p.w1;
#pragma unroll
            for (int q = 0; q < CT / 2; ++q) {
                const long o = pos * C + q * 32 + g * 8;
                float v[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = fmaf(accx[2 * q][j][r], SPLIT_LO_INV, acc[2 * q][j][r]);
                    v[4 + r] = fmaf(accx[2 * q + 1][j][r], SPLIT_LO_INV, acc[2 * q + 1][j][r]);
                }
                if (ok) {
                    v[0] += bias[2 * q].x; v[1] += bias[2 * q].y; v[2] += bias[2 * q].z; v[3] += bias[2 * q].w;
                    v[4] += bias[2 * q + 1].x; v[5] += bias[2 * q + 1].y; v[6] += bias[2 * q + 1].z; v[7] += bias[2 * q + 1].w;
                    if (p.act == ADVH_ACT_LEAKY) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] = v[r] > 0.f ? v[r] : p.slope * v[r];
                    }
                    if (p.resid) {
                        float rr[8];
                        load_h_rt<8>((const _Float16*)p.resid, o, r_lo, rr);
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] += rr[r];
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] = 0.f;
                }
                store_h_rt<8>((_Float16*)p.out_h, o, o_lo, v);
                if (p.out_h2) {
                    float v2[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r) v2[r] = v[r] > 0.f ? v[r] : p.slope2 * v[r];
                    store_h_rt<8>((_Float16*)p.out_h2, o, o_lo, v2);
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

static int taps_x3_lds(int span, int nj) { return 4 * 2 * 64 * 64 * 2 + 2 * ((((64 * nj + span) * 8 + 63) / 64 * 64) * 16); }   // 4 weight slots + one line buffer, two planes each

static int taps_span(const advh_taps_desc* d) {
    int lo = 0, hi = 0;
    for (int t = 0; t < d->ntap; ++t) { lo = d->toff[t] < lo ? d->toff[t] : lo; hi = d->toff[t] > hi ? d->toff[t] : hi; }
    return hi - lo;
}

static int taps_lds(int C, int ntap, int span, int nj) {
    return ntap * C * C * 2 + 2 * (((64 * nj + span) * (C / 8) + 63) / 64 * 64) * 16;
}

}  // namespace advh

using namespace advh;

// column tiles per wavefront: the widest tile whose weights + two line buffers fit the 160 KiB of LDS
extern "C" int advh_conv_taps_tile(int C, int ntap, int span) {
    for (int nj = 4; nj >= 2; --nj)
        if (taps_lds(C, ntap, span, nj) <= 160 * 1024) return 64 * nj;
    return 0;
}

extern "C" int advh_conv_taps_lds_bytes(int C, int ntap, int span) {
    const int tt = advh_conv_taps_tile(C, ntap, span);
    return tt ? taps_lds(C, ntap, span, tt / 64) : -1;
}

extern "C" int advh_conv_taps_f16(const advh_taps_desc* d, int C, advh_stream_t stream) {
    if (!d || !d->X || !d->W || !d->out_h || d->M <= 0 || d->ntap <= 0 || d->ntap > 16 || d->Hg <= 0 || d->Wg <= 0) return ADVH_EINVAL;
    if (C != 32 && C != 64) return ADVH_EUNSUPPORTED;
    if (d->act != ADVH_ACT_NONE && d->act != ADVH_ACT_LEAKY) return ADVH_EINVAL;
    const int span = taps_span(d);
    const int tt = advh_conv_taps_tile(C, d->ntap, span);
    if (!tt) return ADVH_EUNSUPPORTED;
    const int nj = tt / 64, lds = taps_lds(C, d->ntap, span, nj);
    typedef void (*kern_t)(const advh_taps_desc);
    static const kern_t kerns[2][3] = {{conv_taps_kernel<32, 2>, conv_taps_kernel<32, 3>, conv_taps_kernel<32, 4>},
                                       {conv_taps_kernel<64, 2>, conv_taps_kernel<64, 2, 6>, conv_taps_kernel<64, 2, 8>}};   // 192 / 256 positions: six / eight wavefronts of two column tiles
    const int ci = C == 64, ji = nj - 2;
    if (advh_ensure_lds((const void*)kerns[ci][ji]) != ADVH_OK) return ADVH_ELAUNCH;
    const int ntiles = (d->M + tt - 1) / tt;
    const int per_cu = lds <= 40 * 1024 ? 4 : (lds <= 53 * 1024 ? 3 : (lds <= 80 * 1024 ? 2 : 1));
    int grid = 256 * per_cu;
    if (grid > ntiles) grid = ntiles;
    hipLaunchKernelGGL(kerns[ci][ji], dim3(grid), dim3(ci && nj == 4 ? 512 : (ci && nj == 3 ? 384 : 256)), lds, (hipStream_t)stream, *d);
    return ADVH_LAUNCH_CHECK();
}

extern "C" int advh_conv_taps2d_f16(const advh_taps2d_desc* d, int C, advh_stream_t stream) {
    if (!d || !d->X || !d->W || !d->out_h || d->B <= 0 || d->H <= 0 || d->W_ <= 0 || d->PH < 1 || d->PW < 1) return ADVH_EINVAL;
    if (C != 32 && C != 64) return ADVH_EUNSUPPORTED;
    if (d->act != ADVH_ACT_NONE && d->act != ADVH_ACT_LEAKY) return ADVH_EINVAL;
    const int lds = 9 * C * C * 2 + 2 * ((18 * 18 * (C / 8) + 63) / 64 * 64) * 16;
    const int ci = C == 64;
    const void* fn = ci ? (const void*)conv_taps2d_kernel<64, 8> : (const void*)conv_taps2d_kernel<32>;
    if (advh_ensure_lds(fn) != ADVH_OK) return ADVH_ELAUNCH;
    const long ntiles = (long)d->B * ((d->H + 15) / 16) * ((d->W_ + 15) / 16);
    const int per_cu = lds <= 80 * 1024 ? 2 : 1;
    long grid = 256L * per_cu;
    if (grid > ntiles) grid = ntiles;
    if (ci) hipLaunchKernelGGL((conv_taps2d_kernel<64, 8>), dim3((unsigned)grid), dim3(512), lds, (hipStream_t)stream, *d);
    else hipLaunchKernelGGL(conv_taps2d_kernel<32>, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, *d);
    return ADVH_LAUNCH_CHECK();
}

// column tiles of the split-arithmetic 64-channel kernel: 4 (256 positions per tile) where weights ring + two line buffers fit, else 3
extern "C" int advh_conv_taps_split_tile(int C, int ntap, int span) {
    if (C != 64 || ntap <= 0 || ntap > 16 || span < 0) return 0;
    return taps_x3_lds(span, 4) <= 160 * 1024 ? 256 : 0;
}

extern "C" int advh_conv_taps_split(const advh_taps_desc* d, int C, int64_t x_lo, int64_t w_lo, int64_t r_lo, int64_t o_lo, advh_stream_t stream) {
    if (!d || !d->X || !d->W || !d->out_h || d->M <= 0 || d->ntap <= 0 || d->ntap > 16 || d->Hg <= 0 || d->Wg <= 0) return ADVH_EINVAL;
    if (x_lo <= 0 || w_lo <= 0 || o_lo <= 0 || (d->resid && r_lo <= 0) || x_lo % 8 || w_lo % 8 || o_lo % 8 || r_lo % 8) return ADVH_EINVAL;
    if (C != 64 || d->pre_act) return ADVH_EUNSUPPORTED;
    if (d->act != ADVH_ACT_NONE && d->act != ADVH_ACT_LEAKY) return ADVH_EINVAL;
    const int span = taps_span(d);
    const int tt = advh_conv_taps_split_tile(C, d->ntap, span);
    if (!tt) return ADVH_EUNSUPPORTED;
    const int lds = taps_x3_lds(span, 4);
    // eight wavefronts of 64 channels x 32 positions: two per SIMD (64 accumulator registers each) cover each other's LDS and barrier time --
    // with four wavefronts of 64 x 64 (one per SIMD) the kernel ran level with the implicit GEMM (profiles/r03_conv_taps_x3_experiment.txt)
    const void* fn = (const void*)conv_taps_x3_kernel<2, 8>;
    if (advh_ensure_lds(fn) != ADVH_OK) return ADVH_ELAUNCH;
    const int ntiles = (d->M + tt - 1) / tt;
    const int grid = ntiles < 256 ? ntiles : 256;
    hipLaunchKernelGGL((conv_taps_x3_kernel<2, 8>), dim3(grid), dim3(512), lds, (hipStream_t)stream, *d, (long)x_lo, (long)w_lo, (long)r_lo, (long)o_lo);
    return ADVH_LAUNCH_CHECK();
}

ADVH_SPLIT_FLAG_SETTER(advh_split_flag_conv_taps)
