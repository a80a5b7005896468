// Grouped positional convolution of wav2vec2 (modeling_wav2vec2.py:326-379: Conv1d(H, H, 128, padding 64, groups 16),
// last frame dropped, GELU, added to the residual stream) as an LDS line-tile kernel.
//
// As an implicit GEMM every output frame re-reads its 128-tap operand row (12 KB at 48 channels per group) through
// L2: 7.5 GB global->LDS for 64 x 3 clips, 0.72 ms at 500 TFLOP/s.  Here a workgroup owns ONE (group, clip): the clip's
// T + 127 gathered rows (advh_posconv_gather's layout, 48 or 64 channels) are staged in LDS once and every tap reads
// them there; only the group's weights (590 KB, L2-resident: all workgroups walk the groups in step) stream through a
// 3-slot LDS ring in blocks of four 32-deep k-steps.  K order = (tap, channel); a k-step's four 8-channel chunks may
// straddle two taps -- each lane group carries its own (tap, chunk) pair.  Weights are the MFMA A operand (rows =
// output channels), frames the B operand, so a lane ends up with 4 consecutive channels of one frame: 16-byte fp32
// stores of h + GELU(conv + bias).
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

constexpr int PC_K = 128, PC_RING = 3, PC_BLK = 4;               // taps; ring slots; k-steps per weight block

#define DS_READ128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define LGKM_WAIT(n) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory")

// One weight block (PC_BLK k-steps) for a wavefront that owns NJ frame tiles: the operands of k-step ks+1 are read from
// LDS between the MFMAs of k-step ks (register double buffer), so only the first reads of a block are exposed.
template <int CC, int NJ>
__device__ __forceinline__ void posconv_block(f32x4 (&acc)[4][CC / 2], unsigned wb, const unsigned (&xrow)[4], int& cc, int& tap, int fr) {
    constexpr int CG = CC * 8, NI = CG / 16, PITCH = 128, NR = NI + NJ;
    // byte offset of (tap, chunk cc) relative to the lane's frame row: row + tap, slot cc ^ ((row + tap) & 7); 16 mt is a multiple of 8
    auto xo_of = [&](int c, int t) { return (unsigned)(t * PITCH + ((c ^ ((fr + t) & 7)) << 4)); };
    if constexpr (NJ > 0) {
        f16x8 wf[2][NI], xf[2][NJ];
        unsigned xo[2];
        xo[0] = xo_of(cc, tap);
#pragma unroll
        for (int i = 0; i < NI; ++i) DS_READ128(wf[0][i], wb, i * 16 * 64);
#pragma unroll
        for (int j = 0; j < NJ; ++j) { const unsigned a = xrow[j] + xo[0]; DS_READ128(xf[0][j], a, 0); }
#pragma unroll
        for (int ks = 0; ks < PC_BLK; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            cc += 4;
            if (cc >= CC) { cc -= CC; ++tap; }
            xo[nxt] = xo_of(cc, tap);
            const unsigned wn = wb + (unsigned)((ks + 1) * CG * 64);
            LGKM_WAIT(0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < NJ * NI; ++m) {
                const int j = m / NI, i = m % NI;
                acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[cur][i], xf[cur][j], acc[j][i], 0, 0, 0);
                if (ks + 1 < PC_BLK && m < NR) {
                    if (m < NI) DS_READ128(wf[nxt][m < NI ? m : 0], wn, (m < NI ? m : 0) * 16 * 64);
                    else { const unsigned a = xrow[m - NI < NJ ? m - NI : 0] + xo[nxt]; DS_READ128(xf[nxt][m - NI < NJ ? m - NI : 0], a, 0); }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (ks + 1 < PC_BLK) {                                // fewer MFMAs than operand reads (NJ = 1): issue the rest
#pragma unroll
                for (int m = NJ * NI; m < NR; ++m) {
                    if (m < NI) DS_READ128(wf[nxt][m < NI ? m : 0], wn, (m < NI ? m : 0) * 16 * 64);
                    else { const unsigned a = xrow[m - NI < NJ ? m - NI : 0] + xo[nxt]; DS_READ128(xf[nxt][m - NI < NJ ? m - NI : 0], a, 0); }
                }
            }
        }
    } else {
#pragma unroll
        for (int ks = 0; ks < PC_BLK; ++ks) {
            cc += 4;
            if (cc >= CC) { cc -= CC; ++tap; }
        }
    }
}

template <int CC>                                                // 16-byte chunks per row = channels per group / 8 (6 or 8)
__global__ __launch_bounds__(256, 2) void posconv_tile_kernel(const advh_posconv_desc p) {
    constexpr int CG = CC * 8, NI = CG / 16, SLOTS = 8, PITCH = SLOTS * 16;   // 128-byte rows, chunk c at slot c ^ (row & 7) as in the GEMM tiles
    constexpr int KSTEPS = PC_K * CC / 4, NBLK = KSTEPS / PC_BLK, WBLK = PC_BLK * CG * 64, WL = WBLK / 16 / 256, MT = 4;
    static_assert(WBLK % (256 * 16) == 0 && KSTEPS % PC_BLK == 0, "weight block = whole loads per thread");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int rows = p.T + PC_K - 1, nt = (p.T + 15) / 16;
    const int nX = (rows * SLOTS + 63) & ~63;                    // chunks of the staged clip (whole-wave loads)
    char* Xl = lds + PC_RING * WBLK;
    const _Float16* Wg_ = (const _Float16*)p.W;
    const _Float16* X = (const _Float16*)p.xg;
    const int tiles = p.G * p.B;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int grp = tile / p.B, b = tile - grp * p.B;
        const _Float16* xs = X + ((long)grp * p.B + b) * (p.T + PC_K) * CG;
        const _Float16* wsrc = Wg_ + (long)grp * KSTEPS * CG * 32;
        __syncthreads();                                         // the previous tile's LDS reads are done
        for (int i = tid; i < nX; i += 256) {
            int row = i / SLOTS;
            const int slot = i - row * SLOTS;
            if (row >= rows) row = 0;
            const int c = slot ^ ((i / SLOTS) & 7);                 // LDS slot -> logical chunk (slots holding c >= CC are padding)
            const _Float16* src = xs + (long)row * CG + (c < CC ? c : 0) * 8;
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(Xl + (size_t)(i - lane) * 16), 16, 0, 0);
        }
        auto issue_w = [&](int blk) {                            // [4 k-steps][CG rows][32 k] -> 64-byte LDS rows, swizzled
            char* dst = lds + (blk % PC_RING) * WBLK;
            const _Float16* src = wsrc + (long)blk * (WBLK / 2);
#pragma unroll
            for (int u = 0; u < WL; ++u) {
                const int i = tid + 256 * u, rho = i >> 2, pos = i & 3;
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src + rho * 32 + ((pos ^ ((rho >> 1) & 2)) * 8)),
                                                 LDS_PTR(dst + (size_t)(i - lane) * 16), 16, 0, 0);
            }
        };
        issue_w(0);
        issue_w(1);
        f32x4 acc[MT][NI];
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int i = 0; i < NI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        // this lane group's chunk of k-step s is 4 s + g -> (tap, chunk-in-row); kept as a byte offset into the staged clip
        int cc = g, tap = 0;                                     // g < 4 <= CC: tap 0
        const unsigned lds0 = (unsigned)(unsigned long)LDS_PTR(lds);
        unsigned xrow[MT];
#pragma unroll
        for (int j = 0; j < MT; ++j) xrow[j] = lds0 + PC_RING * WBLK + (unsigned)((16 * min(wv + 4 * j, nt - 1) + fr) * PITCH);
        const int nj = max(0, min(MT, (nt - wv + 3) / 4));        // frame tiles wv, wv+4, ... < nt of this wavefront (wave-uniform)
        const unsigned wlane = lds0 + (unsigned)((fr * 4 + (g ^ ((fr >> 1) & 2))) * 16);
        for (int blk = 0; blk < NBLK; ++blk) {
            if (blk + 1 < NBLK) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WL) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                     // block blk landed for everyone; slot (blk+2) % 3 is free
            if (blk + 2 < NBLK) issue_w(blk + 2);
            const unsigned wb = wlane + (unsigned)(blk % PC_RING) * WBLK;
            switch (nj) {
                case 4: posconv_block<CC, 4>(acc, wb, xrow, cc, tap, fr); break;
                case 3: posconv_block<CC, 3>(acc, wb, xrow, cc, tap, fr); break;
                case 2: posconv_block<CC, 2>(acc, wb, xrow, cc, tap, fr); break;
                case 1: posconv_block<CC, 1>(acc, wb, xrow, cc, tap, fr); break;
                default: posconv_block<CC, 0>(acc, wb, xrow, cc, tap, fr); break;
            }
        }
        // h[b][t][grp*CG + 16 i + 4 g + r] += GELU(acc + bias)
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int mt = wv + 4 * j, t = 16 * mt + fr;
            if (mt >= nt || t >= p.T) continue;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int c = grp * CG + 16 * i + 4 * g;
                const long o = ((long)b * p.T + t) * p.H + c;
                const float4 bs = *(const float4*)(p.bias + c);
                const float4 r = *(const float4*)(p.resid + o);
                float4 v;
                v.x = r.x + gelu_fast(acc[j][i][0] + bs.x);
                v.y = r.y + gelu_fast(acc[j][i][1] + bs.y);
                v.z = r.z + gelu_fast(acc[j][i][2] + bs.z);
                v.w = r.w + gelu_fast(acc[j][i][3] + bs.w);
                *(float4*)(p.out + o) = v;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int CC> static int pc_lds_bytes(int T) {
    const int rows = T + PC_K - 1, nX = (rows * 8 + 63) & ~63;             // 8 slots per staged row
    return PC_RING * PC_BLK * CC * 8 * 64 + nX * 16;
}

}  // namespace advh

using namespace advh;

extern "C" int advh_posconv_tile_lds_bytes(int Cg, int T) {
    if (T <= 0 || T > 256) return -1;
    const int b = Cg == 48 ? pc_lds_bytes<6>(T) : (Cg == 64 ? pc_lds_bytes<8>(T) : -1);
    return b <= 160 * 1024 ? b : -1;
}

extern "C" int advh_posconv_tile_f16(const advh_posconv_desc* d, advh_stream_t stream) {
    if (!d || !d->xg || !d->W || !d->bias || !d->resid || !d->out || d->B <= 0 || d->T <= 0 || d->G <= 0 || d->H <= 0) return ADVH_EINVAL;
    if (d->H % d->G || d->K != PC_K) return ADVH_EUNSUPPORTED;
    const int Cg = d->H / d->G;
    const int lds = advh_posconv_tile_lds_bytes(Cg, d->T);
    if (lds < 0) return ADVH_EUNSUPPORTED;
    const void* fn = Cg == 48 ? (const void*)posconv_tile_kernel<6> : (const void*)posconv_tile_kernel<8>;
    if (advh_ensure_lds(fn) != ADVH_OK) return ADVH_ELAUNCH;
    const long tiles = (long)d->G * d->B;
    const long per_cu = lds <= 80 * 1024 ? 2 : 1;
    long grid = 256 * per_cu;
    if (grid > tiles) grid = tiles;
    hipStream_t s = (hipStream_t)stream;
    if (Cg == 48) hipLaunchKernelGGL(posconv_tile_kernel<6>, dim3((unsigned)grid), dim3(256), lds, s, *d);
    else hipLaunchKernelGGL(posconv_tile_kernel<8>, dim3((unsigned)grid), dim3(256), lds, s, *d);
    return ADVH_LAUNCH_CHECK();
}
