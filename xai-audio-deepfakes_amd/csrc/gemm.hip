// Implicit-GEMM on the gfx950 matrix cores: fp16 operands, fp32 accumulate.
//
//     out[row(m), col(n)] = act( sum_k A(m, k) * W[n][k] + bias[n] ) + resid[row(m), col(n)]
//
// One kernel serves every dense contraction of the hot path:
//   * wav2vec2 feature-encoder Conv1d layers 1-6 (transformers/.../modeling_wav2vec2.py:254-323):
//     channels-last activations make the im2col row of output t the CONTIGUOUS slab
//     x[s*t : s*t+k, :], so the conv is a GEMM whose A rows overlap (row stride s*C, K = k*C);
//   * the grouped positional Conv1d (k=128, 16 groups; :326-379) the same way, batched over groups;
//   * every Linear of the encoder (:438-572) and the feature projection (:422-434);
//   * the U-Net's Conv2d / dilated Conv2d / ConvTranspose2d (addvisor.py:12-84) on zero-haloed
//     NHWC maps: a tap is a constant offset from the row's base address, so K walks a table of
//     16-byte chunk offsets (`ktab`), optionally over two sources (skip-concat by pointer).
//
// Structure (cdna_hip_programming.md §5): 256 threads = 4 wavefronts, BMxBNx64 tile, both
// operands staged global->LDS with 16-byte `global_load_lds` (per-lane source address, linear LDS
// image, XOR swizzle applied on the SOURCE side and on the ds_read side), v_mfma_f32_16x16x32_f16,
// several resident workgroups per CU for overlap.  W is the MFMA A operand and the activations the
// B operand, so a lane's 4 accumulator registers are 4 CONSECUTIVE output channels of one row:
// the epilogue stores 8 B (fp16) / 16 B (fp32) per lane without any shuffle.
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

constexpr int BK = 64;          // halfs per K-step = 8 chunks of 16 B
#ifdef ADVH_STAMPS                // diagnostic build only (-DADVH_STAMPS): core-clock and 100 MHz stamps around one workgroup of the x3 / fp16 tiles
__device__ long long g_gemm_stamps[8];
__device__ long long g_kstep[8];               // wave 0 of the mid-grid workgroup, K-step 6: core-clock stamps inside the step
__device__ long long g_wg_rec[4 * 16384];   // per workgroup of the last x3 launch: start, K loop end, end (100 MHz ticks), HW_ID | XCC_ID << 32
#endif

__device__ __forceinline__ float apply_act(float x, int act, float slope) {
    if (act == ADVH_ACT_GELU) return gelu_fast(x);
    if (act == ADVH_ACT_LEAKY) return x > 0.f ? x : slope * x;
    return x;
}

// m -> (b, h, w) with m = (b*Hg + h)*Wg + w, without integer division: double-precision reciprocal
// multiply + one correction step (exact for m, d < 2^31).
struct RowDecomp {
    double iw, ih;
    unsigned Wg, Hg;
    __device__ __forceinline__ RowDecomp(unsigned Wg_, unsigned Hg_) : iw(1.0 / Wg_), ih(1.0 / Hg_), Wg(Wg_), Hg(Hg_) {}
    __device__ __forceinline__ static unsigned divq(unsigned n, unsigned d, double inv) {
        unsigned q = (unsigned)((double)n * inv);
        long r = (long)n - (long)q * d;
        if (r < 0) --q; else if (r >= (long)d) ++q;
        return q;
    }
    __device__ __forceinline__ void operator()(unsigned m, unsigned& b, unsigned& h, unsigned& w) const {
        unsigned t = divq(m, Wg, iw);
        w = m - t * Wg;
        if (Hg == 1) { h = 0; b = t; }
        else { b = divq(t, Hg, ih); h = t - b * Hg; }
    }
};

// Epilogue shared by the GEMM kernels: lane (fr = lane & 15, fq = lane >> 4) holds, per (ni, mi), the 4 values of
// MFMA rows 16 ni + 4 fq .. + 3 of the weight operand for row m.  bias + activation + residual + fp16 / fp32 stores.
// Narrow form (p.wide == 0): weight row R is output channel R, so these are channels n .. n+3 (8-byte fp16 stores).
// Wide form (p.wide == 1): the host packed the weight rows permuted inside every 32-row block (R -> channel
// 32 (R>>5) + 8 ((R>>2)&3) + 4 ((R>>4)&1) + (R&3)), so the tile pair (2q, 2q+1) of a lane is 8 CONSECUTIVE channels:
// 16-byte fp16 / 32-byte fp32 accesses, 64 / 128 contiguous bytes per row and instruction.
// value path of one VW-wide group: bias, pre-activation copy, activation, activation-derivative factor
// fp16-side accesses of the epilogue: plain fp16, or (SPLIT) the hi / lo plane pair of device_math.h's split format
// (lo plane p.o_lo elements behind the hi plane, same addressing).
template <int VW, bool SPLIT>
__device__ __forceinline__ void store_h(void* base, long o, long o_lo, const float (&v)[VW]) {
    typedef _Float16 hvec __attribute__((ext_vector_type(VW)));
    if constexpr (SPLIT) {
        hvec hv, lv;
        split_f32_vec<VW>(v, hv, lv);
        *(hvec*)((_Float16*)base + o) = hv;
        *(hvec*)((_Float16*)base + o + o_lo) = lv;
    } else {
        hvec hv;
#pragma unroll
        for (int r = 0; r < VW; ++r) hv[r] = (_Float16)v[r];
        *(hvec*)((_Float16*)base + o) = hv;
    }
}
template <int VW, bool SPLIT>
__device__ __forceinline__ void load_h(const void* base, long o, long o_lo, float (&v)[VW]) {
    typedef _Float16 hvec __attribute__((ext_vector_type(VW)));
    const hvec hv = *(const hvec*)((const _Float16*)base + o);
    if constexpr (SPLIT) {
        const hvec lv = *(const hvec*)((const _Float16*)base + o + o_lo);
#pragma unroll
        for (int r = 0; r < VW; ++r) v[r] = join_f32(hv[r], lv[r]);
    } else {
#pragma unroll
        for (int r = 0; r < VW; ++r) v[r] = (float)hv[r];
    }
}

template <int VW, bool SPLIT = false>
__device__ __forceinline__ void epilogue_value(const advh_gemm_desc& p, float (&v)[VW], const float* bias, int n, long o) {
    if (bias) {
#pragma unroll
        for (int c = 0; c < VW; c += 4) { float4 bb = *(const float4*)(bias + n + c); v[c] += bb.x; v[c + 1] += bb.y; v[c + 2] += bb.z; v[c + 3] += bb.w; }
    }
    if (p.out_pre) store_h<VW, SPLIT>(p.out_pre, o, p.o_lo, v);
#pragma unroll
    for (int r = 0; r < VW; ++r) v[r] = apply_act(v[r], p.act, p.slope);
    if (p.dact_src) {
        float zz[VW];
        load_h<VW, SPLIT>(p.dact_src, o, p.o_lo, zz);
#pragma unroll
        for (int r = 0; r < VW; ++r) v[r] *= gelu_grad(zz[r]);
    }
}

template <int VW, bool SPLIT = false>
__device__ __forceinline__ void epilogue_resid(const advh_gemm_desc& p, float (&v)[VW], long o) {
    if (p.resid) {
        if (p.resid_f32) {
#pragma unroll
            for (int c = 0; c < VW; c += 4) { float4 rr = *(const float4*)((const float*)p.resid + o + c); v[c] += rr.x; v[c + 1] += rr.y; v[c + 2] += rr.z; v[c + 3] += rr.w; }
        } else {
            float rr[VW];
            load_h<VW, SPLIT>(p.resid, o, p.o_lo, rr);
#pragma unroll
            for (int r = 0; r < VW; ++r) v[r] += rr[r];
        }
    }
}

template <int VW, bool SPLIT = false>
__device__ __forceinline__ void epilogue_write(const advh_gemm_desc& p, float (&v)[VW], long o) {
    if (p.out_f) {
#pragma unroll
        for (int c = 0; c < VW; c += 4) *(float4*)((float*)p.out_f + o + c) = make_float4(v[c], v[c + 1], v[c + 2], v[c + 3]);
    }
    if (p.out_h) store_h<VW, SPLIT>(p.out_h, o, p.o_lo, v);
    if (p.out_h2) {
        float w[VW];
#pragma unroll
        for (int r = 0; r < VW; ++r) w[r] = v[r] > 0.f ? v[r] : p.slope2 * v[r];
        store_h<VW, SPLIT>(p.out_h2, o, p.o_lo, w);
    }
}

template <int VW, bool SPLIT = false>
__device__ __forceinline__ void epilogue_store(const advh_gemm_desc& p, float (&v)[VW], bool ok, const float* bias, int n, long o) {
    if (ok) {
        epilogue_value<VW, SPLIT>(p, v, bias, n, o);
        epilogue_resid<VW, SPLIT>(p, v, o);
    } else {
#pragma unroll
        for (int r = 0; r < VW; ++r) v[r] = 0.f;
    }
    epilogue_write<VW, SPLIT>(p, v, o);
}

// Lean form for desc.plain_out (every row valid, output row m at o_c0 + m * o_sW, one column block): no row
// decomposition, no runtime divisions -- the epilogue of the Linear layers, where K = 768 makes it 10-30 % of a tile.
template <int MI, int NI, bool SPLIT = false>
__device__ __forceinline__ void gemm_epilogue_lean(const advh_gemm_desc& p, f32x4 (&acc)[NI][MI], int mw0, int nw0, int fr, int fq, int z,
                                                   long zo) {
    const float* bias = p.bias ? p.bias + (long)p.bias_sZ * z : nullptr;
    if (p.wide) {
        // The residual of a row (all its column pairs) is loaded before anything of that row is stored: with
        // out == resid (h += ...) the compiler must keep every load behind the previous store, which made the epilogue a
        // chain of eight exposed memory latencies per lane; now it is four (two rows at a time would cost a workgroup
        // per CU: 156 VGPRs).
        const bool pre = p.resid != nullptr;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = mw0 + mi * 16 + fr;
            if (m >= p.M) continue;
            const long orow = (long)m * p.o_sW + p.o_c0 + zo;
            float rv[NI / 2][8];
            if (pre) {
#pragma unroll
                for (int q = 0; q < NI / 2; ++q) {
                    const int n = nw0 + q * 32 + fq * 8;
                    const long o = orow + (n < p.N ? n : 0);
                    if (p.resid_f32) {
                        const float4 r0 = *(const float4*)((const float*)p.resid + o), r1 = *(const float4*)((const float*)p.resid + o + 4);
                        rv[q][0] = r0.x; rv[q][1] = r0.y; rv[q][2] = r0.z; rv[q][3] = r0.w;
                        rv[q][4] = r1.x; rv[q][5] = r1.y; rv[q][6] = r1.z; rv[q][7] = r1.w;
                    } else {
                        load_h<8, SPLIT>(p.resid, o, p.o_lo, rv[q]);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < NI / 2; ++q) {
                const int n = nw0 + q * 32 + fq * 8;
                if (n >= p.N) continue;
                float v[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[r] = acc[2 * q][mi][r]; v[4 + r] = acc[2 * q + 1][mi][r]; }
                epilogue_value<8, SPLIT>(p, v, bias, n, orow + n);
                if (pre) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] += rv[q][r];
                }
                epilogue_write<8, SPLIT>(p, v, orow + n);
            }
        }
        return;
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = mw0 + mi * 16 + fr;
        if (m >= p.M) continue;
        const long orow = (long)m * p.o_sW + p.o_c0 + zo;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = nw0 + ni * 16 + fq * 4;
            if (n >= p.N) continue;
            float v[4] = {acc[ni][mi][0], acc[ni][mi][1], acc[ni][mi][2], acc[ni][mi][3]};
            epilogue_store<4, SPLIT>(p, v, true, bias, n, orow + n);
        }
    }
}

// bias of the NQ groups of 8 consecutive channels n, n + 32, ... a lane owns in the wide layout (zeros without a bias)
template <int NQ>
__device__ __forceinline__ void load_bias8(const advh_gemm_desc& p, int z, int n0, float (&bb)[NQ][8]) {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int k = 0; k < 8; ++k) bb[q][k] = 0.f;
    if (!p.bias) return;
    const float* bias = p.bias + (long)p.bias_sZ * z;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int n = n0 + q * 32;
        const float4 b0 = *(const float4*)(bias + (n < p.N ? n : 0)), b1 = *(const float4*)(bias + (n < p.N ? n : 0) + 4);
        bb[q][0] = b0.x; bb[q][1] = b0.y; bb[q][2] = b0.z; bb[q][3] = b0.w;
        bb[q][4] = b1.x; bb[q][5] = b1.y; bb[q][6] = b1.z; bb[q][7] = b1.w;
    }
}

// Tight form of the lean epilogue: the three shapes the Linear layers of the embedder use, each compiled as its own
// straight-line code (MODE 0: bias -> fp16-side output; 1: bias + GELU -> fp16-side output; 2: bias + fp32 residual
// -> fp32 output, the residual stream).  The general forms below test every descriptor option per value group, which
// unrolls to ~30 000 instructions (~200 KiB) per kernel: skipping through that text costs an instruction-cache miss per
// taken branch -- measured 16 500 cycles per workgroup (a quarter of a K = 768 tile) against ~3 000 here.
// PRE = rows whose residual is loaded before the first store (out == resid, so a load cannot pass an earlier store).
template <int MI, int NI, bool SPLIT, int MODE, int PRE>
__device__ __forceinline__ void gemm_epilogue_tight(const advh_gemm_desc& p, f32x4 (&acc)[NI][MI], int mw0, int nw0, int fr, int fq, int z,
                                                    long zo) {
    constexpr int NQ = NI / 2;
    static_assert(MI % PRE == 0, "residual rows per group");
    float bb[NQ][8];
    load_bias8<NQ>(p, z, nw0 + fq * 8, bb);
#pragma unroll
    for (int g = 0; g < MI; g += PRE) {
        float rv[MODE == 2 ? PRE : 1][NQ][8];
        if constexpr (MODE == 2) {
#pragma unroll
            for (int i = 0; i < PRE; ++i) {
                const int m = min(mw0 + (g + i) * 16 + fr, p.M - 1);
                const long orow = (long)m * p.o_sW + p.o_c0 + zo;
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int n = nw0 + q * 32 + fq * 8;
                    const float* r = (const float*)p.resid + orow + (n < p.N ? n : 0);
                    const float4 r0 = *(const float4*)r, r1 = *(const float4*)(r + 4);
                    rv[i][q][0] = r0.x; rv[i][q][1] = r0.y; rv[i][q][2] = r0.z; rv[i][q][3] = r0.w;
                    rv[i][q][4] = r1.x; rv[i][q][5] = r1.y; rv[i][q][6] = r1.z; rv[i][q][7] = r1.w;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < PRE; ++i) {
            const int mi = g + i, m = mw0 + mi * 16 + fr;
            if (m >= p.M) continue;
            const long orow = (long)m * p.o_sW + p.o_c0 + zo;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int n = nw0 + q * 32 + fq * 8;
                if (n >= p.N) continue;
                float v[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[r] = acc[2 * q][mi][r] + bb[q][r]; v[4 + r] = acc[2 * q + 1][mi][r] + bb[q][4 + r]; }
                if constexpr (MODE == 1) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] = gelu_fast(v[r]);
                }
                if constexpr (MODE == 2) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] += rv[i][q][r];
                    float* o = (float*)p.out_f + orow + n;
                    *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
                    *(float4*)(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
                } else {
                    store_h<8, SPLIT>(p.out_h, orow + n, p.o_lo, v);
                }
            }
        }
    }
}

// Staged form of tight modes 0 / 1 for a 64-column wavefront tile: the wavefront writes its (16 MI) x 64 tile (one
// plane, or hi + lo planes) into the LDS the K loop no longer needs -- row-major, 16-byte chunks XOR-swizzled by the
// row -- and reads it back with 8 lanes per row, so that a store instruction covers 8 rows x one full 128-byte line
// instead of 16 rows x 64 bytes.  Wavefront-local (LDS operations of a wavefront complete in order): no barrier.
// ACT: ADVH_ACT_*; ROWS: output row m lives at the (b, h, w) address of the general form (convolutions: window test,
// halo_zero) instead of o_c0 + m * o_sW.
template <int MI, int NI, bool SPLIT, int ACT, bool ROWS = false>
__device__ __forceinline__ void gemm_epilogue_staged(const advh_gemm_desc& p, f32x4 (&acc)[NI][MI], char* stage, int mw0, int nw0, int lane,
                                                     int z, long zo) {
    static_assert(NI == 4, "64-column wavefront tile, fp16-side output");
    constexpr int PL = MI * 16 * 128;                          // bytes of one plane of the wavefront's tile
    const int fr = lane & 15, fq = lane >> 4;
    float bb[2][8];
    load_bias8<2>(p, z, nw0 + fq * 8, bb);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int r = mi * 16 + fr;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[k] = acc[2 * q][mi][k] + bb[q][k]; v[4 + k] = acc[2 * q + 1][mi][k] + bb[q][4 + k]; }
            if constexpr (ACT == ADVH_ACT_GELU) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = gelu_fast(v[k]);
            }
            if constexpr (ACT == ADVH_ACT_LEAKY) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = v[k] > 0.f ? v[k] : p.slope * v[k];
            }
            char* dst = stage + r * 128 + (((q * 4 + fq) ^ (r & 7)) << 4);
            f16x8 hv, lv;
            if constexpr (SPLIT) split_f32_vec<8>(v, hv, lv);
            else {
#pragma unroll
                for (int k = 0; k < 8; ++k) hv[k] = (_Float16)v[k];
            }
            *(f16x8*)dst = hv;
            if constexpr (SPLIT) *(f16x8*)(dst + PL) = lv;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int rr = lane >> 3, c = lane & 7, n = nw0 + c * 8;
    _Float16* out = (_Float16*)p.out_h;
    unsigned Wg = p.Wg, Hg = p.Hg;
    asm volatile("" : "+s"(Wg), "+s"(Hg));                 // the reciprocals are computed here, not hoisted above the K loop
    const RowDecomp rd(Wg, Hg);
#pragma unroll
    for (int it = 0; it < MI * 2; ++it) {
        const int row = it * 8 + rr, m = mw0 + row;
        const char* src = stage + row * 128 + ((c ^ (row & 7)) << 4);
        f16x8 hv = *(const f16x8*)src;
        f16x8 lv;
        if constexpr (SPLIT) lv = *(const f16x8*)(src + PL);
        if (m < p.M && n < p.N) {
            long o;
            if constexpr (ROWS) {
                unsigned w, h, b;
                rd((unsigned)m, b, h, w);
                const bool ok = (int)h >= p.h0 && (int)h < p.h1 && (int)w >= p.w0 && (int)w < p.w1;
                if (!ok && !p.halo_zero) continue;
                if (!ok) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) { hv[k] = (_Float16)0.f; lv[k] = (_Float16)0.f; }
                }
                o = (long)b * p.o_sB + (long)h * p.o_sH + (long)w * p.o_sW + p.o_c0 + zo + n;
            } else {
                o = (long)m * p.o_sW + p.o_c0 + zo + n;
            }
            *(f16x8*)(out + o) = hv;
            if constexpr (SPLIT) *(f16x8*)(out + o + p.o_lo) = lv;
        }
        if constexpr (ROWS && !SPLIT) {
            if (it & 1) __builtin_amdgcn_sched_barrier(0);     // keep two rows' address arithmetic in flight, not eight: the fp16 kernels' 4 wavefronts per SIMD need <= 128 VGPRs
        }
    }
}

// Tight form of the general (row-decomposing) epilogue for what every convolution of the forward path asks for: wide
// packing, one column block, bias (or none) + activation -> fp16-side output; window test and halo_zero as in the general form.
template <int MI, int NI, bool SPLIT, int ACT>
__device__ __forceinline__ void gemm_epilogue_rows_tight(const advh_gemm_desc& p, f32x4 (&acc)[NI][MI], int mw0, int nw0, int fr, int fq, int z,
                                                         long zo) {
    constexpr int NQ = NI / 2;
    float bb[NQ][8];
    load_bias8<NQ>(p, z, nw0 + fq * 8, bb);
    const RowDecomp rd(p.Wg, p.Hg);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const unsigned m = mw0 + mi * 16 + fr;
        if (m >= (unsigned)p.M) continue;
        unsigned w, h, b;
        rd(m, b, h, w);
        const bool ok = (int)h >= p.h0 && (int)h < p.h1 && (int)w >= p.w0 && (int)w < p.w1;
        if (!ok && !p.halo_zero) continue;
        const long orow = (long)b * p.o_sB + (long)h * p.o_sH + (long)w * p.o_sW + p.o_c0 + zo;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int n = nw0 + q * 32 + fq * 8;
            if (n >= p.N) continue;
            float v[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[k] = acc[2 * q][mi][k] + bb[q][k]; v[4 + k] = acc[2 * q + 1][mi][k] + bb[q][4 + k]; }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if constexpr (ACT == ADVH_ACT_GELU) v[k] = gelu_fast(v[k]);
                if constexpr (ACT == ADVH_ACT_LEAKY) v[k] = v[k] > 0.f ? v[k] : p.slope * v[k];
                if (!ok) v[k] = 0.f;
            }
            store_h<8, SPLIT>(p.out_h, orow + n, p.o_lo, v);
        }
    }
}

// Staged form of tight mode 2 (fp32 residual stream, out_f = resid + acc + bias): the residual is prefetched in the store
// layout (16 lanes x float4 per row), the tile goes through LDS as fp32 (16-byte chunks XOR-swizzled by the row), and every
// load / store instruction covers 4 rows x 256 contiguous bytes instead of 16 rows x 128.  PASSES: the wavefront's stage holds
// 16 MI / PASSES rows of fp32 (fp32-class kernels: 16 KiB = the whole 64 x 64 tile, one pass; fp16 kernels: four passes of 16 rows, which keeps the prefetched residual at 16 VGPRs beside the 64 accumulators).
template <int MI, int NI, int PASSES>
__device__ __forceinline__ void gemm_epilogue_staged_f32(const advh_gemm_desc& p, f32x4 (&acc)[NI][MI], char* stage, int mw0, int nw0, int lane,
                                                         int z, long zo) {
    static_assert(NI == 4 && MI % PASSES == 0, "64-column wavefront tile");
    constexpr int MP = MI / PASSES;                        // 16-row blocks per pass
    const int fr = lane & 15, fq = lane >> 4;
    const int rr = lane >> 4, c = lane & 15, n = nw0 + c * 4;
    float bb[2][8];
    if constexpr (PASSES == 1) load_bias8<2>(p, z, nw0 + fq * 8, bb);
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        float4 rv[MP * 4];
#pragma unroll
        for (int it = 0; it < MP * 4; ++it) {
            const int m = min(mw0 + ps * MP * 16 + it * 4 + rr, p.M - 1);
            rv[it] = *(const float4*)((const float*)p.resid + (long)m * p.o_sW + p.o_c0 + zo + (n < p.N ? n : 0));
        }
        if constexpr (PASSES > 1) load_bias8<2>(p, z, nw0 + fq * 8, bb);   // per pass (an L1 hit behind the residual loads): 16 fewer live VGPRs
#pragma unroll
        for (int mb = 0; mb < MP; ++mb) {
            const int mi = ps * MP + mb, r = mb * 16 + fr;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int ch = 2 * (q * 4 + fq);
                *(float4*)(stage + r * 256 + ((ch ^ (r & 15)) << 4)) =
                    make_float4(acc[2 * q][mi][0] + bb[q][0], acc[2 * q][mi][1] + bb[q][1], acc[2 * q][mi][2] + bb[q][2], acc[2 * q][mi][3] + bb[q][3]);
                *(float4*)(stage + r * 256 + (((ch + 1) ^ (r & 15)) << 4)) =
                    make_float4(acc[2 * q + 1][mi][0] + bb[q][4], acc[2 * q + 1][mi][1] + bb[q][5], acc[2 * q + 1][mi][2] + bb[q][6], acc[2 * q + 1][mi][3] + bb[q][7]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < MP * 4; ++it) {
            const int row = it * 4 + rr, m = mw0 + ps * MP * 16 + row;
            const float4 v = *(const float4*)(stage + row * 256 + ((c ^ (row & 15)) << 4));
            if (m < p.M && n < p.N)
                *(float4*)((float*)p.out_f + (long)m * p.o_sW + p.o_c0 + zo + n) = make_float4(v.x + rv[it].x, v.y + rv[it].y, v.z + rv[it].z, v.w + rv[it].w);
        }
        if constexpr (PASSES > 1) {                        // the next pass overwrites the stage: its reads must have completed
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// which tight form serves this descriptor (-1: none); uniform, evaluated once per workgroup
__device__ __forceinline__ int tight_mode(const advh_gemm_desc& p) {
    if (!p.wide || p.out_pre || p.dact_src || p.out_h2) return -1;
    if (p.out_h && !p.out_f && !p.resid) return p.act == ADVH_ACT_NONE ? 0 : p.act == ADVH_ACT_GELU ? 1 : -1;
    if (p.out_f && !p.out_h && p.resid && p.resid_f32 && p.act == ADVH_ACT_NONE) return 2;
    return -1;
}

// stage: this wavefront's 16 MI x 64 x 2 (x 2 planes) bytes of LDS, free once the K loop's last barrier has passed
// (nullptr: the tile shape has no staged form)
template <int MI, int NI, bool SPLIT, int PRE>
__device__ __forceinline__ void gemm_epilogue_plain(const advh_gemm_desc& p, f32x4 (&acc)[NI][MI], char* stage, int mw0, int nw0, int fr, int fq,
                                                    int z, long zo) {
    const int mode = tight_mode(p);
    if constexpr (NI == 4) {
        if (stage && mode == 0) return gemm_epilogue_staged<MI, NI, SPLIT, 0>(p, acc, stage, mw0, nw0, fr + 16 * fq, z, zo);
        if (stage && mode == 1) return gemm_epilogue_staged<MI, NI, SPLIT, 1>(p, acc, stage, mw0, nw0, fr + 16 * fq, z, zo);
        // the fp32-class kernels' stage holds the wavefront's whole tile as fp32, the fp16 kernels stage it in four passes
        if (stage && mode == 2) return gemm_epilogue_staged_f32<MI, NI, SPLIT ? 1 : 4>(p, acc, stage, mw0, nw0, fr + 16 * fq, z, zo);
    }
    if (mode == 0) gemm_epilogue_tight<MI, NI, SPLIT, 0, PRE>(p, acc, mw0, nw0, fr, fq, z, zo);
    else if (mode == 1) gemm_epilogue_tight<MI, NI, SPLIT, 1, PRE>(p, acc, mw0, nw0, fr, fq, z, zo);
    else if (mode == 2) gemm_epilogue_tight<MI, NI, SPLIT, 2, PRE>(p, acc, mw0, nw0, fr, fq, z, zo);
    else gemm_epilogue_lean<MI, NI, SPLIT>(p, acc, mw0, nw0, fr, fq, z, zo);
}

template <int MI, int NI, bool SPLIT = false>
__device__ __forceinline__ void gemm_epilogue(const advh_gemm_desc& p, f32x4 (&acc)[NI][MI], int mw0, int nw0, int fr, int fq, int z,
                                              long zo = -1);            // zo: output offset of batch z (default o_sZ * z)

// Staged form for the second convolution of a HiFi-GAN ResBlock step (hifigan.py:180 -> Kong et al. ResBlock1: x = x + conv2(...)):
// bias, no activation, optional fp16-side residual (same layout as the output), fp16-side output and optionally its leaky copy
// (out_h2, the next convolution's input).  The tile goes through LDS as fp32 -- the residual is added to the unrounded value,
// exactly as in the generic form -- in PASSES passes of 16 MI / PASSES rows (fp32-class kernels: one pass; fp16 kernels: four, which keeps them at 128 VGPRs);
// the residual of a pass is prefetched in the store layout before its first store (out_h may be resid), 16 lanes x 8 bytes = one
// full 128-byte line per row and plane, 4 rows per instruction.
template <int MI, int NI, bool SPLIT, int PASSES>
__device__ __forceinline__ void gemm_epilogue_rows_staged_resid(const advh_gemm_desc& p, f32x4 (&acc)[NI][MI], char* stage, int mw0, int nw0,
                                                                int lane, int z, long zo) {
    static_assert(NI == 4 && MI % PASSES == 0, "64-column wavefront tile");
    constexpr int MP = MI / PASSES;
    typedef _Float16 f16x4v __attribute__((ext_vector_type(4)));
    const int fr = lane & 15, fq = lane >> 4;
    const int rr = lane >> 4, c = lane & 15, n = nw0 + c * 4;
    unsigned Wg = p.Wg, Hg = p.Hg;
    asm volatile("" : "+s"(Wg), "+s"(Hg));
    const RowDecomp rd(Wg, Hg);
    const bool has_r = p.resid != nullptr;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        long ofs[MP * 4];                                   // output element offset of this lane's 4 channels per row; -1: row not written
        bool okr[MP * 4];
        f16x4v rh[MP * 4], rl[MP * 4];
#pragma unroll
        for (int it = 0; it < MP * 4; ++it) {
            const int m = mw0 + ps * MP * 16 + it * 4 + rr;
            ofs[it] = -1;
            okr[it] = false;
            rh[it] = f16x4v{(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
            rl[it] = rh[it];
            if (m < p.M && n < p.N) {
                unsigned w, h, b;
                rd((unsigned)m, b, h, w);
                const bool ok = (int)h >= p.h0 && (int)h < p.h1 && (int)w >= p.w0 && (int)w < p.w1;
                if (ok || p.halo_zero) {
                    ofs[it] = (long)b * p.o_sB + (long)h * p.o_sH + (long)w * p.o_sW + p.o_c0 + zo + n;
                    okr[it] = ok;
                    if (ok && has_r) {
                        rh[it] = *(const f16x4v*)((const _Float16*)p.resid + ofs[it]);
                        if constexpr (SPLIT) rl[it] = *(const f16x4v*)((const _Float16*)p.resid + ofs[it] + p.o_lo);
                    }
                }
            }
        }
        float bb[2][8];
        load_bias8<2>(p, z, nw0 + fq * 8, bb);
#pragma unroll
        for (int mb = 0; mb < MP; ++mb) {
            const int mi = ps * MP + mb, r = mb * 16 + fr;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int ch = 2 * (q * 4 + fq);
                *(float4*)(stage + r * 256 + ((ch ^ (r & 15)) << 4)) =
                    make_float4(acc[2 * q][mi][0] + bb[q][0], acc[2 * q][mi][1] + bb[q][1], acc[2 * q][mi][2] + bb[q][2], acc[2 * q][mi][3] + bb[q][3]);
                *(float4*)(stage + r * 256 + (((ch + 1) ^ (r & 15)) << 4)) =
                    make_float4(acc[2 * q + 1][mi][0] + bb[q][4], acc[2 * q + 1][mi][1] + bb[q][5], acc[2 * q + 1][mi][2] + bb[q][6], acc[2 * q + 1][mi][3] + bb[q][7]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < MP * 4; ++it) {
            const int row = it * 4 + rr;
            const float4 t = *(const float4*)(stage + row * 256 + ((c ^ (row & 15)) << 4));
            if (ofs[it] < 0) continue;
            float v[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!okr[it]) v[k] = 0.f;
                else if (has_r) v[k] += SPLIT ? join_f32(rh[it][k], rl[it][k]) : (float)rh[it][k];
            }
            store_h<4, SPLIT>(p.out_h, ofs[it], p.o_lo, v);
            if (p.out_h2) {
                float w2[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) w2[k] = v[k] > 0.f ? v[k] : p.slope2 * v[k];
                store_h<4, SPLIT>(p.out_h2, ofs[it], p.o_lo, w2);
            }
        }
        if constexpr (PASSES > 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// the row-decomposing epilogue of the two single-buffered kernels: staged / tight form where the descriptor allows, else general
template <int MI, int NI, bool SPLIT>
__device__ __forceinline__ void gemm_epilogue_rows(const advh_gemm_desc& p, f32x4 (&acc)[NI][MI], char* stage, int mw0, int nw0, int fr, int fq,
                                                   int z, long zo) {
    const bool simple = p.wide && p.n_div >= p.N && p.ph_r <= 0 && p.out_h && !p.out_f && !p.resid && !p.out_pre && !p.dact_src && !p.out_h2;
    if (simple) {
        if constexpr (NI == 4) {
            if (stage) {
                if (p.act == ADVH_ACT_LEAKY) return gemm_epilogue_staged<MI, NI, SPLIT, ADVH_ACT_LEAKY, true>(p, acc, stage, mw0, nw0, fr + 16 * fq, z, zo);
                if (p.act == ADVH_ACT_GELU) return gemm_epilogue_staged<MI, NI, SPLIT, ADVH_ACT_GELU, true>(p, acc, stage, mw0, nw0, fr + 16 * fq, z, zo);
                return gemm_epilogue_staged<MI, NI, SPLIT, ADVH_ACT_NONE, true>(p, acc, stage, mw0, nw0, fr + 16 * fq, z, zo);
            }
        }
        if (p.act == ADVH_ACT_LEAKY) return gemm_epilogue_rows_tight<MI, NI, SPLIT, ADVH_ACT_LEAKY>(p, acc, mw0, nw0, fr, fq, z, zo);
        if (p.act == ADVH_ACT_GELU) return gemm_epilogue_rows_tight<MI, NI, SPLIT, ADVH_ACT_GELU>(p, acc, mw0, nw0, fr, fq, z, zo);
        return gemm_epilogue_rows_tight<MI, NI, SPLIT, ADVH_ACT_NONE>(p, acc, mw0, nw0, fr, fq, z, zo);
    }
    if constexpr (NI == 4) {
        const bool resid_form = stage && p.wide && p.n_div >= p.N && p.ph_r <= 0 && p.out_h && !p.out_f && !p.out_pre && !p.dact_src &&
                                p.act == ADVH_ACT_NONE && (p.out_h2 || (p.resid && !p.resid_f32)) && !(p.resid && p.resid_f32);
        if (resid_form) return gemm_epilogue_rows_staged_resid<MI, NI, SPLIT, SPLIT ? 1 : 4>(p, acc, stage, mw0, nw0, fr + 16 * fq, z, zo);
    }
    gemm_epilogue<MI, NI, SPLIT>(p, acc, mw0, nw0, fr, fq, z, zo);
}

template <int MI, int NI, bool SPLIT>
__device__ __forceinline__ void gemm_epilogue(const advh_gemm_desc& p, f32x4 (&acc)[NI][MI], int mw0, int nw0, int fr, int fq, int z, long zo) {
    static_assert(NI % 2 == 0, "the wide epilogue pairs n-tiles");
    const float* bias = p.bias ? p.bias + (long)p.bias_sZ * z : nullptr;
    const RowDecomp rd(p.Wg, p.Hg);
    const bool oneblk = p.n_div >= p.N && p.ph_r <= 0;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        unsigned m = mw0 + mi * 16 + fr;
        if (m >= (unsigned)p.M) continue;
        unsigned w, h, b;
        rd(m, b, h, w);
        bool ok = (int)h >= p.h0 && (int)h < p.h1 && (int)w >= p.w0 && (int)w < p.w1;
        if (!ok && !p.halo_zero) continue;
        long orow = (long)b * p.o_sB + (long)h * p.o_sH + (long)w * p.o_sW + p.o_c0 + (zo >= 0 ? zo : p.o_sZ * z);
        if (p.wide) {
#pragma unroll
            for (int q = 0; q < NI / 2; ++q) {
                int n = nw0 + q * 32 + fq * 8;
                if (n >= p.N) continue;
                if (p.ph_r > 0) {
                    int to = (int)w * p.ph_r + n / p.n_div - p.ph_pad;
                    if (to < 0 || to >= p.ph_T) continue;
                }
                long o;
                if (oneblk) o = orow + n;                    // n_div >= N: one column block, no runtime divisions
                else {
                    const int qn = n / p.n_div;
                    o = orow + (p.n_sub > 1 ? (long)(qn / p.n_sub) * p.o_sNhh + (long)(qn % p.n_sub) * p.o_sNhi : (long)qn * p.o_sNhi) + (n % p.n_div);
                }
                float v[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[r] = acc[2 * q][mi][r]; v[4 + r] = acc[2 * q + 1][mi][r]; }
                epilogue_store<8, SPLIT>(p, v, ok, bias, n, o);
            }
        } else {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                int n = nw0 + ni * 16 + fq * 4;
                if (n >= p.N) continue;
                if (p.ph_r > 0) {                              // transposed-conv phase window (column-dependent validity)
                    int to = (int)w * p.ph_r + n / p.n_div - p.ph_pad;
                    if (to < 0 || to >= p.ph_T) continue;
                }
                long o;
                if (oneblk) o = orow + n;
                else {
                    const int qn = n / p.n_div;
                    o = orow + (p.n_sub > 1 ? (long)(qn / p.n_sub) * p.o_sNhh + (long)(qn % p.n_sub) * p.o_sNhi : (long)qn * p.o_sNhi) + (n % p.n_div);
                }
                float v[4] = {acc[ni][mi][0], acc[ni][mi][1], acc[ni][mi][2], acc[ni][mi][3]};
                epilogue_store<4, SPLIT>(p, v, ok, bias, n, o);
            }
        }
    }
}

// WPE = wavefronts per SIMD the register allocation must allow (amdgpu-waves-per-eu through __launch_bounds__):
// the single-buffered loop below hides the global->LDS latency with OTHER workgroups of the CU, so occupancy is
// the lever (4 wavefronts per SIMD = 126 VGPRs for the 64 x 64 wave tile, no spills).
// PLAIN = the operand rows are affine in the row index (desc.plain: one source, identity K table, row m at a_c0 + m * a_sW):
// all Linear layers and the feature-encoder Conv1d layers.  The loader then needs no per-row offsets, no K-table load per
// K-step and no 64-bit address arithmetic per DMA (one pointer + scalar steps), and the A fragments are read one at a
// time: 108 instead of 126 VGPRs and 8-20 % faster on those shapes (tools/experiments/gemm_occ5.hip).
template <int BM, int BN, int WM, int WN, int WPE, bool PLAIN = false>
__global__ __launch_bounds__(64 * WM * WN, WPE) void gemm_f16_kernel(const advh_gemm_desc p) {
    constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 16, NI = TN / 16;
    constexpr int NT = 64 * WM * WN, RPP = NT / 8;  // threads; tile rows covered by one loader pass
    constexpr int NA = BM / RPP, NB = BN / RPP;     // 16-byte chunks per thread per K-step
    static_assert(NA >= 1 && NB >= 1 && BM % RPP == 0 && BN % RPP == 0, "loader passes");
    __shared__ __attribute__((aligned(16))) char smem[(BM + BN) * BK * 2];
    char* ldsA = smem;
    char* ldsB = smem + BM * BK * 2;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv / WN, wn = wv % WN;

    // XCD-aware tile order: workgroups that share an XCD (id % 8) walk neighbouring tiles
    const int tilesN = (p.N + BN - 1) / BN;
    int nwg = gridDim.x;
    int id = blockIdx.x;
    {
        const int q8 = nwg / 8, r8 = nwg % 8, xcd = id % 8;
        id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + id / 8;
    }
    int z = blockIdx.z;
    if (p.z_inner) {                                     // batches of one tile next to each other on one XCD
        z = id % p.nz;
        id /= p.nz;
        nwg /= p.nz;
    }
    // super-columns: walk all M tiles of `sc` N-tiles before moving on, so the weight slice in flight (sc*BN*K*2
    // bytes) stays L2-resident instead of cycling a > 4 MiB weight through every XCD's L2 (host picks sc; 0 = off)
    int tile_n, tile_m;
    {
        const int tilesM = nwg / tilesN, sc = (p.sc > 0 && p.sc < tilesN) ? p.sc : tilesN;
        const int s = id / (tilesM * sc), rem = id - s * tilesM * sc;
        const int wcols = min(sc, tilesN - s * sc);
        tile_m = rem / wcols;
        tile_n = s * sc + rem - tile_m * wcols;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int zh = p.nz_lo > 1 ? z / p.nz_lo : z, zw = p.nz_lo > 1 ? z % p.nz_lo : 0;

    const _Float16* A0 = (const _Float16*)p.A0 + (p.a_sZ[0] * zh + p.a_sZ2[0] * zw) * 8;
    const _Float16* A1 = (const _Float16*)p.A1 + (p.a_sZ[1] * zh + p.a_sZ2[1] * zw) * 8;
    const _Float16* Wp = (const _Float16*)p.W + p.w_sZ * z;

    // ---- loader setup: this thread's chunk column q and its NA rows' base offsets (chunk units)
    const int ldrow = tid >> 3;                          // + RPP*i
    const int q = (tid & 7) ^ (ldrow & 7);               // logical K-chunk this lane fetches (swizzled source)
    // row base of the "safe" row used by invalid rows (halo / M tail): first valid row of item 0
    unsigned rb0[PLAIN ? 1 : NA], rb1[PLAIN ? 1 : NA];
    const _Float16* wrow[PLAIN ? 1 : NB];
    const _Float16* ap[PLAIN ? NA : 1];                  // PLAIN: row pointers (rows past M re-read row M-1: never stored)
    const _Float16* wp0 = nullptr;
    long wstep = 0;
    if constexpr (PLAIN) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int m = min(m0 + ldrow + RPP * i, p.M - 1);
            ap[i] = A0 + ((long)m * p.a_sW[0] + p.a_c0[0] + q) * 8;
        }
        const long wld = p.w_ld ? p.w_ld : (long)p.Ktot;
        wp0 = Wp + (long)(n0 + ldrow) * wld + q * 8;
        wstep = (long)RPP * wld;
    } else {
        long safe0 = (long)p.h0 * p.a_sH[0] + (long)p.w0 * p.a_sW[0] + p.a_c0[0];
        long safe1 = (long)p.h0 * p.a_sH[1] + (long)p.w0 * p.a_sW[1] + p.a_c0[1];
        const RowDecomp rd(p.Wg, p.Hg);
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            unsigned m = m0 + ldrow + RPP * i;
            unsigned w, h, b;
            rd(m, b, h, w);
            bool ok = m < (unsigned)p.M && (int)h >= p.h0 && (int)h < p.h1 && (int)w >= p.w0 && (int)w < p.w1;
            long r0 = ok ? (long)b * p.a_sB[0] + (long)h * p.a_sH[0] + (long)w * p.a_sW[0] + p.a_c0[0] : safe0;
            long r1 = ok ? (long)b * p.a_sB[1] + (long)h * p.a_sH[1] + (long)w * p.a_sW[1] + p.a_c0[1] : safe1;
            rb0[i] = (unsigned)r0;
            rb1[i] = (unsigned)r1;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) wrow[i] = Wp + (long)(n0 + ldrow + RPP * i) * (p.w_ld ? p.w_ld : (long)p.Ktot) + q * 8;
    }

    // ---- fragment read offsets (bytes) inside a tile: row r, logical chunk c -> (r*8 + (c ^ (r&7)))*16
    const int fr = lane & 15, fq = lane >> 4;
    int offA[2], offB[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        int c = (kk * 4 + fq) ^ (fr & 7);
        offA[kk] = ((wm * TM + fr) * 8 + c) * 16;
        offB[kk] = ((wn * TN + fr) * 8 + c) * 16;
    }

    f32x4 acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.Ktot / BK;
    int kq = PLAIN ? 0 : p.ktab[q];
    for (int kt = 0; kt < nk; ++kt) {
        if constexpr (PLAIN) {
#pragma unroll
            for (int i = 0; i < NA; ++i)
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(ap[i] + kt * BK), LDS_PTR(ldsA + (wv * 64 + NT * i) * 16), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < NB; ++i)
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(wp0 + i * wstep + kt * BK), LDS_PTR(ldsB + (wv * 64 + NT * i) * 16), 16, 0, 0);
        } else {
            const bool s1 = kq < 0;
            const unsigned ko = (unsigned)kq & 0x7fffffffu;
            const _Float16* base = s1 ? A1 : A0;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const _Float16* g = base + ((unsigned long)((s1 ? rb1[i] : rb0[i]) + ko)) * 8;
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(g), LDS_PTR(ldsA + (wv * 64 + NT * i) * 16), 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < NB; ++i)
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(wrow[i] + kt * BK),
                                                 LDS_PTR(ldsB + (wv * 64 + NT * i) * 16), 16, 0, 0);
            if (kt + 1 < nk) kq = p.ktab[(kt + 1) * 8 + q];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if constexpr (PLAIN) {
                f16x8 b[NI];
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) b[ni] = *(const f16x8*)(ldsB + offB[kk] + ni * 16 * 128);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const f16x8 a = *(const f16x8*)(ldsA + offA[kk] + mi * 16 * 128);
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[ni], a, acc[ni][mi], 0, 0, 0);
                }
            } else {
                f16x8 a[MI], b[NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) a[mi] = *(const f16x8*)(ldsA + offA[kk] + mi * 16 * 128);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) b[ni] = *(const f16x8*)(ldsB + offB[kk] + ni * 16 * 128);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[ni], a[mi], acc[ni][mi], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    char* stage = (NI == 4 && MI * 16 * 128 * WM * WN <= (BM + BN) * BK * 2) ? smem + wv * (MI * 16 * 128) : nullptr;
    if (PLAIN && p.plain_out) gemm_epilogue_plain<MI, NI, false, 1>(p, acc, stage, m0 + wm * TM, n0 + wn * TN, fr, fq, z, p.o_sZ * zh + p.o_sZ2 * zw);
    else gemm_epilogue_rows<MI, NI, false>(p, acc, stage, m0 + wm * TM, n0 + wn * TN, fr, fq, z, p.o_sZ * zh + p.o_sZ2 * zw);
}

template <int BM, int BN, int WM, int WN, int WPE>
static int launch(const advh_gemm_desc& d, hipStream_t s) {
    const int tilesM = (d.M + BM - 1) / BM, tilesN = (d.N + BN - 1) / BN;
    if (tilesN * BN > d.w_rows) return ADVH_EINVAL;
    const int nz = d.nz > 0 ? d.nz : 1;
    if (d.z_inner && (long)tilesM * tilesN * nz > 0x7fffffffL) return ADVH_EINVAL;
    dim3 grid(d.z_inner ? tilesM * tilesN * nz : tilesM * tilesN, 1, d.z_inner ? 1 : nz);
    if constexpr (BM == 128 && BN == 128 && WPE == 4) {
        if (d.plain) {
            hipLaunchKernelGGL((gemm_f16_kernel<BM, BN, WM, WN, WPE, true>), grid, dim3(64 * WM * WN), 0, s, d);
            return ADVH_LAUNCH_CHECK();
        }
    }
    hipLaunchKernelGGL((gemm_f16_kernel<BM, BN, WM, WN, WPE>), grid, dim3(64 * WM * WN), 0, s, d);
    return ADVH_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------------
// fp32-class instance ("x3"): both operands arrive in the split format of device_math.h (hi plane + lo plane, the lo
// plane a fixed distance behind: desc.a_lo / w_lo), the four planes of a K-step are staged by the same LDS DMA with the
// same swizzle, and every (weight fragment, activation fragment) pair costs three MFMAs:
//     acc  += Wh * Ah                       (leading term)
//     accx += Wh * Al + Wl * Ah             (cross terms, carry 2^-11; the Wl * Al term, <= 2^-22 relative, is dropped)
// result = acc + accx * 2^-11: every product is exact in fp32, so the only rounding is the fp32 accumulation -- the
// arithmetic class of the reference's fp32 convolutions / Linears (addvisor.py:12-84, modeling_wav2vec2.py:254-572),
// at 1/3 of the fp16 MFMA rate (833 TFLOP/s dense peak) instead of the 157 TFLOP/s of v_mfma_f32_*_f32.
// 64 KiB of LDS per 128 x 128 K-step and 2 x 64 accumulator registers => two workgroups per CU.
template <int BM, int BN, int WM, int WN, bool PLAIN>
__global__ __launch_bounds__(64 * WM * WN, 2) void gemm_x3_kernel(const advh_gemm_desc p) {
    constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 16, NI = TN / 16;
    constexpr int NT = 64 * WM * WN, RPP = NT / 8;
    constexpr int NA = BM / RPP, NB = BN / RPP;
    constexpr int PA = BM * BK * 2, PB = BN * BK * 2;   // bytes of one plane of a K-step
    static_assert(NA >= 1 && NB >= 1 && BM % RPP == 0 && BN % RPP == 0, "loader passes");
    extern __shared__ __attribute__((aligned(16))) char dsm3[];
#ifdef ADVH_STAMPS
    if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) g_gemm_stamps[4] = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x < 16384) {
        g_wg_rec[4 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
        g_wg_rec[4 * blockIdx.x + 3] = (long long)__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)) | ((long long)__builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)) << 32);
    }
#endif
    char* ldsA = dsm3;                                  // [A hi | A lo | W hi | W lo]
    char* ldsB = dsm3 + 2 * PA;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv / WN, wn = wv % WN;
    const int tilesN = (p.N + BN - 1) / BN;
    int nwg = gridDim.x;
    int id = blockIdx.x;
    {
        const int q8 = nwg / 8, r8 = nwg % 8, xcd = id % 8;
        id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + id / 8;
    }
    int z = blockIdx.z;
    if (p.z_inner) {
        z = id % p.nz;
        id /= p.nz;
        nwg /= p.nz;
    }
    int tile_n, tile_m;
    {
        const int tilesM = nwg / tilesN, sc = (p.sc > 0 && p.sc < tilesN) ? p.sc : tilesN;
        const int s = id / (tilesM * sc), rem = id - s * tilesM * sc;
        const int wcols = min(sc, tilesN - s * sc);
        tile_m = rem / wcols;
        tile_n = s * sc + rem - tile_m * wcols;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int zh = p.nz_lo > 1 ? z / p.nz_lo : z, zw = p.nz_lo > 1 ? z % p.nz_lo : 0;

    const _Float16* A0 = (const _Float16*)p.A0 + (p.a_sZ[0] * zh + p.a_sZ2[0] * zw) * 8;
    const _Float16* A1 = (const _Float16*)p.A1 + (p.a_sZ[1] * zh + p.a_sZ2[1] * zw) * 8;
    const _Float16* Wp = (const _Float16*)p.W + p.w_sZ * z;
    const long alo0 = p.a_lo[0] * 8, alo1 = p.a_lo[1] * 8, wlo = p.w_lo;

    const int ldrow = tid >> 3;
    const int q = (tid & 7) ^ (ldrow & 7);
    unsigned rb0[PLAIN ? 1 : NA], rb1[PLAIN ? 1 : NA];
    const _Float16* wrow[PLAIN ? 1 : NB];
    const _Float16* ap[PLAIN ? NA : 1];
    const _Float16* wp0 = nullptr;
    long wstep = 0;
    if constexpr (PLAIN) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int m = min(m0 + ldrow + RPP * i, p.M - 1);
            ap[i] = A0 + ((long)m * p.a_sW[0] + p.a_c0[0] + q) * 8;
        }
        const long wld = p.w_ld ? p.w_ld : (long)p.Ktot;
        wp0 = Wp + (long)(n0 + ldrow) * wld + q * 8;
        wstep = (long)RPP * wld;
    } else {
        long safe0 = (long)p.h0 * p.a_sH[0] + (long)p.w0 * p.a_sW[0] + p.a_c0[0];
        long safe1 = (long)p.h0 * p.a_sH[1] + (long)p.w0 * p.a_sW[1] + p.a_c0[1];
        const RowDecomp rd(p.Wg, p.Hg);
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            unsigned m = m0 + ldrow + RPP * i;
            unsigned w, h, b;
            rd(m, b, h, w);
            bool ok = m < (unsigned)p.M && (int)h >= p.h0 && (int)h < p.h1 && (int)w >= p.w0 && (int)w < p.w1;
            long r0 = ok ? (long)b * p.a_sB[0] + (long)h * p.a_sH[0] + (long)w * p.a_sW[0] + p.a_c0[0] : safe0;
            long r1 = ok ? (long)b * p.a_sB[1] + (long)h * p.a_sH[1] + (long)w * p.a_sW[1] + p.a_c0[1] : safe1;
            rb0[i] = (unsigned)r0;
            rb1[i] = (unsigned)r1;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) wrow[i] = Wp + (long)(n0 + ldrow + RPP * i) * (p.w_ld ? p.w_ld : (long)p.Ktot) + q * 8;
    }

    const int fr = lane & 15, fq = lane >> 4;
    int offA[2], offB[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        int c = (kk * 4 + fq) ^ (fr & 7);
        offA[kk] = ((wm * TM + fr) * 8 + c) * 16;
        offB[kk] = ((wn * TN + fr) * 8 + c) * 16;
    }

    f32x4 acc[NI][MI], accx[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) { acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f}; accx[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    const int nk = p.Ktot / BK;
    int kq = PLAIN ? 0 : p.ktab[q];
#ifdef ADVH_STAMPS
    if (blockIdx.x == gridDim.x / 2 && tid == 0) { g_gemm_stamps[0] = __builtin_amdgcn_s_memtime(); g_gemm_stamps[1] = __builtin_amdgcn_s_memrealtime(); }
#endif
    for (int kt = 0; kt < nk; ++kt) {
#ifdef ADVH_STAMPS
        if (blockIdx.x == gridDim.x / 2 && tid == 0 && kt == 6) g_kstep[0] = __builtin_amdgcn_s_memtime();
        if (blockIdx.x == gridDim.x / 2 && tid == 0 && kt == 7) g_kstep[5] = __builtin_amdgcn_s_memtime();
#endif
        if constexpr (PLAIN) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(ap[i] + kt * BK), LDS_PTR(ldsA + (wv * 64 + NT * i) * 16), 16, 0, 0);
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(ap[i] + alo0 + kt * BK), LDS_PTR(ldsA + PA + (wv * 64 + NT * i) * 16), 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(wp0 + i * wstep + kt * BK), LDS_PTR(ldsB + (wv * 64 + NT * i) * 16), 16, 0, 0);
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(wp0 + wlo + i * wstep + kt * BK), LDS_PTR(ldsB + PB + (wv * 64 + NT * i) * 16), 16, 0, 0);
            }
        } else {
            const bool s1 = kq < 0;
            const unsigned ko = (unsigned)kq & 0x7fffffffu;
            const _Float16* base = s1 ? A1 : A0;
            const long alo = s1 ? alo1 : alo0;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const _Float16* g = base + ((unsigned long)((s1 ? rb1[i] : rb0[i]) + ko)) * 8;
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(g), LDS_PTR(ldsA + (wv * 64 + NT * i) * 16), 16, 0, 0);
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(g + alo), LDS_PTR(ldsA + PA + (wv * 64 + NT * i) * 16), 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(wrow[i] + kt * BK), LDS_PTR(ldsB + (wv * 64 + NT * i) * 16), 16, 0, 0);
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(wrow[i] + wlo + kt * BK), LDS_PTR(ldsB + PB + (wv * 64 + NT * i) * 16), 16, 0, 0);
            }
            if (kt + 1 < nk) kq = p.ktab[(kt + 1) * 8 + q];
        }
#ifdef ADVH_STAMPS
        const bool stamp_here = blockIdx.x == gridDim.x / 2 && tid == 0 && kt == 6;
        if (stamp_here) g_kstep[1] = __builtin_amdgcn_s_memtime();     // DMA issued
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef ADVH_STAMPS
        if (stamp_here) g_kstep[2] = __builtin_amdgcn_s_memtime();     // this wavefront's DMA landed
#endif
        __syncthreads();
#ifdef ADVH_STAMPS
        if (stamp_here) g_kstep[3] = __builtin_amdgcn_s_memtime();     // barrier passed
#endif
        __builtin_amdgcn_s_setprio(1);     // the MFMA phase wins issue arbitration over the co-resident workgroup's DMA issue: +1-2 % (warm clock)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            f16x8 bh[NI], bl[NI];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                bh[ni] = *(const f16x8*)(ldsB + offB[kk] + ni * 16 * 128);
                bl[ni] = *(const f16x8*)(ldsB + PB + offB[kk] + ni * 16 * 128);
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const f16x8 ah = *(const f16x8*)(ldsA + offA[kk] + mi * 16 * 128);
                const f16x8 al = *(const f16x8*)(ldsA + PA + offA[kk] + mi * 16 * 128);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) accx[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[ni], al, accx[ni][mi], 0, 0, 0);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[ni], ah, acc[ni][mi], 0, 0, 0);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) accx[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[ni], ah, accx[ni][mi], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
#ifdef ADVH_STAMPS
        if (stamp_here) g_kstep[4] = __builtin_amdgcn_s_memtime();     // MFMAs issued
#endif
        __syncthreads();
    }
#ifdef ADVH_STAMPS
    if (blockIdx.x == gridDim.x / 2 && tid == 0) { g_gemm_stamps[2] = __builtin_amdgcn_s_memtime(); g_gemm_stamps[3] = __builtin_amdgcn_s_memrealtime(); }
    if (tid == 0 && blockIdx.x < 16384) g_wg_rec[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[ni][mi][r] = fmaf(accx[ni][mi][r], SPLIT_LO_INV, acc[ni][mi][r]);

#ifdef ADVH_STAMPS
    if (blockIdx.x == gridDim.x / 2 && tid == 0) g_gemm_stamps[6] = __builtin_amdgcn_s_memtime();
#endif
    char* stage = (NI == 4 && MI * 16 * 256 * WM * WN <= 2 * (BM + BN) * BK * 2) ? dsm3 + wv * (MI * 16 * 256) : nullptr;
    if (PLAIN && p.plain_out) gemm_epilogue_plain<MI, NI, true, MI>(p, acc, stage, m0 + wm * TM, n0 + wn * TN, fr, fq, z, p.o_sZ * zh + p.o_sZ2 * zw);
    else gemm_epilogue_rows<MI, NI, true>(p, acc, stage, m0 + wm * TM, n0 + wn * TN, fr, fq, z, p.o_sZ * zh + p.o_sZ2 * zw);
#ifdef ADVH_STAMPS
    if (blockIdx.x == gridDim.x / 2 && tid == 0) g_gemm_stamps[7] = __builtin_amdgcn_s_memtime();
    if (tid == 0 && blockIdx.x < 16384) g_wg_rec[4 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (blockIdx.x == gridDim.x / 2 && tid == 0) g_gemm_stamps[5] = __builtin_amdgcn_s_memtime();
#endif
}


template <int BM, int BN, int WM, int WN>
static int launch_x3(const advh_gemm_desc& d, hipStream_t s) {
    const int tilesM = (d.M + BM - 1) / BM, tilesN = (d.N + BN - 1) / BN;
    if (tilesN * BN > d.w_rows) return ADVH_EINVAL;
    const int nz = d.nz > 0 ? d.nz : 1;
    if (d.z_inner && (long)tilesM * tilesN * nz > 0x7fffffffL) return ADVH_EINVAL;
    dim3 grid(d.z_inner ? tilesM * tilesN * nz : tilesM * tilesN, 1, d.z_inner ? 1 : nz);
    constexpr int lds = 2 * (BM + BN) * BK * 2;
    if (d.plain) hipLaunchKernelGGL((gemm_x3_kernel<BM, BN, WM, WN, true>), grid, dim3(64 * WM * WN), lds, s, d);
    else hipLaunchKernelGGL((gemm_x3_kernel<BM, BN, WM, WN, false>), grid, dim3(64 * WM * WN), lds, s, d);
    return ADVH_LAUNCH_CHECK();
}

}  // namespace advh

using namespace advh;

int advh_init_rest() {
    const int maxlds = 160 * 1024;
#define X3_ATTR(BM_, BN_, WM_, WN_)                                                                                                        \
    if (hipFuncSetAttribute((const void*)gemm_x3_kernel<BM_, BN_, WM_, WN_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, maxlds) != hipSuccess || \
        hipFuncSetAttribute((const void*)gemm_x3_kernel<BM_, BN_, WM_, WN_, false>, hipFuncAttributeMaxDynamicSharedMemorySize, maxlds) != hipSuccess)   \
        return ADVH_ELAUNCH;
    X3_ATTR(128, 128, 2, 2) X3_ATTR(256, 64, 4, 1) X3_ATTR(256, 32, 4, 1)
#undef X3_ATTR

    return ADVH_OK;
}

#ifdef ADVH_STAMPS
extern "C" int advh_debug_wg_records(long long* out, int n) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_rec), sizeof(long long) * 4 * n) == hipSuccess ? ADVH_OK : ADVH_ELAUNCH; }
extern "C" int advh_debug_kstep(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_kstep), sizeof(long long) * 8) == hipSuccess ? ADVH_OK : ADVH_ELAUNCH; }
extern "C" int advh_debug_gemm_stamps(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gemm_stamps), sizeof(long long) * 8) == hipSuccess ? ADVH_OK : ADVH_ELAUNCH; }
#endif

extern "C" int advh_gemm_f16(const advh_gemm_desc* d, int tile, advh_stream_t stream) {
    if (!d || !d->A0 || !d->W || !d->ktab || (!d->out_h && !d->out_f)) return ADVH_EINVAL;
    if (d->w_rows < d->N || d->M <= 0 || d->N <= 0 || d->Ktot <= 0 || d->Ktot % BK || d->N % 4 || d->Hg <= 0 || d->Wg <= 0) return ADVH_EINVAL;
    if (d->n_div <= 0 || d->n_div % 4 || d->h0 < 0 || d->w0 < 0 || d->h1 > d->Hg || d->w1 > d->Wg || d->h0 >= d->h1 || d->w0 >= d->w1)
        return ADVH_EINVAL;
    if (d->act < ADVH_ACT_NONE || d->act > ADVH_ACT_LEAKY) return ADVH_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    // same rule as addvisor_hip/gemm.py pick_tile
    if (tile == ADVH_TILE_AUTO) tile = d->N > 64 ? ADVH_TILE_128x128 : (d->N > 32 ? ADVH_TILE_256x64 : ADVH_TILE_256x32);
    if (d->nz_lo < 0 || (d->nz_lo > 1 && (d->nz <= 0 || d->nz % d->nz_lo))) return ADVH_EINVAL;
    if (d->plain && !d->ktab_identity) return ADVH_EINVAL;
    if (d->plain_out && (d->n_div < d->N || d->ph_r > 0 || d->n_sub > 1 || d->h0 != 0 || d->w0 != 0 || d->h1 != d->Hg || d->w1 != d->Wg))
        return ADVH_EINVAL;
    if (d->split) {                                      // fp32-class instance: split-format operands, 3 MFMAs per fragment pair
        switch (tile) {
            case ADVH_TILE_128x128: return launch_x3<128, 128, 2, 2>(*d, s);
            case ADVH_TILE_256x64: return launch_x3<256, 64, 4, 1>(*d, s);
            case ADVH_TILE_256x32: return launch_x3<256, 32, 4, 1>(*d, s);
            default: return ADVH_EUNSUPPORTED;
        }
    }
    switch (tile) {
        case ADVH_TILE_128x128: return launch<128, 128, 2, 2, 4>(*d, s);   // 4 wavefronts per SIMD: <= 128 VGPRs, checked spill-free
        case ADVH_TILE_256x64: return launch<256, 64, 4, 1, 3>(*d, s);
        case ADVH_TILE_256x32: return launch<256, 32, 4, 1, 3>(*d, s);
        case ADVH_TILE_256x128_W8: return launch<256, 128, 4, 2, 3>(*d, s);
        case ADVH_TILE_128x256_W8: return launch<128, 256, 2, 4, 3>(*d, s);
        default: return ADVH_EINVAL;       // the ring / pipelined / persistent variants of rounds 1-2 measured slower on every pipeline shape and were removed
    }
}

ADVH_SPLIT_FLAG_SETTER(advh_split_flag_gemm)
