// fp32-class attention backward on the fp16 matrix cores (split arithmetic), head dims <= 64, T <= 256.
//
// Same contract as attention_bwd_f32.hip (dqkv from (qkv, dctx), all split-format plane pairs; the reference differentiates
// with fp32 autograd: captum_saliency.py:116-135, loss_function.py:46-53 through modeling_wav2vec2.py:438-463) and the same two
// passes -- A: one 16-query tile per wavefront (row max / 1/sum / delta, dQ), B: one 16-key tile per wavefront (dK, dV), scores
// recomputed in the orientation that makes every product sum over the accumulator's row index -- but every product is three
// v_mfma_f32_16x16x32_f16 on (hi, lo) operands (acc += Ah*Bh; accx += Ah*Bl + Al*Bh; result acc + accx * 2^-11) instead of eight
// v_mfma_f32_16x16x4_f32: 5.3x fewer matrix-core cycles (the fp32 form was 48 % MFMA-busy for 594 us per launch of 64 x 16 heads x
// 199 frames and 15.6 % of the IntegratedGradients step, profiles/r03_attention_bwd_f32_pmc.txt).
//  * Q, K, V, dO arrive split: staging is a plain copy of both planes into LDS (16-byte chunks XOR-swizzled by the row), and the
//    fragments of a product over d are ds_read_b128 of 8 consecutive d of one row, exactly the MFMA operand.
//  * Products over keys / queries need the TRANSPOSED operand (8 consecutive rows of one column): ds_read_b64_tr_b16 returns
//    lane i of a 16-lane group column c0 + i of four consecutive rows -- two of them per plane and 32-deep step, no transposed copy.
//    The contraction index may be visited in any order as long as both operands agree: step ss takes rows 32 ss + 4 g + j (j < 4)
//    and 32 ss + 16 + 4 g + j - 4 (j >= 4), which are accumulator registers j of score tiles 2 ss and 2 ss + 1 -- dS / P go from
//    registers into the other operand after an in-register split (P <= 1 unchecked, dS with the range check of device_math.h).
//  * S, dP, softmax statistics, P and dS are fp32; only the matrix products see 22-bit operands (S and dP: exact products of
//    split inputs; dQ / dK / dV: P and dS rounded to hi + lo).
// Head dim 128 (XLS-R's 120) does not fit two matrices in two planes: attention_bwd_f32_kernel keeps that case.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include "addvisor_hip.h"
#include "common.h"
#include "device_math.h"

namespace advh {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __fp16 trvec __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef const __attribute__((address_space(3))) char* lds_cptr;
constexpr float LOG2E_X3 = 1.4426950408889634f;

template <int NT, int D>
struct AttBwdX3 {
    static constexpr int NS = (NT + 1) / 2, ROWS = NS * 32;          // rows per plane: whole 32-row steps, rows >= T are zero
    static constexpr int PLANE = ROWS * D;                           // halfs
    static constexpr int LDS_BYTES = 4 * PLANE * 2 + 3 * ROWS * 4;   // two matrices x two planes + row statistics
};

// element offset (halfs) of 16-byte chunk c of row r
template <int D> __device__ __forceinline__ int swz(int r, int c) { return r * D + ((c ^ (r & (D / 8 - 1))) << 3); }

// rows [0, T) x channels [0, dm) of both planes of a split matrix (row stride ld) -> LDS planes, zero elsewhere
template <int ROWS, int D, int NTH>
__device__ __forceinline__ void stage_planes(_Float16* ph, _Float16* pl, const _Float16* src, long lo, long ld, int T, int dm, int tid) {
    constexpr int CH = D / 8;
    for (int i = tid; i < ROWS * CH; i += NTH) {
        const int row = i / CH, c = i % CH;
        f16x8 h = {0, 0, 0, 0, 0, 0, 0, 0}, l = {0, 0, 0, 0, 0, 0, 0, 0};
        if (row < T && c * 8 < dm) {
            h = *(const f16x8*)(src + (long)row * ld + c * 8);
            l = *(const f16x8*)(src + lo + (long)row * ld + c * 8);
        }
        *(f16x8*)(ph + swz<D>(row, c)) = h;
        *(f16x8*)(pl + swz<D>(row, c)) = l;
    }
}

// transposed fragment: lane (g, fr) receives column c0 + fr of rows r0 + 4 g .. + 3 and r0 + 16 + 4 g .. + 3 of an LDS plane.
// r0 is a multiple of 32, so the swizzle term of a row depends on the lane only: tr_off<D>(g, fr, c0) is the lane's byte offset of the
// first read within a 32-row step (computed once per wavefront and column group), the second read sits 16 rows further.
template <int D> __device__ __forceinline__ int tr_off(int g, int fr, int c0) {
    const int q = fr >> 2, p = fr & 3;
    return 2 * swz<D>(4 * g + q, (c0 >> 3) + (p >> 1)) + (p & 1) * 8;
}
template <int D>
__device__ __forceinline__ f16x8 tr8(lds_cptr plane, int off) {
    const auto* pa = (const __attribute__((address_space(3))) trvec*)(plane + off);
    const auto* pb = (const __attribute__((address_space(3))) trvec*)(plane + off + 16 * (2 * D));
    trvec lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) trvec*)pa);
    trvec hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) trvec*)pb);
    f16x8 r;
    __builtin_memcpy(&r, &lo, 8);
    __builtin_memcpy((char*)&r + 8, &hi, 8);
    return r;
}

__device__ __forceinline__ f16x8 gfrag(const _Float16* p, bool in) {
    return in ? *(const f16x8*)p : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
}

#define MFMA_X3(accm, accx, ah, al, bh, bl)                                          \
    do {                                                                             \
        accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, accx, 0, 0, 0);        \
        accm = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, accm, 0, 0, 0);        \
        accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, accx, 0, 0, 0);        \
    } while (0)

template <int NT, int D, int NW>
__global__ __launch_bounds__(64 * NW) void attention_bwd_x3_kernel(const _Float16* __restrict__ qkv, long qkv_lo, const _Float16* __restrict__ dctx,
                                                                   long dctx_lo, _Float16* __restrict__ dqkv, long dqkv_lo, int T, int H, int dm,
                                                                   float scale) {
    typedef AttBwdX3<NT, D> G;
    constexpr int ROWS = G::ROWS, PLANE = G::PLANE, KK = D / 32, DG = D / 16, NS = G::NS;
    extern __shared__ __attribute__((aligned(16))) _Float16 smx[];
    _Float16* P0h = smx;                                  // K (pass A), Q (pass B)
    _Float16* P0l = smx + PLANE;
    _Float16* P1h = smx + 2 * PLANE;                      // V (pass A), dO (pass B)
    _Float16* P1l = smx + 3 * PLANE;
    float* rmax = (float*)(smx + 4 * PLANE);
    float* rinv = rmax + ROWS;
    float* rdel = rinv + ROWS;
    const lds_cptr L0h = (lds_cptr)((__attribute__((address_space(3))) void*)P0h);
    const lds_cptr L0l = L0h + 2 * PLANE, L1h = L0h + 4 * PLANE, L1l = L0h + 6 * PLANE;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int fr = lane & 15, g = lane >> 4;
    const int head = blockIdx.x, b = blockIdx.y;
    const long ld = 3L * H;
    const _Float16* base = qkv + (long)b * T * ld + head * dm;           // q at +0, k at +H, v at +2H
    const _Float16* dob = dctx + (long)b * T * H + head * dm;
    _Float16* dbase = dqkv + (long)b * T * ld + head * dm;
    const float c2 = scale * LOG2E_X3;
    const int nst = (T + 31) / 32;                                       // 32-row steps that hold any row < T
    int troff[DG], rowoff[KK];                                           // per-lane LDS offsets (bytes / halfs) within a 32-row step / 16-row tile
#pragma unroll
    for (int dt = 0; dt < DG; ++dt) troff[dt] = tr_off<D>(g, fr, 16 * dt);
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) rowoff[kk] = swz<D>(fr, kk * 4 + g);

    for (int i = tid; i < 3 * ROWS; i += 64 * NW) rmax[i] = 0.f;          // rows past the last query tile: finite statistics (pass B selects them away)
    stage_planes<ROWS, D, 64 * NW>(P0h, P0l, base + H, qkv_lo, ld, T, dm, tid);
    stage_planes<ROWS, D, 64 * NW>(P1h, P1l, base + 2 * H, qkv_lo, ld, T, dm, tid);
    __syncthreads();

    // ------------------------------------------------------------------ pass A: query tiles
    for (int qt = wv; qt * 16 < T; qt += NW) {
        const int qrow = qt * 16 + fr, qr = qrow < T ? qrow : T - 1;
        // S^T for all key tiles, then dP^T for all key tiles: one pair of global fragments live at a time (both products in one loop
        // needed q and dO fragments + four accumulators beside the 2 NT score registers and spilled at NT >= 13)
        f32x4 s[NT], dp[NT];
        {
            f16x8 qh[KK], ql[KK];
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                const int c = kk * 4 + g;
                qh[kk] = gfrag(base + (long)qr * ld + c * 8, c * 8 < dm);
                ql[kk] = gfrag(base + qkv_lo + (long)qr * ld + c * 8, c * 8 < dm);
            }
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                f32x4 sm_ = {0.f, 0.f, 0.f, 0.f}, sx = sm_;
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) {
                    const int o = kt * 16 * D + rowoff[kk];
                    const f16x8 kh = *(const f16x8*)(P0h + o), kl = *(const f16x8*)(P0l + o);
                    MFMA_X3(sm_, sx, kh, kl, qh[kk], ql[kk]);          // S^T  [key][q]
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) s[kt][r] = fmaf(sx[r], SPLIT_LO_INV, sm_[r]);
                if ((kt & 1) == 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
        {
            f16x8 oh[KK], ol[KK];
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                const int c = kk * 4 + g;
                oh[kk] = gfrag(dob + (long)qr * H + c * 8, c * 8 < dm);
                ol[kk] = gfrag(dob + dctx_lo + (long)qr * H + c * 8, c * 8 < dm);
            }
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                f32x4 pm = {0.f, 0.f, 0.f, 0.f}, px = pm;
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) {
                    const int o = kt * 16 * D + rowoff[kk];
                    const f16x8 vh = *(const f16x8*)(P1h + o), vl = *(const f16x8*)(P1l + o);
                    MFMA_X3(pm, px, vh, vl, oh[kk], ol[kk]);           // dP^T [key][q]
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) dp[kt][r] = fmaf(px[r], SPLIT_LO_INV, pm[r]);
                if ((kt & 1) == 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = (kt * 16 + g * 4 + r < T) ? s[kt][r] * c2 : -INFINITY;       // log2 domain
                s[kt][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float e = __builtin_amdgcn_exp2f(s[kt][r] - mx); s[kt][r] = e; sum += e; }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.f / sum;
        float del = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { s[kt][r] *= inv; del += s[kt][r] * dp[kt][r]; }
        del += __shfl_xor(del, 16, 64);
        del += __shfl_xor(del, 32, 64);
        if (g == 0) { rmax[qrow] = mx; rinv[qrow] = inv; rdel[qrow] = del; }                 // qrow < ROWS always
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[kt][r] = s[kt][r] * (dp[kt][r] - del) * scale;     // dS^T [key][q]; 0 for keys >= T (p = 0)
        f32x4 om[DG], ox[DG];
#pragma unroll
        for (int dt = 0; dt < DG; ++dt) { om[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; ox[dt] = om[dt]; }
#pragma unroll
        for (int ss = 0; ss < NS; ++ss) {
            float v[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] = s[2 * ss][r]; v[4 + r] = (2 * ss + 1 < NT) ? s[(2 * ss + 1 < NT) ? 2 * ss + 1 : 0][r] : 0.f; }
            f16x8 bh, bl;
            split_f32_vec<8>(v, bh, bl);
#pragma unroll
            for (int dt = 0; dt < DG; ++dt) {
                const f16x8 ah = tr8<D>(L0h, 32 * ss * (2 * D) + troff[dt]), al = tr8<D>(L0l, 32 * ss * (2 * D) + troff[dt]);
                MFMA_X3(om[dt], ox[dt], ah, al, bh, bl);           // dQ^T [d][q] += K^T dS^T
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (qrow < T) {
#pragma unroll
            for (int dt = 0; dt < DG; ++dt) {
                if (dt * 16 + g * 4 >= dm) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaf(ox[dt][r], SPLIT_LO_INV, om[dt][r]);
                store_h_rt<4>(dbase, (long)qrow * ld + dt * 16 + g * 4, dqkv_lo, v);
            }
        }
    }
    __syncthreads();

    // ------------------------------------------------------------------ pass B: key tiles
    stage_planes<ROWS, D, 64 * NW>(P0h, P0l, base, qkv_lo, ld, T, dm, tid);              // Q
    stage_planes<ROWS, D, 64 * NW>(P1h, P1l, dob, dctx_lo, (long)H, T, dm, tid);         // dO
    __syncthreads();
    for (int kt = wv; kt * 16 < T; kt += NW) {
        const int krow = kt * 16 + fr, kr_ = krow < T ? krow : T - 1;
        f16x8 kh[KK], kl[KK], vh[KK], vl[KK];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const int c = kk * 4 + g;
            const bool in = c * 8 < dm;
            kh[kk] = gfrag(base + H + (long)kr_ * ld + c * 8, in);
            kl[kk] = gfrag(base + H + qkv_lo + (long)kr_ * ld + c * 8, in);
            vh[kk] = gfrag(base + 2 * H + (long)kr_ * ld + c * 8, in);
            vl[kk] = gfrag(base + 2 * H + qkv_lo + (long)kr_ * ld + c * 8, in);
        }
        f32x4 dkm[DG], dkx[DG], dvm[DG], dvx[DG];
#pragma unroll
        for (int dt = 0; dt < DG; ++dt) { dkm[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dkx[dt] = dkm[dt]; dvm[dt] = dkm[dt]; dvx[dt] = dkm[dt]; }
        for (int ss = 0; ss < nst; ++ss) {
            float p[8], ds[8];
            const int sb = ss * 32 * D;                                       // first element of the step's rows in a plane
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 sm_ = {0.f, 0.f, 0.f, 0.f}, sx = sm_, pm = sm_, px = sm_;
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) {
                    const int o = sb + h * 16 * D + rowoff[kk];             // rows >= T are zero in LDS
                    const f16x8 qah = *(const f16x8*)(P0h + o), qal = *(const f16x8*)(P0l + o);
                    const f16x8 oah = *(const f16x8*)(P1h + o), oal = *(const f16x8*)(P1l + o);
                    MFMA_X3(sm_, sx, qah, qal, kh[kk], kl[kk]);     // S  [q][key]
                    MFMA_X3(pm, px, oah, oal, vh[kk], vl[kk]);      // dP [q][key]
                }
                const int q0 = ss * 32 + h * 16 + g * 4;            // this lane's four query rows; its key column = krow
                const float4 mx4 = *(const float4*)(rmax + q0), iv4 = *(const float4*)(rinv + q0), dl4 = *(const float4*)(rdel + q0);
                const float mxs[4] = {mx4.x, mx4.y, mx4.z, mx4.w}, ivs[4] = {iv4.x, iv4.y, iv4.z, iv4.w}, dls[4] = {dl4.x, dl4.y, dl4.z, dl4.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {                       // branch-free: statistics of rows >= T are finite (initialised), selected away
                    const float pv = __builtin_amdgcn_exp2f(fmaf(sx[r], SPLIT_LO_INV, sm_[r]) * c2 - mxs[r]) * ivs[r];   // v_exp_f32: argument <= ~0, tiny results may flush
                    const float dv = pv * (fmaf(px[r], SPLIT_LO_INV, pm[r]) - dls[r]) * scale;
                    const bool ok = q0 + r < T && krow < T;
                    p[4 * h + r] = ok ? pv : 0.f;
                    ds[4 * h + r] = ok ? dv : 0.f;
                }
            }
            f16x8 ph, pl, dh, dl;
#pragma unroll
            for (int j = 0; j < 8; ++j) { _Float16 a, c; split_f32_raw(p[j], a, c); ph[j] = a; pl[j] = c; }
            split_f32_vec<8>(ds, dh, dl);
#pragma unroll
            for (int dt = 0; dt < DG; ++dt) {
                const int a = 2 * sb + troff[dt];                   // one per-lane address per column group; planes and the +16 rows are immediates
                const f16x8 qth = tr8<D>(L0h, a), qtl = tr8<D>(L0l, a);
                const f16x8 oth = tr8<D>(L1h, a), otl = tr8<D>(L1l, a);
                MFMA_X3(dkm[dt], dkx[dt], qth, qtl, dh, dl);        // dK^T [d][key] += Q^T dS
                MFMA_X3(dvm[dt], dvx[dt], oth, otl, ph, pl);        // dV^T [d][key] += dO^T P
            }
        }
        if (krow < T) {
#pragma unroll
            for (int dt = 0; dt < DG; ++dt) {
                if (dt * 16 + g * 4 >= dm) continue;
                float kv[4], vv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { kv[r] = fmaf(dkx[dt][r], SPLIT_LO_INV, dkm[dt][r]); vv[r] = fmaf(dvx[dt][r], SPLIT_LO_INV, dvm[dt][r]); }
                store_h_rt<4>(dbase, (long)krow * ld + H + dt * 16 + g * 4, dqkv_lo, kv);
                store_h_rt<4>(dbase, (long)krow * ld + 2 * H + dt * 16 + g * 4, dqkv_lo, vv);
            }
        }
    }
}

template <int NT, int D, int NW>
static int launch_att_bwd_x3(const void* qkv, long qkv_lo, const void* dctx, long dctx_lo, void* dqkv, long dqkv_lo, int B, int T, int H,
                             int heads, int dm, float scale, hipStream_t s) {
    constexpr int lds = AttBwdX3<NT, D>::LDS_BYTES;
    static_assert(lds <= 160 * 1024, "two matrices in two planes must fit");
    if (advh_ensure_lds((const void*)attention_bwd_x3_kernel<NT, D, NW>) != ADVH_OK) return ADVH_ELAUNCH;
    hipLaunchKernelGGL((attention_bwd_x3_kernel<NT, D, NW>), dim3(heads, B), dim3(64 * NW), lds, s, (const _Float16*)qkv, qkv_lo,
                       (const _Float16*)dctx, dctx_lo, (_Float16*)dqkv, dqkv_lo, T, H, dm, scale);
    return ADVH_LAUNCH_CHECK();
}

}  // namespace advh

using namespace advh;

// head dims <= 64 (D = 32 | 64), T <= 256; arguments validated by advh_attention_bwd_split (attention_bwd_f32.hip)
int advh_attention_bwd_x3_launch(const void* qkv, long qkv_lo, const void* dctx, long dctx_lo, void* dqkv, long dqkv_lo, int B, int T, int H,
                                 int heads, hipStream_t s) {
    const int dm = H / heads;
    const int D = dm <= 32 ? 32 : 64;
    const float scale = 1.f / sqrtf((float)dm);
    const int nt = (T + 15) / 16;
#define ATX(NT_, D_, NW_) return launch_att_bwd_x3<NT_, D_, NW_>(qkv, qkv_lo, dctx, dctx_lo, dqkv, dqkv_lo, B, T, H, heads, dm, scale, s)
    // eight wavefronts (two per SIMD, 256 VGPRs each: the 13-tile / head-dim-64 instance keeps 68 bytes per lane of scratch for values
    // that live across the passes, outside the product loops, and still beats four wavefronts 293 to 392 us); 16 key tiles x head dim 64
    // need 2 x 16 score registers per lane in pass A: four wavefronts with the whole register file (356 us at 64 x 12 x 249) instead of
    // 352 bytes of scratch (372 us).  profiles/r03_attention_bwd_x3.txt
#define ATX_D(D_)                                                                     \
    do {                                                                              \
        if (nt <= 4) ATX(4, D_, 8); else if (nt <= 8) ATX(8, D_, 8); else if (nt <= 13) ATX(13, D_, 8); else ATX(16, D_, 4); \
    } while (0)
    if (D == 32) ATX_D(32);
    else ATX_D(64);
#undef ATX_D
#undef ATX
    return ADVH_EUNSUPPORTED;
}

ADVH_SPLIT_FLAG_SETTER(advh_split_flag_attention_bwd_x3)
