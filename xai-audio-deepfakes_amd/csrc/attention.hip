// Fused self-attention for short sequences (T <= 256): softmax(Q K^T / sqrt(d)) V, no mask
// (transformers/.../modeling_wav2vec2.py:438-548; the reference forces the math SDP path,
// train_addvisor.py:21-23).  wav2vec2 sees T = 199 frames for a 4 s clip (249 for 5 s), so the whole
// K and V of one (clip, head) fit in LDS and the softmax is single-pass: no online rescaling.
//
// One workgroup per (head, clip); each of its 4 wavefronts owns 16-query tiles.  S^T = K Q^T is
// computed with the KEY on the MFMA row, so a lane ends up with 4*NT scores of ONE query: the row
// max / sum are register reductions plus two cross-lane steps, and the fp16 P registers are already
// the B operand of O^T = V^T P^T (cdna_hip_programming.md §3 "An accumulator tile as the next MFMA's
// operand": the k order inside a 32-key step is permuted, and V^T is read from LDS in that same order).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include "addvisor_hip.h"
#include "common.h"
#include "device_math.h"

namespace advh {

constexpr float LOG2E = 1.4426950408889634f;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// D = head dim rounded up to a multiple of 32 (MFMA k step); dm = the real head dim (a multiple of 8): chunks beyond
// it are zero-filled on load and skipped on store (XLS-R-2B: 1920 / 16 heads = 120 -> D = 128).
template <int NT, int D>
__global__ __launch_bounds__(256, (D <= 64 ? 2 : 1)) void attention_kernel(const _Float16* __restrict__ qkv, _Float16* __restrict__ ctx,
                                                        int T, int H, int dm, float scale) {
    constexpr int NKEY = NT * 16, CH = D / 8, VP = NKEY + 64 /* row pitch incl. the per-8-rows skew that spreads the transposing stores over the banks */, NS = (NT + 1) / 2, KK = D / 32, DT = D / 16;
    __shared__ __attribute__((aligned(16))) _Float16 Ks[NKEY * D];
    __shared__ __attribute__((aligned(16))) _Float16 Vt[D * VP];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int head = blockIdx.x, b = blockIdx.y;
    const long ld = 3L * H;
    const _Float16* base = qkv + (long)b * T * ld + head * dm;
    const int chm = dm / 8;                          // real 16-byte chunks per row

    // ---- stage K (row-major, 16-byte chunks XOR-swizzled by row) and V^T; keys >= T are zero
    // all global loads of this thread are issued before the first LDS store: one memory latency per workgroup instead
    // of one per loop iteration (the transposing stores are scalar and would otherwise serialise behind each load)
    constexpr int NIT = (NKEY * CH + 255) / 256;
    f16x8 kreg[NIT], vreg[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = tid + it * 256, key = i / CH, c = i % CH;
        kreg[it] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
        vreg[it] = kreg[it];
        if (i < NKEY * CH && key < T && c < chm) {
            kreg[it] = *(const f16x8*)(base + (long)key * ld + H + c * 8);
            vreg[it] = *(const f16x8*)(base + (long)key * ld + 2 * H + c * 8);
        }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = tid + it * 256, key = i / CH, c = i % CH;
        if (i >= NKEY * CH) break;
        *(f16x8*)(Ks + key * D + ((c ^ (key & (CH - 1))) * 8)) = kreg[it];
#pragma unroll
        for (int j = 0; j < 8; ++j) Vt[(c * 8 + j) * VP + (c & 7) * 8 + key] = vreg[it][j];
    }
    __syncthreads();

    const int fr = lane & 15, g = lane >> 4;
    for (int qt = wv; qt * 16 < T; qt += 4) {
        int qrow = qt * 16 + fr;
        int qr = qrow < T ? qrow : T - 1;
        f16x8 qf[KK];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
            qf[kk] = (kk * 4 + g < chm) ? *(const f16x8*)(base + (long)qr * ld + kk * 32 + g * 8) : f16x8{0, 0, 0, 0, 0, 0, 0, 0};

        // S^T tiles: rows = keys (4g + r within the tile), column = this lane's query
        f32x4 s[NT];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                int key = kt * 16 + fr, c = kk * 4 + g;
                f16x8 kf = *(const f16x8*)(Ks + key * D + ((c ^ (key & (CH - 1))) * 8));
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[kk], s[kt], 0, 0, 0);
            }
            if ((kt & 1) == 1) __builtin_amdgcn_sched_barrier(0);       // keep the K-fragment reads from being hoisted en bloc (VGPR pressure)
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = (kt * 16 + g * 4 + r < T) ? s[kt][r] * (scale * LOG2E) : -INFINITY;   // log2 domain: exp(x) = v_exp_f32(x log2 e)
                s[kt][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { float e = __builtin_amdgcn_exp2f(s[kt][r] - mx); s[kt][r] = e; sum += e; }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.f / sum;

        // P as the B operand of each 32-key step: elements 0-3 from tile 2s, 4-7 from tile 2s+1
        f16x8 pf[NS];
#pragma unroll
        for (int ss = 0; ss < NS; ++ss) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pf[ss][r] = (_Float16)(s[2 * ss][r] * inv);
                pf[ss][4 + r] = (2 * ss + 1 < NT) ? (_Float16)(s[(2 * ss + 1 < NT) ? 2 * ss + 1 : 0][r] * inv) : (_Float16)0.f;
            }
        }
        // O^T = V^T P^T, key steps outermost so only one step's V^T fragments are live at a time
        f32x4 o[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ss = 0; ss < NS; ++ss) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const _Float16* vr = Vt + (dt * 16 + fr) * VP + (((dt * 16 + fr) >> 3) & 7) * 8 + ss * 32 + g * 4;
                f16x4 lo = *(const f16x4*)vr;
                f16x4 hi = (2 * ss + 1 < NT) ? *(const f16x4*)(vr + 16) : f16x4{0, 0, 0, 0};
                f16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[ss], o[dt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (qrow < T) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                if (dt * 16 + g * 4 >= dm) continue;
                f16x4 hv = {(_Float16)o[dt][0], (_Float16)o[dt][1], (_Float16)o[dt][2], (_Float16)o[dt][3]};
                *(f16x4*)(ctx + ((long)b * T + qrow) * H + head * dm + dt * 16 + g * 4) = hv;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Head dim 64 (wav2vec2-base / -large): K and V are both staged ROW-major by the LDS DMA (no register round trip, no
// scalar transposing stores, no V^T buffer) and the V^T operand of O^T = V^T P^T is read with the transposing LDS load
// ds_read_b64_tr_b16: lane i of a 16-lane group receives dim d0+i of four consecutive keys, and the key order
// (4g+q | 16+4g+q) of the two reads is exactly the order the P registers already have.  53 KB of LDS at T = 199
// => three workgroups per CU instead of two.
typedef __fp16 trvec __attribute__((__vector_size__(4 * sizeof(__fp16))));
#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int NT>
__global__ __launch_bounds__(256, 2) void attention_tr_kernel(const _Float16* __restrict__ qkv, _Float16* __restrict__ ctx,
                                                              int T, int H, float scale) {
    constexpr int D = 64, NKEY = NT * 16, CH = 8, NS = (NT + 1) / 2, KK = 2, DT = 4;
    __shared__ __attribute__((aligned(16))) _Float16 Ks[NKEY * D];
    __shared__ __attribute__((aligned(16))) _Float16 Vs[NKEY * D];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int head = blockIdx.x, b = blockIdx.y;
    const long ld = 3L * H;
    const _Float16* base = qkv + (long)b * T * ld + head * D;
    for (int i = tid; i < NKEY * CH; i += 256) {          // chunk c of key r sits at slot c ^ (r & 7); keys >= T re-read key T-1 (finite,
        const int key = i / CH, c = (i % CH) ^ (key & 7);  // masked by the softmax / multiplied by P = 0)
        const _Float16* src = base + (long)min(key, T - 1) * ld + c * 8;
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src + H), LDS_PTR((char*)Ks + (size_t)(i - lane) * 16), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src + 2 * H), LDS_PTR((char*)Vs + (size_t)(i - lane) * 16), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int fr = lane & 15, g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const auto* vs3 = (const __attribute__((address_space(3))) char*)LDS_PTR(Vs);
    for (int qt = wv; qt * 16 < T; qt += 4) {
        const int qrow = qt * 16 + fr;
        const int qr = qrow < T ? qrow : T - 1;
        f16x8 qf[KK];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) qf[kk] = *(const f16x8*)(base + (long)qr * ld + kk * 32 + g * 8);
        f32x4 s[NT];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                const int key = kt * 16 + fr, c = kk * 4 + g;
                const f16x8 kf = *(const f16x8*)(Ks + key * D + ((c ^ (key & 7)) * 8));
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[kk], s[kt], 0, 0, 0);
            }
            if ((kt & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = (kt * 16 + g * 4 + r < T) ? s[kt][r] * (scale * LOG2E) : -INFINITY;   // log2 domain: exp(x) = v_exp_f32(x log2 e)
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
        f16x8 pf[NS];
#pragma unroll
        for (int ss = 0; ss < NS; ++ss)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pf[ss][r] = (_Float16)(s[2 * ss][r] * inv);
                pf[ss][4 + r] = (2 * ss + 1 < NT) ? (_Float16)(s[(2 * ss + 1 < NT) ? 2 * ss + 1 : 0][r] * inv) : (_Float16)0.f;
            }
        f32x4 o[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ss = 0; ss < NS; ++ss) {
            // keys 32 ss + 4g + q (elements 0-3) and 32 ss + 16 + 4g + q (elements 4-7); a missing odd tile re-reads the even one (P = 0)
            const int ra = 32 * ss + 4 * g + q, rb = (2 * ss + 1 < NT) ? ra + 16 : ra;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int chn = dt * 2 + (pp >> 1), sub = (pp & 1) * 8;
                trvec lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                    (__attribute__((address_space(3))) trvec*)(vs3 + ((size_t)ra * CH + (chn ^ (ra & 7))) * 16 + sub));
                trvec hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                    (__attribute__((address_space(3))) trvec*)(vs3 + ((size_t)rb * CH + (chn ^ (rb & 7))) * 16 + sub));
                f16x8 vf;
                __builtin_memcpy(&vf, &lo, 8);
                __builtin_memcpy((char*)&vf + 8, &hi, 8);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[ss], o[dt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (qrow < T) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                f16x4 hv = {(_Float16)o[dt][0], (_Float16)o[dt][1], (_Float16)o[dt][2], (_Float16)o[dt][3]};
                *(f16x4*)(ctx + ((long)b * T + qrow) * H + head * D + dt * 16 + g * 4) = hv;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// fp32-class instance: q, k, v and the context are plane pairs in the split format of device_math.h; both products cost
// three MFMAs per fragment pair (hi*hi into the main accumulator, hi*lo + lo*hi into the cross accumulator, 2^-11),
// the softmax is fp32 as before and the probabilities are split before O^T = V^T P^T.  K and V^T (hi and lo) of one
// (clip, head) stay in LDS: 144 KB at T <= 256, head dim <= 64 => one workgroup per CU (attention is 4 % of the FLOPs);
// eight wavefronts share it, so the 13 query tiles of a 4 s clip take two rounds and each SIMD has a second wavefront to
// issue while the first waits on LDS.
template <int NT, int D>
__global__ __launch_bounds__(512, 1) void attention_x3_kernel(const _Float16* __restrict__ qkv, long qkv_lo, _Float16* __restrict__ ctx,
                                                              long ctx_lo, int T, int H, int dm, float scale) {
    constexpr int NKEY = NT * 16, CH = D / 8, VP = NKEY + 64, NS = (NT + 1) / 2, KK = D / 32, DT = D / 16;
    extern __shared__ __attribute__((aligned(16))) _Float16 att_lds[];
    _Float16* Ks = att_lds;                      // [2][NKEY * D]
    _Float16* Vt = att_lds + 2 * NKEY * D;       // [2][D * VP]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int head = blockIdx.x, b = blockIdx.y;
    const long ld = 3L * H;
    const _Float16* base = qkv + (long)b * T * ld + head * dm;
    const int chm = dm / 8;

    constexpr int NIT = (NKEY * CH + 511) / 512;
    for (int pl = 0; pl < 2; ++pl) {             // plane by plane: half the staging registers
        const _Float16* bp = base + (pl ? qkv_lo : 0);
        f16x8 kreg[NIT], vreg[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 512, key = i / CH, c = i % CH;
            kreg[it] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            vreg[it] = kreg[it];
            if (i < NKEY * CH && key < T && c < chm) {
                kreg[it] = *(const f16x8*)(bp + (long)key * ld + H + c * 8);
                vreg[it] = *(const f16x8*)(bp + (long)key * ld + 2 * H + c * 8);
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 512, key = i / CH, c = i % CH;
            if (i >= NKEY * CH) break;
            *(f16x8*)(Ks + pl * NKEY * D + key * D + ((c ^ (key & (CH - 1))) * 8)) = kreg[it];
#pragma unroll
            for (int j = 0; j < 8; ++j) Vt[pl * D * VP + (c * 8 + j) * VP + (c & 7) * 8 + key] = vreg[it][j];
        }
    }
    __syncthreads();

    const int fr = lane & 15, g = lane >> 4;
    for (int qt = wv; qt * 16 < T; qt += 8) {            // 8 wavefronts (2 per SIMD): 13 query tiles at T = 199 take 2 rounds
        const int qrow = qt * 16 + fr;
        const int qr = qrow < T ? qrow : T - 1;
        f16x8 qh[KK], ql[KK];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const bool in = kk * 4 + g < chm;
            qh[kk] = in ? *(const f16x8*)(base + (long)qr * ld + kk * 32 + g * 8) : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            ql[kk] = in ? *(const f16x8*)(base + qkv_lo + (long)qr * ld + kk * 32 + g * 8) : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
        f32x4 s[NT];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            f32x4 sm = f32x4{0.f, 0.f, 0.f, 0.f}, sx = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                const int key = kt * 16 + fr, c = kk * 4 + g;
                const int o = key * D + ((c ^ (key & (CH - 1))) * 8);
                const f16x8 kh = *(const f16x8*)(Ks + o), kl = *(const f16x8*)(Ks + NKEY * D + o);
                sx = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, ql[kk], sx, 0, 0, 0);
                sm = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, qh[kk], sm, 0, 0, 0);
                sx = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl, qh[kk], sx, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) s[kt][r] = fmaf(sx[r], SPLIT_LO_INV, sm[r]);
            if ((kt & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = (kt * 16 + g * 4 + r < T) ? s[kt][r] * (scale * LOG2E) : -INFINITY;
                s[kt][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float e = exp2f(s[kt][r] - mx); s[kt][r] = e; sum += e; }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.f / sum;
        f16x8 ph[NS], pl_[NS];
#pragma unroll
        for (int ss = 0; ss < NS; ++ss)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                _Float16 h0, l0, h1 = (_Float16)0.f, l1 = (_Float16)0.f;
                split_f32_raw(s[2 * ss][r] * inv, h0, l0);
                if (2 * ss + 1 < NT) split_f32_raw(s[(2 * ss + 1 < NT) ? 2 * ss + 1 : 0][r] * inv, h1, l1);
                ph[ss][r] = h0; pl_[ss][r] = l0; ph[ss][4 + r] = h1; pl_[ss][4 + r] = l1;
            }
        f32x4 om[DT], ox[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) { om[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; ox[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ss = 0; ss < NS; ++ss) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int vo = (dt * 16 + fr) * VP + (((dt * 16 + fr) >> 3) & 7) * 8 + ss * 32 + g * 4;
                const bool two = 2 * ss + 1 < NT;
                const f16x4 a0 = *(const f16x4*)(Vt + vo), a1 = two ? *(const f16x4*)(Vt + vo + 16) : f16x4{0, 0, 0, 0};
                const f16x4 b0 = *(const f16x4*)(Vt + D * VP + vo), b1 = two ? *(const f16x4*)(Vt + D * VP + vo + 16) : f16x4{0, 0, 0, 0};
                const f16x8 vh = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                const f16x8 vl = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                ox[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, pl_[ss], ox[dt], 0, 0, 0);
                om[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, ph[ss], om[dt], 0, 0, 0);
                ox[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, ph[ss], ox[dt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (qrow < T) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                if (dt * 16 + g * 4 >= dm) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaf(ox[dt][r], SPLIT_LO_INV, om[dt][r]);
                store_h_rt<4>(ctx, ((long)b * T + qrow) * H + head * dm + dt * 16 + g * 4, ctx_lo, v);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// fp32-class attention for head dims 72 .. 128 (XLS-R-2B, the reference's own embedder: 1920 / 16 heads = 120): K and V^T of a
// whole clip in both planes do not fit 160 KB at D = 128, so the keys stream through LDS in blocks of NTB * 16 with an online
// softmax (running row maximum m and sum l in the log2 domain; O rescaled by 2^(m_old - m_new) when a block raises the
// maximum).  Eight wavefronts = eight query tiles per round; a 4 s clip (13 tiles) takes two rounds, each re-staging the
// blocks (the price of keeping one query tile's state per wavefront: 64 accumulator registers for O at D = 128).
template <int NTB, int D>
__global__ __launch_bounds__(512, 1) void attention_x3_stream_kernel(const _Float16* __restrict__ qkv, long qkv_lo, _Float16* __restrict__ ctx,
                                                                     long ctx_lo, int T, int H, int dm, float scale) {
    constexpr int KB = NTB * 16, CH = D / 8, VP = KB + 64, NS = (NTB + 1) / 2, KK = D / 32, DT = D / 16;
    extern __shared__ __attribute__((aligned(16))) _Float16 att_lds[];
    _Float16* Ks = att_lds;                      // [2][KB * D]
    _Float16* Vt = att_lds + 2 * KB * D;         // [2][D * VP]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int head = blockIdx.x, b = blockIdx.y;
    const long ld = 3L * H;
    const _Float16* base = qkv + (long)b * T * ld + head * dm;
    const int chm = dm / 8;
    const int fr = lane & 15, g = lane >> 4;
    const int nqt = (T + 15) / 16, nblk = (T + KB - 1) / KB;
    constexpr int NIT = (KB * CH + 511) / 512;

    for (int q0 = 0; q0 < nqt; q0 += 8) {
        const int qt = q0 + wv;
        const bool live = qt < nqt;                  // wave-uniform; idle wavefronts still take part in staging and barriers
        const int qrow = qt * 16 + fr;
        const int qr = qrow < T ? qrow : T - 1;
        f16x8 qh[KK], ql[KK];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const bool in = live && kk * 4 + g < chm;
            qh[kk] = in ? *(const f16x8*)(base + (long)qr * ld + kk * 32 + g * 8) : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            ql[kk] = in ? *(const f16x8*)(base + qkv_lo + (long)qr * ld + kk * 32 + g * 8) : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
        float m_run = -INFINITY, l_run = 0.f;
        f32x4 om[DT], ox[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) { om[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; ox[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

        for (int blk = 0; blk < nblk; ++blk) {
            const int k0 = blk * KB;
            __syncthreads();                         // every wavefront is done with the previous block
#pragma unroll 1
            for (int pl = 0; pl < 4; ++pl) {         // K hi, K lo, V hi, V lo: one tensor plane at a time (staging registers)
                const bool isv = pl >= 2;
                const _Float16* bp = base + ((pl & 1) ? qkv_lo : 0) + (isv ? 2 * H : H);
                f16x8 reg[NIT];
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int i = tid + it * 512, key = i / CH, c = i % CH;
                    reg[it] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
                    if (i < KB * CH && k0 + key < T && c < chm) reg[it] = *(const f16x8*)(bp + (long)(k0 + key) * ld + c * 8);
                }
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int i = tid + it * 512, key = i / CH, c = i % CH;
                    if (i >= KB * CH) break;
                    if (!isv) *(f16x8*)(Ks + (pl & 1) * KB * D + key * D + ((c ^ (key & (CH - 1))) * 8)) = reg[it];
                    else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) Vt[(pl & 1) * D * VP + (c * 8 + j) * VP + (c & 7) * 8 + key] = reg[it][j];
                    }
                }
            }
            __syncthreads();
            if (!live) continue;
            f32x4 s[NTB];
#pragma unroll
            for (int kt = 0; kt < NTB; ++kt) {
                f32x4 sm = f32x4{0.f, 0.f, 0.f, 0.f}, sx = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) {
                    const int key = kt * 16 + fr, c = kk * 4 + g;
                    const int o = key * D + ((c ^ (key & (CH - 1))) * 8);
                    const f16x8 kh = *(const f16x8*)(Ks + o), kl = *(const f16x8*)(Ks + KB * D + o);
                    sx = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, ql[kk], sx, 0, 0, 0);
                    sm = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, qh[kk], sm, 0, 0, 0);
                    sx = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl, qh[kk], sx, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) s[kt][r] = fmaf(sx[r], SPLIT_LO_INV, sm[r]);
                if ((kt & 1) == 1) __builtin_amdgcn_sched_barrier(0);
            }
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < NTB; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = (k0 + kt * 16 + g * 4 + r < T) ? s[kt][r] * (scale * LOG2E) : -INFINITY;
                    s[kt][r] = v;
                    mx = fmaxf(mx, v);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);     // finite: every block holds at least one key < T
            const float alpha = exp2f(m_run - m_new); // 0 for the first block (m_run = -inf)
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < NTB; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float e = exp2f(s[kt][r] - m_new); s[kt][r] = e; sum += e; }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            l_run = l_run * alpha + sum;
            m_run = m_new;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { om[dt][r] *= alpha; ox[dt][r] *= alpha; }
            // un-normalised probabilities (<= 1) as the split B operand, built per 32-key step right before its MFMAs (building all
            // NS pairs up front kept 32 more registers live and spilled 216 B per lane at D = 128: profiles/r03_xlsr2b_kernel_summary.txt);
            // the 1 / l normalisation happens once at the end
#pragma unroll
            for (int ss = 0; ss < NS; ++ss) {
                f16x8 ph_, pl_s;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    _Float16 h0, l0, h1 = (_Float16)0.f, l1 = (_Float16)0.f;
                    split_f32_raw(s[2 * ss][r], h0, l0);
                    if (2 * ss + 1 < NTB) split_f32_raw(s[(2 * ss + 1 < NTB) ? 2 * ss + 1 : 0][r], h1, l1);
                    ph_[r] = h0; pl_s[r] = l0; ph_[4 + r] = h1; pl_s[4 + r] = l1;
                }
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const int vo = (dt * 16 + fr) * VP + (((dt * 16 + fr) >> 3) & 7) * 8 + ss * 32 + g * 4;
                    const bool two = 2 * ss + 1 < NTB;
                    const f16x4 a0 = *(const f16x4*)(Vt + vo), a1 = two ? *(const f16x4*)(Vt + vo + 16) : f16x4{0, 0, 0, 0};
                    const f16x4 b0 = *(const f16x4*)(Vt + D * VP + vo), b1 = two ? *(const f16x4*)(Vt + D * VP + vo + 16) : f16x4{0, 0, 0, 0};
                    const f16x8 vh = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                    const f16x8 vl = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                    ox[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, pl_s, ox[dt], 0, 0, 0);
                    om[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, ph_, om[dt], 0, 0, 0);
                    ox[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, ph_, ox[dt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (live && qrow < T) {
            const float inv = 1.f / l_run;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                if (dt * 16 + g * 4 >= dm) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaf(ox[dt][r], SPLIT_LO_INV, om[dt][r]) * inv;
                store_h_rt<4>(ctx, ((long)b * T + qrow) * H + head * dm + dt * 16 + g * 4, ctx_lo, v);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// fp32-class attention, head dim 64 (wav2vec2-base / -large): the split-format counterpart of attention_tr_kernel.  K and V
// of both planes are staged ROW-major by the LDS DMA (no register round trip, no scalar transposing stores, no V^T buffer:
// attention_x3_kernel spent 35 % of its LDS cycles on bank conflicts of exactly those) and the V^T operand of O^T = V^T P^T is
// read with ds_read_b64_tr_b16 from either plane.  106 KB of LDS at T = 199; eight wavefronts.
template <int NT>
__global__ __launch_bounds__(512, 1) void attention_x3_tr_kernel(const _Float16* __restrict__ qkv, long qkv_lo, _Float16* __restrict__ ctx,
                                                                 long ctx_lo, int T, int H, float scale) {
    constexpr int D = 64, NKEY = NT * 16, CH = 8, NS = (NT + 1) / 2, KK = 2, DT = 4;
    constexpr int PLB = NKEY * D * 2;            // bytes of one plane of K (or V)
    extern __shared__ __attribute__((aligned(16))) _Float16 att_lds[];
    char* Ks = (char*)att_lds;                   // [2 planes][NKEY][64] fp16, chunk c of key r at slot c ^ (r & 7)
    char* Vs = Ks + 2 * PLB;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int head = blockIdx.x, b = blockIdx.y;
    const long ld = 3L * H;
    const _Float16* base = qkv + (long)b * T * ld + head * D;
    for (int i = tid; i < NKEY * CH; i += 512) {          // keys >= T re-read key T-1 (finite; masked by the softmax / multiplied by P = 0)
        const int key = i / CH, c = (i % CH) ^ (key & 7);
        const _Float16* src = base + (long)min(key, T - 1) * ld + c * 8;
        const size_t dst = (size_t)(i - lane) * 16;
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src + H), LDS_PTR(Ks + dst), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src + qkv_lo + H), LDS_PTR(Ks + PLB + dst), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src + 2 * H), LDS_PTR(Vs + dst), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src + qkv_lo + 2 * H), LDS_PTR(Vs + PLB + dst), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int fr = lane & 15, g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const auto* vs3 = (const __attribute__((address_space(3))) char*)LDS_PTR(Vs);
    for (int qt = wv; qt * 16 < T; qt += 8) {
        const int qrow = qt * 16 + fr;
        const int qr = qrow < T ? qrow : T - 1;
        f16x8 qh[KK], ql[KK];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            qh[kk] = *(const f16x8*)(base + (long)qr * ld + kk * 32 + g * 8);
            ql[kk] = *(const f16x8*)(base + qkv_lo + (long)qr * ld + kk * 32 + g * 8);
        }
        f32x4 s[NT];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            f32x4 sm = f32x4{0.f, 0.f, 0.f, 0.f}, sx = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                const int key = kt * 16 + fr, c = kk * 4 + g;
                const int o = (key * CH + (c ^ (key & 7))) * 16;
                const f16x8 kh = *(const f16x8*)(Ks + o), kl = *(const f16x8*)(Ks + PLB + o);
                sx = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, ql[kk], sx, 0, 0, 0);
                sm = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, qh[kk], sm, 0, 0, 0);
                sx = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl, qh[kk], sx, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) s[kt][r] = fmaf(sx[r], SPLIT_LO_INV, sm[r]);
            if ((kt & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = (kt * 16 + g * 4 + r < T) ? s[kt][r] * (scale * LOG2E) : -INFINITY;
                s[kt][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float e = exp2f(s[kt][r] - mx); s[kt][r] = e; sum += e; }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.f / sum;
        f16x8 ph[NS], pl_[NS];
#pragma unroll
        for (int ss = 0; ss < NS; ++ss)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                _Float16 h0, l0, h1 = (_Float16)0.f, l1 = (_Float16)0.f;
                split_f32_raw(s[2 * ss][r] * inv, h0, l0);
                if (2 * ss + 1 < NT) split_f32_raw(s[(2 * ss + 1 < NT) ? 2 * ss + 1 : 0][r] * inv, h1, l1);
                ph[ss][r] = h0; pl_[ss][r] = l0; ph[ss][4 + r] = h1; pl_[ss][4 + r] = l1;
            }
        f32x4 om[DT], ox[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) { om[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; ox[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ss = 0; ss < NS; ++ss) {
            // keys 32 ss + 4g + q (elements 0-3) and 32 ss + 16 + 4g + q (elements 4-7); a missing odd tile re-reads the even one (P = 0)
            const int ra = 32 * ss + 4 * g + q, rb = (2 * ss + 1 < NT) ? ra + 16 : ra;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int chn = dt * 2 + (pp >> 1), sub = (pp & 1) * 8;
                const size_t oa = ((size_t)ra * CH + (chn ^ (ra & 7))) * 16 + sub, ob = ((size_t)rb * CH + (chn ^ (rb & 7))) * 16 + sub;
                trvec h0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) trvec*)(vs3 + oa));
                trvec h1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) trvec*)(vs3 + ob));
                trvec l0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) trvec*)(vs3 + PLB + oa));
                trvec l1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) trvec*)(vs3 + PLB + ob));
                f16x8 vh, vl;
                __builtin_memcpy(&vh, &h0, 8);
                __builtin_memcpy((char*)&vh + 8, &h1, 8);
                __builtin_memcpy(&vl, &l0, 8);
                __builtin_memcpy((char*)&vl + 8, &l1, 8);
                ox[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, pl_[ss], ox[dt], 0, 0, 0);
                om[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, ph[ss], om[dt], 0, 0, 0);
                ox[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, ph[ss], ox[dt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (qrow < T) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaf(ox[dt][r], SPLIT_LO_INV, om[dt][r]);
                store_h_rt<4>(ctx, ((long)b * T + qrow) * H + head * D + dt * 16 + g * 4, ctx_lo, v);
            }
        }
    }
}

}  // namespace advh

using namespace advh;

extern "C" int advh_attention_f16(const void* qkv, void* ctx, int B, int T, int H, int heads, advh_stream_t stream) {
    if (!qkv || !ctx || B <= 0 || T <= 0 || heads <= 0 || H % heads) return ADVH_EINVAL;
    const int dm = H / heads;
    if (T > 256 || dm % 8 || dm > 128) return ADVH_EUNSUPPORTED;
    const int D = dm <= 32 ? 32 : (dm <= 64 ? 64 : 128);
    const float scale = 1.f / sqrtf((float)dm);
    dim3 grid(heads, B), block(256);
    hipStream_t s = (hipStream_t)stream;
    const int nt = (T + 15) / 16;
#define ATT(NT_, D_) hipLaunchKernelGGL((attention_kernel<NT_, D_>), grid, block, 0, s, (const _Float16*)qkv, (_Float16*)ctx, T, H, dm, scale)
#define ATT_TR(NT_) hipLaunchKernelGGL((attention_tr_kernel<NT_>), grid, block, 0, s, (const _Float16*)qkv, (_Float16*)ctx, T, H, scale)
    if (dm == 64) {                                        // row-major staging by DMA + transposing V^T reads
        if (nt <= 4) ATT_TR(4); else if (nt <= 8) ATT_TR(8); else if (nt <= 13) ATT_TR(13); else ATT_TR(16);
    } else if (D == 64) {
        if (nt <= 4) ATT(4, 64); else if (nt <= 8) ATT(8, 64); else if (nt <= 13) ATT(13, 64); else ATT(16, 64);
    } else if (D == 32) {
        if (nt <= 4) ATT(4, 32); else if (nt <= 8) ATT(8, 32); else if (nt <= 13) ATT(13, 32); else ATT(16, 32);
    } else {
        if (nt <= 4) ATT(4, 128); else if (nt <= 13) ATT(13, 128); else ATT(16, 128);
    }
#undef ATT
#undef ATT_TR
    return ADVH_LAUNCH_CHECK();
}

static size_t att_x3_lds(int nt, int D) { return (size_t)2 * (nt * 16 * D + D * (nt * 16 + 64)) * 2; }

// Raise the dynamic-LDS limit of the split attention instances (advh_init).
int advh_init_attention() {
#define X3A(NT_, D_)                                                                                                                \
    if (hipFuncSetAttribute((const void*)attention_x3_kernel<NT_, D_>, hipFuncAttributeMaxDynamicSharedMemorySize,                  \
                            (int)att_x3_lds(NT_, D_)) != hipSuccess) return ADVH_ELAUNCH;
    X3A(4, 32) X3A(8, 32) X3A(13, 32) X3A(16, 32) X3A(4, 64) X3A(8, 64) X3A(13, 64) X3A(16, 64)
#undef X3A
#define X3T(NT_)                                                                                                                    \
    if (hipFuncSetAttribute((const void*)attention_x3_tr_kernel<NT_>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * NT_ * 16 * 64 * 2) != hipSuccess) \
        return ADVH_ELAUNCH;
    X3T(4) X3T(8) X3T(13) X3T(16)
#undef X3T
    if (hipFuncSetAttribute((const void*)attention_x3_stream_kernel<7, 128>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)att_x3_lds(7, 128)) != hipSuccess) return ADVH_ELAUNCH;
    return ADVH_OK;
}

extern "C" int advh_attention_split(const void* qkv, int64_t qkv_lo, void* ctx, int64_t ctx_lo, int B, int T, int H, int heads,
                                    advh_stream_t stream) {
    if (!qkv || !ctx || B <= 0 || T <= 0 || heads <= 0 || H % heads || qkv_lo <= 0 || ctx_lo <= 0 || qkv_lo % 8 || ctx_lo % 4) return ADVH_EINVAL;
    const int dm = H / heads;
    if (T > 256 || dm % 8 || dm > 128) return ADVH_EUNSUPPORTED;
    if (dm > 64) {                                       // keys streamed in blocks of 112 with an online softmax
        hipLaunchKernelGGL((attention_x3_stream_kernel<7, 128>), dim3(heads, B), dim3(512), att_x3_lds(7, 128), (hipStream_t)stream,
                           (const _Float16*)qkv, (long)qkv_lo, (_Float16*)ctx, (long)ctx_lo, T, H, dm, 1.f / sqrtf((float)dm));
        return ADVH_LAUNCH_CHECK();
    }
    const int D = dm <= 32 ? 32 : 64;
    const float scale = 1.f / sqrtf((float)dm);
    dim3 grid(heads, B);
    hipStream_t s = (hipStream_t)stream;
    const int nt = (T + 15) / 16;
    if (dm == 64) {                                      // row-major DMA staging + transposing V^T reads
#define ATTXT(NT_) hipLaunchKernelGGL((attention_x3_tr_kernel<NT_>), grid, dim3(512), 4 * NT_ * 16 * 64 * 2, s, (const _Float16*)qkv, (long)qkv_lo, (_Float16*)ctx, (long)ctx_lo, T, H, scale)
        if (nt <= 4) ATTXT(4); else if (nt <= 8) ATTXT(8); else if (nt <= 13) ATTXT(13); else ATTXT(16);
#undef ATTXT
        return ADVH_LAUNCH_CHECK();
    }
#define ATTX(NT_, D_) hipLaunchKernelGGL((attention_x3_kernel<NT_, D_>), grid, dim3(512), att_x3_lds(NT_, D_), s, (const _Float16*)qkv, (long)qkv_lo, (_Float16*)ctx, (long)ctx_lo, T, H, dm, scale)
    if (D == 64) { if (nt <= 4) ATTX(4, 64); else if (nt <= 8) ATTX(8, 64); else if (nt <= 13) ATTX(13, 64); else ATTX(16, 64); }
    else { if (nt <= 4) ATTX(4, 32); else if (nt <= 8) ATTX(8, 32); else if (nt <= 13) ATTX(13, 32); else ATTX(16, 32); }
#undef ATTX
    return ADVH_LAUNCH_CHECK();
}

ADVH_SPLIT_FLAG_SETTER(advh_split_flag_attention)
