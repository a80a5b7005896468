// fp32-class attention backward: dqkv from (qkv, dctx) for softmax(Q K^T / sqrt(d)) V, T <= 256, head dim <= 128.
//
// The reference differentiates the embedder with fp32 autograd (captum_saliency.py:116-135 through
// transformers/models/wav2vec2/modeling_wav2vec2.py:438-463; loss_function.py:46-53 in the training step).  The fp16
// kernel of backward.hip rounds P and dS to 11 bits; here every product runs on the fp32-input matrix instruction
// v_mfma_f32_16x16x4_f32 (157 TFLOP/s dense, 1/16 of the fp16 rate): attention backward is ~3 % of the chain's FLOPs, so
// the native-fp32 rate costs less than the split-format (three fp16 MFMAs + a split of P and dS per use) would save.
// Operands arrive and leave in the split format of device_math.h (hi + lo fp16 planes, ~22 bits), like every other tensor of
// the fp32-class mode; they are joined to fp32 once, when staged into LDS or fetched into registers.
//
// One workgroup per (head, clip), four wavefronts, two passes as in the fp16 kernel (S and dP are recomputed in both
// orientations so that every product sums over the accumulator's ROW index: no atomics, no cross-lane shuffles):
//   A  (wavefront = 16-query tile, scores transposed [key][q]): row max / 1/sum / delta, dQ^T = K^T dS^T
//   B  (wavefront = 16-key tile, scores [q][key]):              dK^T = Q^T dS, dV^T = dO^T P
// MFMA operand layout (16x16x4, fp32): lane l holds A[l % 16][l / 16] and B[l / 16][l % 16]; the accumulator's lane holds
// rows 4 (l / 16) + r, column l % 16.  The contraction index of a product may be visited in any order as long as both
// operands agree, which is used twice:
//   * "row" operands (contraction over d): the four MFMAs of a 16-wide d group take d = 16 G + 4 (l / 16) + i, i = 0..3,
//     so a lane fetches ONE float4 (LDS: ds_read_b128, pitch D + 4 floats => conflict-free) per four MFMAs;
//   * "transposed" operands (contraction over keys / queries): MFMA (tile, r) takes k = 16 tile + 4 (l / 16) + r, which is
//     exactly the accumulator register r of the score tile -- dS / P go from registers straight into the B operand -- and the
//     A operand is a ds_read_b32 of M[k][d0 + l % 16] (conflict-free for the same pitch).
// LDS holds K (pass A) / Q (pass B) as fp32 and, when two matrices fit (TWO: head dim <= 64), V / dO as well; otherwise
// the second matrix's row operands come straight from global memory and pass B runs in two sub-passes (Q staged: dK; dO staged: dV).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include "addvisor_hip.h"
#include "common.h"
#include "device_math.h"

namespace advh {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr float LOG2E_F = 1.4426950408889634f;

#define MFMA4(acc, a4, b4)                                                        \
    do {                                                                          \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32((a4).x, (b4).x, acc, 0, 0, 0); \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32((a4).y, (b4).y, acc, 0, 0, 0); \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32((a4).z, (b4).z, acc, 0, 0, 0); \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32((a4).w, (b4).w, acc, 0, 0, 0); \
    } while (0)

// four consecutive channels d .. d+3 of one row of a split-format matrix (zeros past the real head dim)
__device__ __forceinline__ float4 grow4(const _Float16* base, long lo, long row_off, int d, int dm) {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (d < dm) load_h_rt<4>(base, row_off + d, lo, v);
    return make_float4(v[0], v[1], v[2], v[3]);
}

// rows [0, T) x channels [0, dm) of a split-format matrix (row stride ld) -> fp32 LDS tile [NKEY][D + 4], zero elsewhere
template <int NKEY, int D, int NTH>
__device__ __forceinline__ void stage_f32(float* dst, const _Float16* src, long lo, long ld, int T, int dm, int tid) {
    constexpr int PITCH = D + 4, CH = D / 8;
    for (int i = tid; i < NKEY * CH; i += NTH) {
        const int row = i / CH, c = i % CH;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (row < T && c * 8 < dm) load_h_rt<8>(src, (long)row * ld + c * 8, lo, v);
        *(float4*)(dst + row * PITCH + c * 8) = make_float4(v[0], v[1], v[2], v[3]);
        *(float4*)(dst + row * PITCH + c * 8 + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
}

template <int NT, int D>
struct AttBwdF32 {
    static constexpr int NKEY = NT * 16, PITCH = D + 4, DG = D / 16;
    static constexpr int MAT = NKEY * PITCH;                                         // floats of one staged matrix
    static constexpr bool TWO = (2 * MAT + 3 * NKEY) * 4 <= 160 * 1024;
    static constexpr int LDS_BYTES = ((TWO ? 2 : 1) * MAT + 3 * NKEY) * 4;
};

// NW wavefronts per workgroup: 8 (two per SIMD, <= 256 VGPRs each) where the register budget allows -- head dims <= 64 --, so that
// one wavefront's LDS / global latency and MFMA dependency stalls are covered by the other's issue (one workgroup per CU: the
// staged matrices take most of the LDS); 4 for head dim 128.
template <int NT, int D, int NW>
__global__ __launch_bounds__(64 * NW) void attention_bwd_f32_kernel(const _Float16* __restrict__ qkv, long qkv_lo, const _Float16* __restrict__ dctx,
                                                                long dctx_lo, _Float16* __restrict__ dqkv, long dqkv_lo, int T, int H, int dm,
                                                                float scale) {
    typedef AttBwdF32<NT, D> G;
    constexpr int NKEY = G::NKEY, PITCH = G::PITCH, DG = G::DG;
    constexpr bool TWO = G::TWO;
    extern __shared__ __attribute__((aligned(16))) float smf[];
    float* M0 = smf;                                      // K (pass A), Q (pass B1), dO (pass B2)
    float* M1 = smf + G::MAT;                             // TWO: V (pass A), dO (pass B)
    float* rmax = smf + (TWO ? 2 : 1) * G::MAT;
    float* rinv = rmax + NKEY;
    float* rdel = rinv + NKEY;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int fr = lane & 15, g = lane >> 4;
    const int head = blockIdx.x, b = blockIdx.y;
    const long ld = 3L * H;
    const _Float16* base = qkv + (long)b * T * ld + head * dm;           // q at +0, k at +H, v at +2H
    const _Float16* dob = dctx + (long)b * T * H + head * dm;
    _Float16* dbase = dqkv + (long)b * T * ld + head * dm;
    const float c2 = scale * LOG2E_F;

    stage_f32<NKEY, D, 64 * NW>(M0, base + H, qkv_lo, ld, T, dm, tid);
    if (TWO) stage_f32<NKEY, D, 64 * NW>(M1, base + 2 * H, qkv_lo, ld, T, dm, tid);
    __syncthreads();

    // ------------------------------------------------------------------ pass A: query tiles
    for (int qt = wv; qt * 16 < T; qt += NW) {
        const int qrow = qt * 16 + fr, qr = qrow < T ? qrow : T - 1;
        float4 qf[DG], of[DG];
#pragma unroll
        for (int G_ = 0; G_ < DG; ++G_) {
            qf[G_] = grow4(base, qkv_lo, (long)qr * ld, 16 * G_ + 4 * g, dm);
            of[G_] = grow4(dob, dctx_lo, (long)qr * H, 16 * G_ + 4 * g, dm);
        }
        f32x4 s[NT], dp[NT];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
            dp[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int key = kt * 16 + fr, keyc = key < T ? key : T - 1;
#pragma unroll
            for (int G_ = 0; G_ < DG; ++G_) {
                const float4 kf = *(const float4*)(M0 + key * PITCH + 16 * G_ + 4 * g);
                float4 vf;
                if (TWO) vf = *(const float4*)(M1 + key * PITCH + 16 * G_ + 4 * g);
                else vf = grow4(base + 2 * H, qkv_lo, (long)keyc * ld, 16 * G_ + 4 * g, dm);
                MFMA4(s[kt], kf, qf[G_]);                 // S^T  [key][q]
                MFMA4(dp[kt], vf, of[G_]);                // dP^T [key][q]
            }
            __builtin_amdgcn_sched_barrier(0);
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
            for (int r = 0; r < 4; ++r) { const float e = exp2f(s[kt][r] - mx); s[kt][r] = e; sum += e; }
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
        if (g == 0 && qrow < NKEY) { rmax[qrow] = mx; rinv[qrow] = inv; rdel[qrow] = del; }
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[kt][r] = s[kt][r] * (dp[kt][r] - del) * scale;     // dS^T [key][q]; 0 for keys >= T (p = 0)
        f32x4 o[DG];
#pragma unroll
        for (int dt = 0; dt < DG; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* kr = M0 + (kt * 16 + 4 * g + r) * PITCH + fr;
#pragma unroll
                for (int dt = 0; dt < DG; ++dt)
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kr[16 * dt], s[kt][r], o[dt], 0, 0, 0);   // dQ^T [d][q]
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (qrow < T) {
#pragma unroll
            for (int dt = 0; dt < DG; ++dt) {
                if (dt * 16 + g * 4 >= dm) continue;
                const float v[4] = {o[dt][0], o[dt][1], o[dt][2], o[dt][3]};
                store_h_rt<4>(dbase, (long)qrow * ld + dt * 16 + g * 4, dqkv_lo, v);
            }
        }
    }
    __syncthreads();

    // ------------------------------------------------------------------ pass B: key tiles
    stage_f32<NKEY, D, 64 * NW>(M0, base, qkv_lo, ld, T, dm, tid);                 // Q
    if (TWO) stage_f32<NKEY, D, 64 * NW>(M1, dob, dctx_lo, (long)H, T, dm, tid);   // dO
    __syncthreads();
    for (int kt = wv; kt * 16 < T; kt += NW) {
        const int krow = kt * 16 + fr, kr_ = krow < T ? krow : T - 1;
        float4 kf[DG], vf[DG];
#pragma unroll
        for (int G_ = 0; G_ < DG; ++G_) {
            kf[G_] = grow4(base + H, qkv_lo, (long)kr_ * ld, 16 * G_ + 4 * g, dm);
            vf[G_] = grow4(base + 2 * H, qkv_lo, (long)kr_ * ld, 16 * G_ + 4 * g, dm);
        }
        f32x4 dkt[DG], dvt[DG];
#pragma unroll
        for (int dt = 0; dt < DG; ++dt) { dkt[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dvt[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        for (int qt = 0; qt * 16 < T; ++qt) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
            const int qrow = qt * 16 + fr, qr = qrow < T ? qrow : T - 1;
#pragma unroll
            for (int G_ = 0; G_ < DG; ++G_) {
                const float4 qa = *(const float4*)(M0 + qrow * PITCH + 16 * G_ + 4 * g);
                float4 oa;
                if (TWO) oa = *(const float4*)(M1 + qrow * PITCH + 16 * G_ + 4 * g);
                else oa = grow4(dob, dctx_lo, (long)qr * H, 16 * G_ + 4 * g, dm);
                MFMA4(s, qa, kf[G_]);                     // S  [q][key]
                MFMA4(dp, oa, vf[G_]);                    // dP [q][key]
            }
            float p[4], ds[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = qt * 16 + g * 4 + r;        // this lane's query rows; its key column = krow
                p[r] = 0.f;
                ds[r] = 0.f;
                if (q < T && krow < T) {
                    p[r] = exp2f(s[r] * c2 - rmax[q]) * rinv[q];
                    ds[r] = p[r] * (dp[r] - rdel[q]) * scale;
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* qrw = M0 + (qt * 16 + 4 * g + r) * PITCH + fr;
                const float* orw = M1 + (qt * 16 + 4 * g + r) * PITCH + fr;
#pragma unroll
                for (int dt = 0; dt < DG; ++dt) {
                    dkt[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qrw[16 * dt], ds[r], dkt[dt], 0, 0, 0);              // dK^T [d][key]
                    if (TWO) dvt[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(orw[16 * dt], p[r], dvt[dt], 0, 0, 0);     // dV^T [d][key]
                }
            }
        }
        if (krow < T) {
#pragma unroll
            for (int dt = 0; dt < DG; ++dt) {
                if (dt * 16 + g * 4 >= dm) continue;
                const float kv[4] = {dkt[dt][0], dkt[dt][1], dkt[dt][2], dkt[dt][3]};
                store_h_rt<4>(dbase, (long)krow * ld + H + dt * 16 + g * 4, dqkv_lo, kv);
                if (TWO) {
                    const float vv[4] = {dvt[dt][0], dvt[dt][1], dvt[dt][2], dvt[dt][3]};
                    store_h_rt<4>(dbase, (long)krow * ld + 2 * H + dt * 16 + g * 4, dqkv_lo, vv);
                }
            }
        }
    }
    if (TWO) return;

    // ------------------------------------------------------------------ pass B2 (one matrix fits): dO staged, dV^T = dO^T P
    __syncthreads();
    stage_f32<NKEY, D, 64 * NW>(M0, dob, dctx_lo, (long)H, T, dm, tid);
    __syncthreads();
    for (int kt = wv; kt * 16 < T; kt += NW) {
        const int krow = kt * 16 + fr, kr_ = krow < T ? krow : T - 1;
        float4 kf[DG];
#pragma unroll
        for (int G_ = 0; G_ < DG; ++G_) kf[G_] = grow4(base + H, qkv_lo, (long)kr_ * ld, 16 * G_ + 4 * g, dm);
        f32x4 dvt[DG];
#pragma unroll
        for (int dt = 0; dt < DG; ++dt) dvt[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int qt = 0; qt * 16 < T; ++qt) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            const int qrow = qt * 16 + fr, qr = qrow < T ? qrow : T - 1;
#pragma unroll
            for (int G_ = 0; G_ < DG; ++G_) {
                const float4 qa = grow4(base, qkv_lo, (long)qr * ld, 16 * G_ + 4 * g, dm);
                MFMA4(s, qa, kf[G_]);
            }
            float p[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = qt * 16 + g * 4 + r;
                p[r] = (q < T && krow < T) ? exp2f(s[r] * c2 - rmax[q]) * rinv[q] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* orw = M0 + (qt * 16 + 4 * g + r) * PITCH + fr;
#pragma unroll
                for (int dt = 0; dt < DG; ++dt) dvt[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(orw[16 * dt], p[r], dvt[dt], 0, 0, 0);
            }
        }
        if (krow < T) {
#pragma unroll
            for (int dt = 0; dt < DG; ++dt) {
                if (dt * 16 + g * 4 >= dm) continue;
                const float vv[4] = {dvt[dt][0], dvt[dt][1], dvt[dt][2], dvt[dt][3]};
                store_h_rt<4>(dbase, (long)krow * ld + 2 * H + dt * 16 + g * 4, dqkv_lo, vv);
            }
        }
    }
}

template <int NT, int D>
static int launch_att_bwd_f32(const void* qkv, long qkv_lo, const void* dctx, long dctx_lo, void* dqkv, long dqkv_lo, int B, int T, int H,
                              int heads, int dm, float scale, hipStream_t s) {
    constexpr int lds = AttBwdF32<NT, D>::LDS_BYTES;
    static_assert(lds <= 160 * 1024, "one staged matrix must fit");
    constexpr int NW = D <= 64 ? 8 : 4;
    if (advh_ensure_lds((const void*)attention_bwd_f32_kernel<NT, D, NW>) != ADVH_OK) return ADVH_ELAUNCH;
    hipLaunchKernelGGL((attention_bwd_f32_kernel<NT, D, NW>), dim3(heads, B), dim3(64 * NW), lds, s, (const _Float16*)qkv, qkv_lo,
                       (const _Float16*)dctx, dctx_lo, (_Float16*)dqkv, dqkv_lo, T, H, dm, scale);
    return ADVH_LAUNCH_CHECK();
}

}  // namespace advh

using namespace advh;

int g_att_bwd_force_f32 = 0;

extern "C" int advh_attention_bwd_split(const void* qkv, int64_t qkv_lo, const void* dctx, int64_t dctx_lo, void* dqkv, int64_t dqkv_lo,
                                        int B, int T, int H, int heads, advh_stream_t stream) {
    if (!qkv || !dctx || !dqkv || B <= 0 || T <= 0 || heads <= 0 || H % heads) return ADVH_EINVAL;
    if (qkv_lo <= 0 || dctx_lo <= 0 || dqkv_lo <= 0 || qkv_lo % 8 || dctx_lo % 8 || dqkv_lo % 8) return ADVH_EINVAL;
    const int dm = H / heads;
    if (T > 256 || dm % 8 || dm > 128) return ADVH_EUNSUPPORTED;
    if (dm <= 64 && !g_att_bwd_force_f32)           // three fp16 MFMAs per product instead of eight fp32 MFMAs (attention_bwd_x3.hip)
        return advh_attention_bwd_x3_launch(qkv, qkv_lo, dctx, dctx_lo, dqkv, dqkv_lo, B, T, H, heads, (hipStream_t)stream);
    const int D = dm <= 32 ? 32 : (dm <= 64 ? 64 : 128);
    const float scale = 1.f / sqrtf((float)dm);
    hipStream_t s = (hipStream_t)stream;
    const int nt = (T + 15) / 16;
#define ATB(NT_, D_) return launch_att_bwd_f32<NT_, D_>(qkv, qkv_lo, dctx, dctx_lo, dqkv, dqkv_lo, B, T, H, heads, dm, scale, s)
#define ATB_D(D_)                                                                     \
    do {                                                                              \
        if (nt <= 4) ATB(4, D_); else if (nt <= 8) ATB(8, D_); else if (nt <= 13) ATB(13, D_); else ATB(16, D_); \
    } while (0)
    if (D == 32) ATB_D(32);
    else if (D == 64) ATB_D(64);
    else ATB_D(128);
#undef ATB_D
#undef ATB
    return ADVH_EUNSUPPORTED;
}

ADVH_SPLIT_FLAG_SETTER(advh_split_flag_attention_bwd_f32)
