// Device math shared by the kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace advh {

// erf by Abramowitz & Stegun 7.1.26 (|abs err| <= 1.5e-7, far below the fp16 rounding of the stored
// activation): 2 transcendentals + ~12 VALU ops instead of libm's branchy erff (~50).
__device__ __forceinline__ float fast_erf(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
    const float r = fmaf(-p * t, e, 1.f);
    return copysignf(r, x);
}

// d/dx GELU(x) = Phi(x) + x * phi(x)
__device__ __forceinline__ float gelu_grad(float x) {
    const float cdf = 0.5f * (1.f + fast_erf(x * 0.70710678118654752440f));
    const float pdf = 0.3989422804014327f * __builtin_amdgcn_exp2f(-0.72134752044448170f * x * x);
    return fmaf(x, pdf, cdf);
}

__device__ __forceinline__ float gelu_fast(float x) { return 0.5f * x * (1.f + fast_erf(x * 0.70710678118654752440f)); }

// ---- split-precision ("x3") operand format of the fp32-class mode --------------------------------------------------
// An fp32 value x travels between kernels as TWO fp16 planes: hi = fp16(x) and lo = fp16((x - hi) * 2^11), so that
// x = hi + lo * 2^-11 to ~2^-22 relative (11 + 11 significand bits).  lo is pre-scaled by 2^11 to stay in fp16's normal
// range (|lo| <= |hi|-ish instead of 2^-11 |hi|); values whose hi would be subnormal (|x| < 2^-14) are carried by lo alone
// (absolute error there <= 2^-25 = 3e-8), so the result does not depend on how the matrix cores treat fp16 subnormals.  A product of two split
// values is x*w = xh*wh + (xh*wl + xl*wh) * 2^-11 (+ xl*wl * 2^-22, dropped: <= 2^-22 relative): three fp16 MFMAs with fp32
// accumulation, the cross terms in their own accumulator.  Every fp16 x fp16 product is exact in fp32.
// RANGE: hi is an fp16, so the format covers |x| <= 65504 (fp16's largest finite value; the reference's fp32 reaches 3.4e38).
// A value beyond that SATURATES (hi = +-65504, lo = the clamped remainder: x up to +-65535.98 is still exact) instead of
// turning into inf / NaN downstream, and raises the library's sticky range flag -- one host-mapped word the device writes
// and the host reads without synchronising (advh_split_overflow; the Python binding turns it into an error at its next call).
// A NaN passes through as NaN planes (visible to any isfinite check downstream); +-inf saturates and raises the flag.
constexpr float SPLIT_LO_SCALE = 2048.f, SPLIT_LO_INV = 1.f / 2048.f, SPLIT_MAX = 65504.f;
static __constant__ int* g_split_flag = nullptr;     // this translation unit's copy of the flag pointer (set by advh_init); constant
                                                     // address space: the load is scalar and never ordered against the epilogue's stores
// unchecked conversion: for values that are in range by construction (softmax probabilities, ...)
__device__ __forceinline__ void split_f32_raw(float x, _Float16& hi, _Float16& lo) {
    _Float16 h = (_Float16)x;
    if (fabsf(x) < 6.103515625e-05f) h = (_Float16)0.f;
    hi = h;
    lo = (_Float16)((x - (float)h) * SPLIT_LO_SCALE);
}
// saturating conversion + sticky flag (the rare path of the vector form below, and scalar call sites)
__device__ __forceinline__ void split_f32(float x, _Float16& hi, _Float16& lo) {
    _Float16 h = (_Float16)__builtin_amdgcn_fmed3f(x, -SPLIT_MAX, SPLIT_MAX);
    if (fabsf(x) < 6.103515625e-05f) h = (_Float16)0.f;
    hi = h;
    lo = (_Float16)__builtin_amdgcn_fmed3f((x - (float)h) * SPLIT_LO_SCALE, -SPLIT_MAX, SPLIT_MAX);
    if (fabsf(x) > SPLIT_MAX) {                      // out of range: one exec-masked store on a path that is never taken in range
        int* f = g_split_flag;
        if (f) *(volatile int*)f = 1;
    }
}
// VW values at once (the epilogues convert 4 or 8 consecutive channels per lane): ONE range test on max|v| in front of the
// unchecked conversions -- ~1 extra VALU instruction per value instead of the 3-4 of a per-value clamp + test (round 3: the
// per-value form cost 5 % of gemm_x3_kernel's launch time) -- and the saturating form only when some lane is out of range.
// (NaN does not win a max: NaN values pass through as NaN planes, unflagged; inf is flagged and saturates.)
template <int VW, typename HV>
__device__ __forceinline__ void split_f32_vec(const float (&v)[VW], HV& hv, HV& lv) {
    float m = fabsf(v[0]);
#pragma unroll
    for (int r = 1; r < VW; ++r) m = fmaxf(m, fabsf(v[r]));
    if (m > SPLIT_MAX) {
#pragma unroll
        for (int r = 0; r < VW; ++r) { _Float16 h, l; split_f32(v[r], h, l); hv[r] = h; lv[r] = l; }
    } else {
#pragma unroll
        for (int r = 0; r < VW; ++r) { _Float16 h, l; split_f32_raw(v[r], h, l); hv[r] = h; lv[r] = l; }
    }
}
// defines this translation unit's setter of g_split_flag (called by advh_init on every device it initialises)
#define ADVH_SPLIT_FLAG_SETTER(name)                                                                                       \
    int name(int* flag) {                                                                                                  \
        return hipMemcpyToSymbol(HIP_SYMBOL(advh::g_split_flag), &flag, sizeof(flag)) == hipSuccess ? ADVH_OK : ADVH_ELAUNCH; \
    }
__device__ __forceinline__ float join_f32(_Float16 hi, _Float16 lo) { return fmaf((float)lo, SPLIT_LO_INV, (float)hi); }

// VW consecutive fp16 elements at base + o: plain fp16 (lo_off == 0) or the hi / lo plane pair (lo plane lo_off elements
// behind).  The row kernels take lo_off as a run-time argument (uniform branch; they are HBM-bound).
template <int VW>
__device__ __forceinline__ void store_h_rt(_Float16* base, long o, long lo_off, const float (&v)[VW]) {
    typedef _Float16 hvec __attribute__((ext_vector_type(VW)));
    hvec hv, lv;
    if (lo_off) {
        split_f32_vec<VW>(v, hv, lv);
        *(hvec*)(base + o) = hv;
        *(hvec*)(base + o + lo_off) = lv;
    } else {
#pragma unroll
        for (int r = 0; r < VW; ++r) hv[r] = (_Float16)v[r];
        *(hvec*)(base + o) = hv;
    }
}
template <int VW>
__device__ __forceinline__ void load_h_rt(const _Float16* base, long o, long lo_off, float (&v)[VW]) {
    typedef _Float16 hvec __attribute__((ext_vector_type(VW)));
    const hvec hv = *(const hvec*)(base + o);
    if (lo_off) {
        const hvec lv = *(const hvec*)(base + o + lo_off);
#pragma unroll
        for (int r = 0; r < VW; ++r) v[r] = join_f32(hv[r], lv[r]);
    } else {
#pragma unroll
        for (int r = 0; r < VW; ++r) v[r] = (float)hv[r];
    }
}

}  // namespace advh
