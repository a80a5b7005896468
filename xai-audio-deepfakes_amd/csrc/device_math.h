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

}  // namespace advh
