// LMAC faithfulness metrics (LMAC_metrics.py:31-73): per-clip values in fp32 exactly as the reference
// formulas, dataset sums in fp64 in a fixed order (one workgroup, fixed reduction tree), so the result
// does not depend on how the clips were sharded over GPUs.
#include <hip/hip_runtime.h>
#include "addvisor_hip.h"
#include "common.h"

namespace advh {

__device__ __forceinline__ float score_pred_class(float p) {      // LMAC_metrics.py:43-45
    float pred = p > 0.5f ? 1.f : 0.f;
    return pred * p + (1.f - pred) * (1.f - p);
}

__global__ __launch_bounds__(256) void lmac_metrics_kernel(const float* __restrict__ p, const float* __restrict__ th,
                                                           const float* __restrict__ po, int n, double* __restrict__ sums,
                                                           float* __restrict__ per_clip) {
    __shared__ double red[5][256];
    const int tid = threadIdx.x;
    double acc[5] = {0, 0, 0, 0, 0};
    for (int i = tid; i < n; i += 256) {
        const float eps = 1e-10f;
        float pi = p[i], ti = th[i], oi = po[i];
        float d = pi - 0.5f;
        float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        float faith = (pi - oi) * sgn;                                        // :48-52
        float fid = ((pi > 0.5f) == (ti > 0.5f)) ? 1.f : 0.f;                  // :31-38
        float pc = score_pred_class(pi), oc = score_pred_class(ti);
        float ad = (fmaxf(pc - oc, 0.f) / (pc + eps)) * 100.f;                 // :55-59
        float ai = oc > pc ? 100.f : 0.f;                                      // :62-66
        float ag = (fmaxf(oc - pc, 0.f) / (1.f - pc + eps)) * 100.f;           // :69-73
        float v[5] = {faith, fid, ad, ai, ag};
#pragma unroll
        for (int k = 0; k < 5; ++k) { acc[k] += v[k]; if (per_clip) per_clip[(long)k * n + i] = v[k]; }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) red[k][tid] = acc[k];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s)
#pragma unroll
            for (int k = 0; k < 5; ++k) red[k][tid] += red[k][tid + s];
        __syncthreads();
    }
    if (tid < 5) sums[tid] = red[tid][0];
    if (tid == 5) sums[5] = (double)n;
}

}  // namespace advh

extern "C" int advh_lmac_metrics_accumulate(const float* predictions, const float* theta_out, const float* masked_predictions,
                                            int n, double* sums6, float* per_clip, advh_stream_t stream) {
    if (!predictions || !theta_out || !masked_predictions || !sums6 || n <= 0) return ADVH_EINVAL;
    hipLaunchKernelGGL(advh::lmac_metrics_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, predictions, theta_out,
                       masked_predictions, n, sums6, per_clip);
    return ADVH_LAUNCH_CHECK();
}
