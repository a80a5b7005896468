// Shared host-side helpers of libaddvisor_hip.
#pragma once
#include <hip/hip_runtime.h>
#include "addvisor_hip.h"

// Per-module one-time initialisation (dynamic-LDS attributes); defined in gemm.hip.
int advh_init_rest();
int advh_init_attention();   // attention.hip
// per-translation-unit setters of the split-format range flag pointer (csrc/device_math.h: ADVH_SPLIT_FLAG_SETTER)
int advh_split_flag_attention(int* flag);
int advh_split_flag_attention_bwd_f32(int* flag);
int advh_split_flag_attention_bwd_x3(int* flag);
int advh_split_flag_backward(int* flag);
int advh_split_flag_conv_taps(int* flag);
int advh_split_flag_frontend(int* flag);
int advh_split_flag_frontend_bwd(int* flag);
int advh_split_flag_gemm(int* flag);
int advh_split_flag_hifigan(int* flag);
int advh_split_flag_rowops(int* flag);
int advh_split_flag_unet_misc(int* flag);
int advh_split_flag_unet_train(int* flag);
int advh_split_flag_resblock_pair_x3(int* flag);

// attention backward of the fp32-class mode: the split-arithmetic kernel for head dims <= 64 (attention_bwd_x3.hip), called by
// advh_attention_bwd_split (attention_bwd_f32.hip) after it validated the arguments; g_att_bwd_force_f32: advh_set_option("attention_bwd_mfma_f32")
int advh_attention_bwd_x3_launch(const void* qkv, long qkv_lo, const void* dctx, long dctx_lo, void* dqkv, long dqkv_lo, int B, int T, int H,
                                 int heads, hipStream_t s);
extern int g_att_bwd_force_f32;

// Raise a kernel's dynamic-LDS limit to `bytes` once per (kernel, DEVICE): the attribute belongs to the device's copy of
// the code object, and one process may drive several GPUs (a per-process "done" flag left the second device at 64 KiB).
#include <mutex>
#include <set>
#include <utility>
inline int advh_ensure_lds(const void* fn, int bytes = 160 * 1024) {
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return ADVH_ELAUNCH;
    std::lock_guard<std::mutex> lk(mu);
    if (done.count({fn, dev})) return ADVH_OK;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return ADVH_ELAUNCH;
    done.insert({fn, dev});
    return ADVH_OK;
}

#define ADVH_LAUNCH_CHECK() (hipGetLastError() == hipSuccess ? ADVH_OK : ADVH_ELAUNCH)
