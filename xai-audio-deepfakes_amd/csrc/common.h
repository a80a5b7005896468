// Shared host-side helpers of libaddvisor_hip.
#pragma once
#include <hip/hip_runtime.h>
#include "addvisor_hip.h"

// Per-module one-time initialisation (dynamic-LDS attributes); defined in gemm.hip.
int advh_init_rest();
int advh_init_attention();   // attention.hip

#define ADVH_LAUNCH_CHECK() (hipGetLastError() == hipSuccess ? ADVH_OK : ADVH_ELAUNCH)
