// Shared helpers for libccvpe_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>

#include "../../include/ccvpe_hip.h"

namespace ccvpe {

extern thread_local char g_err[512];

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CCVPE_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return CCVPE_OK;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// sigmoid / swish on the hardware transcendental path: v_exp_f32 (2^x, ~1 ulp) + v_rcp_f32 (~1 ulp)
// instead of libm expf + IEEE divide (~25 VALU instructions per activation; the expand / depthwise
// epilogues evaluate 10^8-10^9 of them per step).  Relative error ~2e-7: far inside the 1e-3 budget,
// and the parity tests (1e-4 per op) cover it.
__device__ __forceinline__ float sigmoidf(float v) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
}
__device__ __forceinline__ float swishf(float v) { return v * sigmoidf(v); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

}  // namespace ccvpe
