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

typedef float cc_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 cc_bf16;
typedef __bf16 cc_bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 cc_bf16x8 __attribute__((ext_vector_type(8)));

// 4 consecutive channels <-> fp32 registers, for fp32 (16 B) or bf16 (8 B, RNE on store) storage
template <typename T> __device__ __forceinline__ cc_f32x4 ld4(const T* p);
template <> __device__ __forceinline__ cc_f32x4 ld4<float>(const float* p) { return *reinterpret_cast<const cc_f32x4*>(p); }
template <> __device__ __forceinline__ cc_f32x4 ld4<cc_bf16>(const cc_bf16* p) {
  const cc_bf16x4 v = *reinterpret_cast<const cc_bf16x4*>(p);
  return (cc_f32x4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
template <typename T> __device__ __forceinline__ void st4(T* p, cc_f32x4 v);
template <> __device__ __forceinline__ void st4<float>(float* p, cc_f32x4 v) { *reinterpret_cast<cc_f32x4*>(p) = v; }
template <> __device__ __forceinline__ void st4<cc_bf16>(cc_bf16* p, cc_f32x4 v) {
  cc_bf16x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = (cc_bf16)v[i];
  *reinterpret_cast<cc_bf16x4*>(p) = o;
}

// out[e] = sum_k part[k*stride + e]  (k < nparts, e < n): LN lanes per element walk the partial rows interleaved and
// are combined through LDS in lane order -> fixed summation order (deterministic), and short serial chains even
// when there are thousands of partial rows for a handful of elements (BatchNorm / bias / weight-gradient merges).
template <int LN>
__global__ __launch_bounds__(256) void sum_parts_lanes_kernel(const float* __restrict__ part, int nparts, long stride,
                                                              int n, float* __restrict__ out) {
  constexpr int EL = 256 / LN;
  __shared__ float red[256];
  const int el = threadIdx.x % EL, ln = threadIdx.x / EL;
  const int e = blockIdx.x * EL + el;
  float a0 = 0.f, a1 = 0.f;
  if (e < n) {
    int k = ln;
    for (; k + LN < nparts; k += 2 * LN) {
      a0 += part[(size_t)k * stride + e];
      a1 += part[(size_t)(k + LN) * stride + e];
    }
    if (k < nparts) a0 += part[(size_t)k * stride + e];
  }
  if (LN == 1) {
    if (e < n) out[e] = a0 + a1;
    return;
  }
  red[ln * EL + el] = a0 + a1;
  __syncthreads();
  if (ln == 0 && e < n) {
    float s = red[el];
#pragma unroll 4
    for (int q = 1; q < LN; ++q) s += red[q * EL + el];
    out[e] = s;
  }
}

inline void launch_sum_parts(const float* part, int nparts, long stride, int n, float* out, hipStream_t st) {
  // enough workgroups to fill the chip when n is large, enough lanes per element when nparts is large
  int ln = 1;
  while (ln < 64 && nparts > 8 * ln && (long)n * ln < 256L * 1024) ln *= 4;
  if (ln == 1)
    hipLaunchKernelGGL((sum_parts_lanes_kernel<1>), dim3((n + 255) / 256), dim3(256), 0, st, part, nparts, stride, n, out);
  else if (ln == 4)
    hipLaunchKernelGGL((sum_parts_lanes_kernel<4>), dim3((n + 63) / 64), dim3(256), 0, st, part, nparts, stride, n, out);
  else if (ln == 16)
    hipLaunchKernelGGL((sum_parts_lanes_kernel<16>), dim3((n + 15) / 16), dim3(256), 0, st, part, nparts, stride, n, out);
  else
    hipLaunchKernelGGL((sum_parts_lanes_kernel<64>), dim3((n + 3) / 4), dim3(256), 0, st, part, nparts, stride, n, out);
}

}  // namespace ccvpe
