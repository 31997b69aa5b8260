// Issue cost of the vector instructions the encoders live on (BN + swish on every element of the 6x-expanded tensors), per
// wave64 instruction, with 1 / 2 / 4 waves per SIMD:   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/micro/valu_rates.hip -o /tmp/vr && /tmp/vr
//   fma      v_fma_f32           pk_fma  v_pk_fma_f32 (two lanes' worth of FMAs per instruction)
//   exp      v_exp_f32           rcp     v_rcp_f32
//   swish    the product form of common.h: fma (BN), mul, exp, add, rcp, mul  = one activated value
//   swish_nr the same with the reciprocal replaced by a degree-3 polynomial of 1/d on (1, 2] + one Newton step (FMA only)
// 16 independent chains per thread so that no instruction waits for its predecessor.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { FMA, PKFMA, EXP, RCP, SWISH, SWISH_NR };

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  float v[16];
  f32x2 w[8];
  for (int i = 0; i < 16; ++i) v[i] = seed * 0.01f * (i + 1) + threadIdx.x * 1e-3f;
  for (int i = 0; i < 8; ++i) w[i] = (f32x2){v[2 * i], v[2 * i + 1]};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (OP == FMA) v[j] = __builtin_fmaf(v[j], 1.0001f, 0.5f);
      if (OP == EXP) v[j] = __builtin_amdgcn_exp2f(v[j]);
      if (OP == RCP) v[j] = __builtin_amdgcn_rcpf(v[j]);
      if (OP == SWISH) {
        const float t = __builtin_fmaf(v[j], 0.999f, 0.01f);
        v[j] = t * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * t));
      }
      if (OP == SWISH_NR) {
        const float t = __builtin_fmaf(v[j], 0.999f, 0.01f);
        const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * __builtin_fabsf(t));
        const float d = 1.0f + e;
        float r = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(-0.1229f, d, 0.7353f), d, -1.6512f), d, 1.9877f);   // ~1/d on (1, 2]
        r = r * __builtin_fmaf(-d, r, 2.0f);
        v[j] = __builtin_fmaf(__builtin_fabsf(t), r, __builtin_fminf(t, 0.f));
      }
    }
    if (OP == PKFMA) {
#pragma unroll
      for (int j = 0; j < 8; ++j) w[j] = w[j] * (f32x2){1.0001f, 1.0002f} + (f32x2){0.5f, 0.25f};
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += v[i];
  for (int i = 0; i < 8; ++i) s += w[i][0] + w[i][1];
  if (s == 12345.678f) out[0] = s;
}

template <int OP>
float run(int blocks, int iters) {
  float* d;
  hipMalloc(&d, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipFree(d);
  return ms * 1e3f;
}

int main() {
  const int iters = 20000;
  for (int wpc = 1; wpc <= 4; wpc *= 2) {
    const int blocks = 256 * wpc;
    const float ghz = 2.4f;
    const float f = run<FMA>(blocks, iters), p = run<PKFMA>(blocks, iters), e = run<EXP>(blocks, iters), r = run<RCP>(blocks, iters);
    const float s = run<SWISH>(blocks, iters), n = run<SWISH_NR>(blocks, iters);
    auto cyc = [&](float us, int per_iter) { return us * 1e3f * ghz / iters / per_iter / wpc; };
    printf("%d wave(s)/SIMD  cycles per wave instruction at %.1f GHz: v_fma_f32 %.2f  v_pk_fma_f32 %.2f  v_exp_f32 %.2f  v_rcp_f32 %.2f | "
           "cycles per activated value (64 lanes): swish %.1f  swish with polynomial reciprocal %.1f\n",
           wpc, ghz, cyc(f, 16), cyc(p, 8), cyc(e, 16), cyc(r, 16), cyc(s, 16), cyc(n, 16));
  }
  return 0;
}
