// Does v_mfma_f32_16x16x4_f32 overlap with vector fp32 work of the same SIMD, or do they share the fp32 pipe?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/micro/mfma_f32_valu_overlap.hip -o /tmp/mfo && /tmp/mfo
// Per loop iteration a wave issues 8 independent MFMAs (16x16x4 f32, or 16x16x32 bf16 for comparison) and NV independent v_fma_f32;
// the three timings per configuration (MFMA only, VALU only, both) tell whether the two streams add up.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <bool MF, int NV, bool BF>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  f32x4 acc[8];
  float v[16];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){seed, seed, seed, seed};
  for (int i = 0; i < 16; ++i) v[i] = seed * (i + 1) + threadIdx.x;
  const float a = seed + threadIdx.x, b = seed * 2;
  bf16x8 ab, bb;
  for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)a; bb[i] = (__bf16)b; }
  for (int it = 0; it < iters; ++it) {
    if (MF) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (BF) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[i], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j & 15] = __builtin_fmaf(v[j & 15], 1.0001f, 0.5f);
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 16; ++i) s += v[i];
  if (s == 12345.678f) out[0] = s;
}

template <bool MF, int NV, bool BF>
float run(int blocks, int iters) {
  float* d;
  hipMalloc(&d, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MF, NV, BF>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MF, NV, BF>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipFree(d);
  return ms * 1e3f;
}

template <int NV, bool BF>
void row(int wg_per_cu) {
  const int blocks = 256 * wg_per_cu, iters = 20000;
  const float m = run<true, 0, BF>(blocks, iters), v = run<false, NV, BF>(blocks, iters), b = run<true, NV, BF>(blocks, iters);
  printf("%s, %d wave(s)/SIMD, 8 MFMA + %2d v_fma per iteration: MFMA only %7.0f us (%5.1f cyc/MFMA)  VALU only %7.0f us (%4.1f cyc/instr)  both %7.0f us  -> %s\n",
         BF ? "16x16x32 bf16" : "16x16x4 f32  ", wg_per_cu, NV, m, m * 2400.f / iters / 8 / wg_per_cu, v, v * 2400.f / iters / NV / wg_per_cu, b,
         b > 0.9f * (m + v) ? "ADD UP (shared pipe)" : (b < 1.1f * (m > v ? m : v) ? "overlap" : "partial overlap"));
}

int main() {
  for (int w = 1; w <= 2; ++w) {
    row<16, false>(w); row<32, false>(w); row<64, false>(w);
    row<16, true>(w); row<32, true>(w);
  }
  return 0;
}
