// Issue-rate microbenchmark of the two fp32 MFMA shapes on gfx950: 4 waves per workgroup, 2 workgroups per CU, register
// operands only, NACC independent accumulators per wave.  Prints TFLOP/s per shape.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a0, float b0) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x, b = b0 + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x, b = b0 + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename K>
static void run(const char* name, K kern, double flop_per_mfma, int nacc, int blocks) {
  float* out;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  const int iters = 2000;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 10, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double fl = (double)blocks * 4 /*waves*/ * iters * 4.0 * nacc * flop_per_mfma;
  printf("%-28s blocks %4d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks, ms, fl / ms / 1e9);
  hipFree(out);
}
int main() {
  for (int blocks : {256, 512, 1024}) {
    run("16x16x4  4 acc", k16<4>, 2048.0, 4, blocks);
    run("16x16x4 20 acc", k16<20>, 2048.0, 20, blocks);
    run("32x32x2  4 acc", k32<4>, 4096.0, 4, blocks);
    run("32x32x2  2 acc", k32<2>, 4096.0, 2, blocks);
  }
  return 0;
}
