// Does v_mfma_f32_32x32x16_bf16 buy anything over v_mfma_f32_16x16x32_bf16 for the wave tile the wide bf16 decoder kernels use?
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -w tools/micro/mfma_bf16_shapes.hip -o /tmp/mbs && /tmp/mbs
// The review asked (three times) for a 32x32x16 re-tiling of conv3x3_kernel<bf16,4,5,2> / upconv_dma_kernel on the grounds of "half
// the fragment bytes per FLOP".  That ratio is per INSTRUCTION; per FLOP of a register-blocked wave tile it is the blocking that
// counts.  This benchmark runs exactly the inner loop of those kernels — operand fragments read from a resident LDS stage with
// ds_read_b128, one K = 32 step per iteration, two waves per SIMD, nothing else — for
//   A: 16x16x32, 4 x 5 blocking (64 pixels x 80 columns per wave: the product kernels' tile): 9 fragment reads, 20 MFMAs per K = 32
//   B: 32x32x16, 2 x 2 blocking (64 x 64): 2 x (2 + 2) = 8 reads, 8 MFMAs per K = 32
//   C: 32x32x16, 2 x 3 blocking (64 x 96): 2 x (2 + 3) = 10 reads, 12 MFMAs per K = 32
//   D: 16x16x32, 4 x 4 blocking (64 x 64): 8 reads, 16 MFMAs per K = 32
// and the same four with the fragment reads removed (register operands): the ceiling of each instruction stream.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MT, int NT, bool LDS>
__global__ __launch_bounds__(512, 1) void k16(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float stage[2][(4 * 64 + 160) * 16];     // [buffer][rows][64 bytes]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * (4 * 64 + 160) * 16; i += 512) (&stage[0][0])[i] = 1e-3f * (i & 255);
  __syncthreads();
  f32x4 acc[MT][NT];
  for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int row = lane & 15, piece = lane >> 4;
  const float* abase = &stage[0][0] + ((wave >> 1) * 64 + row) * 16 + ((piece ^ ((row >> 2) & 1) * 2) * 4);
  const float* bbase = &stage[0][0] + (4 * 64 + (wave & 1) * 80 + row) * 16 + ((piece ^ ((row >> 2) & 1) * 2) * 4);
  f32x4 af[MT], bf[NT];
  for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const f32x4*>(abase + i * 256);
  for (int j = 0; j < NT; ++j) bf[j] = *reinterpret_cast<const f32x4*>(bbase + j * 256);
  for (int it = 0; it < iters; ++it) {
    const int buf = (it & 1) * (4 * 64 + 160) * 16;
    f32x4 an[MT], bn[NT];
    if (LDS) {
#pragma unroll
      for (int i = 0; i < MT; ++i) an[i] = *reinterpret_cast<const f32x4*>(abase + buf + i * 256);
#pragma unroll
      for (int j = 0; j < NT; ++j) bn[j] = *reinterpret_cast<const f32x4*>(bbase + buf + j * 256);
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bf[j]), __builtin_bit_cast(bf16x8, af[i]), acc[i][j], 0, 0, 0);
    if (LDS) {
#pragma unroll
      for (int i = 0; i < MT; ++i) af[i] = an[i];
#pragma unroll
      for (int j = 0; j < NT; ++j) bf[j] = bn[j];
    }
  }
  float s = 0.f;
  for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) s += acc[i][j][0] + acc[i][j][3];
  out[blockIdx.x * 512 + tid] = s;
}

template <int MT, int NT, bool LDS>
__global__ __launch_bounds__(512, 1) void k32(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float stage[2][(4 * 64 + 192) * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * (4 * 64 + 192) * 16; i += 512) (&stage[0][0])[i] = 1e-3f * (i & 255);
  __syncthreads();
  f32x16 acc[MT][NT];
  for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int row = lane & 31, kg = lane >> 5;                     // 32 rows x (2 k groups of 8 bf16) per K = 16 step
  // k step s of the 32-channel stage: 16-byte piece 2 s + kg of the row, swizzled like the product kernels' panels
  const float* abase = &stage[0][0] + ((wave >> 1) * 64 + row) * 16;
  const float* bbase = &stage[0][0] + (4 * 64 + (wave & 1) * 96 + row) * 16;
  const int sw = ((row >> 2) & 1) * 2;
  f32x4 af[2][MT], bf[2][NT];
  for (int s = 0; s < 2; ++s) {
    for (int i = 0; i < MT; ++i) af[s][i] = *reinterpret_cast<const f32x4*>(abase + i * 512 + (((2 * s + kg) ^ sw) * 4));
    for (int j = 0; j < NT; ++j) bf[s][j] = *reinterpret_cast<const f32x4*>(bbase + j * 512 + (((2 * s + kg) ^ sw) * 4));
  }
  for (int it = 0; it < iters; ++it) {
    const int buf = (it & 1) * (4 * 64 + 192) * 16;
    f32x4 an[2][MT], bn[2][NT];
    if (LDS) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int i = 0; i < MT; ++i) an[s][i] = *reinterpret_cast<const f32x4*>(abase + buf + i * 512 + (((2 * s + kg) ^ sw) * 4));
#pragma unroll
        for (int j = 0; j < NT; ++j) bn[s][j] = *reinterpret_cast<const f32x4*>(bbase + buf + j * 512 + (((2 * s + kg) ^ sw) * 4));
      }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bf[s][j]), __builtin_bit_cast(bf16x8, af[s][i]), acc[i][j], 0, 0, 0);
    if (LDS) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int i = 0; i < MT; ++i) af[s][i] = an[s][i];
#pragma unroll
        for (int j = 0; j < NT; ++j) bf[s][j] = bn[s][j];
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) s += acc[i][j][0] + acc[i][j][15];
  out[blockIdx.x * 512 + tid] = s;
}

template <typename K>
static void run(const char* name, K kern, double flop_per_iter_per_wave) {
  const int blocks = 256, iters = 4000;
  float* out;
  hipMalloc(&out, (size_t)blocks * 512 * 4);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, out, 10);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-64s %8.3f ms  %7.0f TFLOP/s\n", name, ms, (double)blocks * 8 * iters * flop_per_iter_per_wave / ms / 1e9);
  hipFree(out);
}

int main() {
  const double f16 = 2.0 * 16 * 16 * 32, f32 = 2.0 * 32 * 32 * 16;
  run("A  16x16x32, 4 x 5 (64 x 80), fragments from LDS", k16<4, 5, true>, 20 * f16);
  run("A' 16x16x32, 4 x 5, register operands", k16<4, 5, false>, 20 * f16);
  run("D  16x16x32, 4 x 4 (64 x 64), fragments from LDS", k16<4, 4, true>, 16 * f16);
  run("D' 16x16x32, 4 x 4, register operands", k16<4, 4, false>, 16 * f16);
  run("B  32x32x16, 2 x 2 (64 x 64), fragments from LDS", k32<2, 2, true>, 8 * f32);
  run("B' 32x32x16, 2 x 2, register operands", k32<2, 2, false>, 8 * f32);
  run("C  32x32x16, 2 x 3 (64 x 96), fragments from LDS", k32<2, 3, true>, 12 * f32);
  run("C' 32x32x16, 2 x 3, register operands", k32<2, 3, false>, 12 * f32);
  return 0;
}
