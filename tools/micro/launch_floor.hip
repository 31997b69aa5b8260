// How long does a small dependent kernel take back to back in one stream?  The squeeze-excite gate (se_gate_kernel: 32 launches per
// forward, ~10 us each whatever the layer) against the floor of its launch geometry:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/micro/launch_floor.hip -o /tmp/lf && /tmp/lf
// variants: threads per workgroup x workgroups x dynamic LDS, body = nothing | one dependent global round trip | three round trips with
// barriers between them (the gate's squeeze -> FC1 -> FC2 chain).
// MI355X, round 6: 2.9-4.2 us per launch for EVERY variant (1 024 x 256 with 24 KB: empty 2.86, one trip 3.07, three trips 3.93) — the
// floor of a dependent small kernel is ~3 us, so the gate's 9.2-9.6 us (rocprofv3, inside a forward) is ~6 us of its own body; a Python
// loop over ops.se_gate cannot see that (10.5 us per call is the ctypes + torch.empty host path).
#include <hip/hip_runtime.h>
#include <cstdio>

template <int TRIPS>
__global__ void k(const float* __restrict__ in, float* __restrict__ out, int n) {
  extern __shared__ float sm[];
  float v = threadIdx.x;
  int idx = (blockIdx.x * blockDim.x + threadIdx.x) % n;
#pragma unroll
  for (int t = 0; t < TRIPS; ++t) {
    v += in[idx];                                   // dependent: the next address comes from the value
    idx = ((int)v & 1023) % n;
    sm[threadIdx.x] = v;
    __syncthreads();
    v += sm[(threadIdx.x + 1) % blockDim.x];
    __syncthreads();
  }
  if (TRIPS == 0 || v == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = v;
}

template <int TRIPS>
static float run(int threads, int blocks, int lds, const float* in, float* out, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k<TRIPS>, dim3(blocks), dim3(threads), lds, 0, in, out, 4096);
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<TRIPS>, dim3(blocks), dim3(threads), lds, 0, in, out, 4096);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps * 1e3f;
}

int main() {
  float *in, *out;
  hipMalloc(&in, 4096 * 4); hipMalloc(&out, 1024 * 1024 * 4);
  hipMemset(in, 0, 4096 * 4);
  const int geo[][3] = {{1024, 256, 24 * 1024}, {1024, 64, 24 * 1024}, {256, 256, 8 * 1024}, {256, 1024, 8 * 1024}, {512, 256, 16 * 1024}, {64, 256, 1024}};
  printf("threads x workgroups (LDS)      empty   1 trip   3 trips   (us per launch, 500 launches back to back)\n");
  for (auto& g : geo) {
    const float a = run<0>(g[0], g[1], g[2], in, out, 500), b = run<1>(g[0], g[1], g[2], in, out, 500), c = run<3>(g[0], g[1], g[2], in, out, 500);
    printf("%5d x %5d (%2d KB)          %6.2f   %6.2f   %6.2f\n", g[0], g[1], g[2] / 1024, a, b, c);
  }
  return 0;
}
