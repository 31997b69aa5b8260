// Checks narrow_impl.h's store_row (v_permlane32_swap pairing) on the GPU: every lane's quads carry their channel index.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ccvpe_amd/csrc tools/micro/store_row_check.hip -o /tmp/store_row_check && /tmp/store_row_check
#include "narrow_impl.h"
#include <vector>
#include <cstring>
namespace ccvpe { thread_local char g_err[512]; int num_cus() { return 256; } }
using namespace ccvpe;

template <int NT>
__global__ void k(bf16_t* out, int N, int ldd) {
  const int lane = threadIdx.x & 63, f = lane & 15, q = lane >> 4;
  f32x4 v[NT];
  for (int n = 0; n < NT; ++n)
    for (int r = 0; r < 4; ++r) v[n][r] = (float)(chan_of_quad(n, q) + r) + 0.5f * 0;
  store_row<NT, false>(out, (size_t)f * ldd, v, q, N);
}

template <int NT>
int run(int N) {
  const int ldd = N;
  bf16_t* d;
  hipMalloc(&d, 16 * ldd * 2);
  hipMemset(d, 0xff, 16 * ldd * 2);
  hipLaunchKernelGGL(k<NT>, dim3(1), dim3(64), 0, 0, d, N, ldd);
  std::vector<unsigned short> h(16 * ldd);
  hipMemcpy(h.data(), d, 16 * ldd * 2, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int f = 0; f < 16; ++f)
    for (int c = 0; c < N; ++c) {
      unsigned u = (unsigned)h[f * ldd + c] << 16;
      float x;
      memcpy(&x, &u, 4);
      if (x != (float)c) { if (bad < 8) printf("NT %d N %d: pixel %d channel %d holds %g\n", NT, N, f, c, x); ++bad; }
    }
  printf("NT %d N %d: %s (%d bad)\n", NT, N, bad ? "FAIL" : "ok", bad);
  hipFree(d);
  return bad;
}
int main() { return run<3>(40) + run<2>(32) + run<4>(64) + run<1>(8) + run<1>(16) + run<3>(48); }
