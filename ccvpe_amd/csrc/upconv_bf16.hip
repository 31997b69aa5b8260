// folded deconv + 3x3 kernels, bf16 storage (upconv_impl.h).
#include "upconv_impl.h"
extern "C" int ccvpe_upconv3x3_bf16(const ccvpe_upconv_desc* d, void* stream) { return ccvpe::upconv_any<ccvpe::bf16_t>(d, stream); }
extern "C" int ccvpe_upconv3x3_route(const ccvpe_upconv_desc* d, int is_bf16) {
  int route = 0;
  const int rc = is_bf16 ? ccvpe::upconv_any<ccvpe::bf16_t>(d, nullptr, &route) : ccvpe::upconv_route_f32(d, &route);
  return rc ? rc : route;
}
