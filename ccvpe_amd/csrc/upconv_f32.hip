// folded deconv + 3x3 kernels, fp32 (upconv_impl.h).
#include "upconv_impl.h"
extern "C" int ccvpe_upconv3x3_f32(const ccvpe_upconv_desc* d, void* stream) { return ccvpe::upconv_any<float>(d, stream); }
namespace ccvpe { int upconv_route_f32(const ccvpe_upconv_desc* d, int* route) { return upconv_any<float>(d, nullptr, route); } }
