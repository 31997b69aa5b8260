// pw_gemm_kernel, f32 instantiations (conv_pw_impl.h).
#include "conv_pw_impl.h"
namespace ccvpe {
template int pw_dispatch<float>(const IgemmParams&, int, int, int, hipStream_t);
}
