// pw_gemm_kernel, bf16 instantiations (conv_pw_impl.h).
#include "conv_pw_impl.h"
namespace ccvpe {
template int pw_dispatch<bf16_t>(const IgemmParams&, int, int, int, hipStream_t);
}
