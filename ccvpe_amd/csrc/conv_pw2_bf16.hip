// bf16 instantiation of the pointwise ring kernel (conv_pw2_impl.h)
#include "conv_pw2_impl.h"
namespace ccvpe {
template bool pw2_supported<bf16_t>(const IgemmParams&, int, int, int);
template int pw2_dispatch<bf16_t>(const IgemmParams&, int, int, int, hipStream_t);
}  // namespace ccvpe
