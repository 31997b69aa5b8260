// conv3x3_kernel, bf16 instantiations (conv3x3_impl.h).
#include "conv3x3_impl.h"
namespace ccvpe {
template int conv3x3_dispatch<bf16_t>(const IgemmParams&, int, int, int, int, hipStream_t);
}
