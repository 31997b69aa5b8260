// conv3x3_kernel, f32 instantiations (conv3x3_impl.h).
#include "conv3x3_impl.h"
namespace ccvpe {
template int conv3x3_dispatch<float>(const IgemmParams&, int, int, int, int, hipStream_t);
}
