// PROJ = true instantiations of the fused forward kernel (projected feature map G; the default render path).
#include "bts_field_kernel.h"

namespace bts {
template int launch_field<false, true>(const FwdParams&, int, int, int, int, hipStream_t);
template int launch_field<true, true>(const FwdParams&, int, int, int, int, hipStream_t);
template int launch_render<true>(const FwdParams&, int, int, int, int, hipStream_t);
}  // namespace bts
