// PROJ = true instantiations of the fused forward (projected feature map G; the default render path): the software-pipelined
// render kernel (bts_render_kernel.h) and -- for A/B measurements in the probe build -- the previous render kernels.  (Field queries on the
// projected map: bts_query.hip.)
#include "bts_render_kernel.h"

namespace bts {
#ifdef BTS_PROBE   // the compact round-1 kernels on the projected map: A/B in the probe build only (BTS_LANE_IS_RAY, BTS_RENDER_V1)
template int launch_field<false, true>(const FwdParams&, int, int, int, int, hipStream_t);
template int launch_render<true>(const FwdParams&, int, int, int, int, hipStream_t);
#endif
int launch_render_pipelined_epi(const FwdParams& p, int C, int HD, int NB, int grid, hipStream_t s);   // bts_fwd_epi.hip
int launch_render_pipelined(const FwdParams& p, int C, int HD, int NB, int grid, hipStream_t s) {
  if (p.invalid_wsum || p.invalid_any) return launch_render_pipelined_epi(p, C, HD, NB, grid, s);
  return launch_render_p<false>(p, C, HD, NB, grid, s);
}
}  // namespace bts
