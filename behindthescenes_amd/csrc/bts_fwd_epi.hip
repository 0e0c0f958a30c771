// The pipelined render kernel with the loss epilogue (per-ray sum of weights * invalid and max of invalid per render view, SURVEY 8f.1:
// a training step then keeps `weights`, `invalid` and `rgb_samps` out of HBM).  Its own translation unit: the instantiations compile in
// parallel with the plain ones of bts_fwd_proj.hip.
#include "bts_render_kernel.h"

namespace bts {
int launch_render_pipelined_epi(const FwdParams& p, int C, int HD, int NB, int grid, hipStream_t s) { return launch_render_p<true>(p, C, HD, NB, grid, s); }
}  // namespace bts
