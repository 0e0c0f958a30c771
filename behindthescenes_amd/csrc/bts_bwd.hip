// Backward of the fused renderer (placeholder until the kernels land).
#include "bts_common.h"
namespace bts {
void set_error(const char* fmt, const char* a = "", long b = 0, long c = 0, long d = 0);
size_t render_bwd_workspace_impl(const BtsFieldCfg*, const BtsRenderArgs*) { return 0; }
int render_bwd_impl(const BtsFieldCfg*, const BtsFieldTensors*, const BtsRenderArgs*, const BtsRenderGrads*, void*, size_t, hipStream_t) {
  set_error("%s: backward not built", "bts_render_bwd");
  return BTS_E_UNSUPPORTED;
}
}  // namespace bts
