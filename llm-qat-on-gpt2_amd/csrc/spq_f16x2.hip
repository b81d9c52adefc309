// placeholder until the 2-limb fp16 path lands
#include "spq_common.h"
namespace spq {
size_t fwd_f16x2_workspace_bytes(int64_t, int64_t, int64_t, int64_t) { return 0; }
int fwd_f16x2(const spq_fwd_args*, hipStream_t) {
  set_error("spq_linear_lora_fwd: SPQ_PATH_F16X2 not built");
  return SPQ_ERR_UNSUPPORTED;
}
}  // namespace spq
