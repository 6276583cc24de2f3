// Block-mix split-operand launches for bf16_t tensors with bf16 block summaries: the opt-in MHLA_FLAG_BF16_SUMMARIES arithmetic
// (see capi_bm_typed.hpp; the default, fp32-grade summaries, is capi_bm_bf16hl.hip).
#include "capi_bm_typed.hpp"

namespace mhla {
namespace capi {
template int bm_fwd_typed<bf16_t, true>(const BmCall&);
template int bm_bwd_typed<bf16_t, true>(const BmCall&);
}  // namespace capi
}  // namespace mhla
