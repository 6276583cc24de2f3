// Block-mix generic / split-operand launches for bf16_t tensors (see capi_bm_typed.hpp).
#include "capi_bm_typed.hpp"

namespace mhla {
namespace capi {
template int bm_fwd_typed<bf16_t>(const BmCall&);
template int bm_bwd_typed<bf16_t>(const BmCall&);
}  // namespace capi
}  // namespace mhla
