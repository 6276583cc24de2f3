// Block-mix generic / split-operand launches for f16_t tensors (see capi_bm_typed.hpp).
#include "capi_bm_typed.hpp"

namespace mhla {
namespace capi {
template int bm_fwd_typed<f16_t, false>(const BmCall&);
template int bm_bwd_typed<f16_t, false>(const BmCall&);
}  // namespace capi
}  // namespace mhla
