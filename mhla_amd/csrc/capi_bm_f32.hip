// Block-mix generic / split-operand launches for float tensors (see capi_bm_typed.hpp).
#include "capi_bm_typed.hpp"

namespace mhla {
namespace capi {
template int bm_fwd_typed<float, false>(const BmCall&);
template int bm_bwd_typed<float, false>(const BmCall&);
}  // namespace capi
}  // namespace mhla
