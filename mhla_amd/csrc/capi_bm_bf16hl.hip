// Block-mix generic / split-operand launches for bf16_t tensors with fp32 block summaries and hi + lo intermediate operands: the
// default arithmetic (the reference's fp32 intermediates over bf16-rounded tensors; see capi_bm_typed.hpp).
#include "capi_bm_typed.hpp"

namespace mhla {
namespace capi {
template int bm_fwd_typed<bf16_t, false>(const BmCall&);
template int bm_bwd_typed<bf16_t, false>(const BmCall&);
}  // namespace capi
}  // namespace mhla
