// The bucket-accumulation kernel(s) of one group configuration, with the multiply-add order of the product-scanning Montgomery
// multiplication pinned (field29.hpp, CSH_PIN_MADS): the carry of a column is the addend of the next column's first multiply-add.
#define CSH_PIN_MADS 3
#include "msm_impl.hpp"

namespace csh {
CSH_MSM_ACCUM_INSTANTIATE(, Bls377G1Cfg)
}  // namespace csh
