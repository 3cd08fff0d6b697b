// Explicit instantiation of the MSM templates for one group configuration (see msm_impl.hpp).
#include "msm_impl.hpp"

namespace csh {
CSH_MSM_ACCUM_INSTANTIATE(extern, GrumpkinG1Cfg)
CSH_MSM_INSTANTIATE(, GrumpkinG1Cfg)
}  // namespace csh
