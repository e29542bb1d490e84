// general controller path in one launch per step (cdpr_general_step.hpp): several steps per launch, windows up to 11 samples (record rows staged through LDS)
#include "cdpr_kernels.hpp"
#include "cdpr_general_step.hpp"
namespace cdpr {
namespace {
#define K_GEN(N, FK, TD) cdpr_gen_step_kernel<N, FK, TD, false, 11, false>
template <int N> GenKernel gen_n(bool fk, bool td) { CDPR_PICK_STAGES(N, K_GEN); }
}  // namespace
GenKernel pick_gen_step11(uint32_t n, bool fk, bool td) { CDPR_PICK_CABLES(gen_n, fk, td); }
}  // namespace cdpr
