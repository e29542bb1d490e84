// per-robot handles on the register-resident path (PR = true): the first-generation kernel in every shape (one step,
// several steps, low-register, rollout); the role-split PR kernel lives in k_onestep.hip
#include "cdpr_kernels.hpp"
namespace cdpr {
namespace {
template <int N, bool SINGLE, bool ROLLOUT, bool LOWREG>
StepKernel stage(bool fk, bool td) {
  if constexpr (N >= 6) {
    if (fk && td) return cdpr_step_kernel<N, true, true, SINGLE, ROLLOUT, LOWREG, false, true>;
    if (fk) return cdpr_step_kernel<N, true, false, SINGLE, ROLLOUT, LOWREG, false, true>;
    if constexpr (!LOWREG)
      if (td) return cdpr_step_kernel<N, false, true, SINGLE, ROLLOUT, false, false, true>;
  }
  if constexpr (!LOWREG) return cdpr_step_kernel<N, false, false, SINGLE, ROLLOUT, false, false, true>;
  return nullptr;
}
template <int N>
StepKernel pr_n(bool single, bool rollout, bool lowreg, bool fk, bool td) {
  if (rollout) return stage<N, false, true, false>(fk, td);
  if (!single) return stage<N, false, false, false>(fk, td);
  return lowreg ? stage<N, true, false, true>(fk, td) : stage<N, true, false, false>(fk, td);
}
}  // namespace
StepKernel pick_pr_kernel(bool single, bool rollout, bool lowreg, uint32_t n, bool fk, bool td) { CDPR_PICK_CABLES(pr_n, single, rollout, lowreg, fk, td); }
}  // namespace cdpr
