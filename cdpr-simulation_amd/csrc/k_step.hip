// first-generation step kernel: one step, several steps, low-register, rollout, PHYS (see cdpr_kernels.hpp)
#include "cdpr_kernels.hpp"
namespace cdpr {
namespace {
#define K_STEP(N, FK, TD) cdpr_step_kernel<N, FK, TD, SINGLE>
template <int N, bool SINGLE> StepKernel stage(bool fk, bool td) { CDPR_PICK_STAGES(N, K_STEP); }
template <int N> StepKernel step_n(bool single, bool fk, bool td) { return single ? stage<N, true>(fk, td) : stage<N, false>(fk, td); }
#define K_ROLL(N, FK, TD) cdpr_step_kernel<N, FK, TD, false, true>
template <int N> StepKernel roll_n(bool fk, bool td) { CDPR_PICK_STAGES(N, K_ROLL); }
// lumped-leg physics / joint stop (PHYS = true): one generic stepping kernel (any steps per launch) and the MPC rollout
#define K_PHYS_STEP(N, FK, TD) cdpr_step_kernel<N, FK, TD, false, false, false, true>
#define K_PHYS_ROLL(N, FK, TD) cdpr_step_kernel<N, FK, TD, false, true, false, true>
template <int N> StepKernel phys_step_n(bool fk, bool td) { CDPR_PICK_STAGES(N, K_PHYS_STEP); }
template <int N> StepKernel phys_roll_n(bool fk, bool td) { CDPR_PICK_STAGES(N, K_PHYS_ROLL); }
template <int N> StepKernel phys_n(bool fk, bool td, int kind) { return kind == kPhysRollout ? phys_roll_n<N>(fk, td) : phys_step_n<N>(fk, td); }
}  // namespace

StepKernel pick_step_kernel(bool single, uint32_t n, bool fk, bool td) { CDPR_PICK_CABLES12(step_n, single, fk, td); }
StepKernel pick_rollout_kernel(uint32_t n, bool fk, bool td) { CDPR_PICK_CABLES12(roll_n, fk, td); }
StepKernel pick_phys_kernel(uint32_t n, bool fk, bool td, int kind) { CDPR_PICK_CABLES(phys_n, fk, td, kind); }
// one-step kernels compiled for two waves per SIMD (FK on, n >= 6): see LOWREG in cdpr_step_kernel.hpp
StepKernel pick_lowreg_kernel(uint32_t n, bool td) {
  switch (n) {
    case 6: return td ? cdpr_step_kernel<6, true, true, true, false, true> : cdpr_step_kernel<6, true, false, true, false, true>;
    case 7: return td ? cdpr_step_kernel<7, true, true, true, false, true> : cdpr_step_kernel<7, true, false, true, false, true>;
    case 8: return td ? cdpr_step_kernel<8, true, true, true, false, true> : cdpr_step_kernel<8, true, false, true, false, true>;
  }
  return nullptr;
}
}  // namespace cdpr
