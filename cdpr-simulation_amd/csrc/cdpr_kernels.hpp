// cdpr_kernels.hpp — the step-kernel families behind function pointers.  Every family is instantiated in a translation
// unit of its own (k_*.hip: they compile in parallel, `make -j`), the host side (cdpr_engine.hip) only picks.
#pragma once
#include "cdpr_step_kernel.hpp"
#include "cdpr_step_kernel_f64.hpp"
#include "cdpr_general_step.hpp"

namespace cdpr {

using StepKernel = void (*)(const StepArgs);
using F64Kernel = void (*)(const F64Args);
using GenKernel = void (*)(const StepArgs, const GenCtl);

// FK / TD exist from six cables on: below that the stage flags fall back to the plain instantiation
#define CDPR_PICK_STAGES(N, EXPR)                 \
  do {                                            \
    if constexpr ((N) >= 6) {                     \
      if (fk && td) return EXPR(N, true, true);   \
      if (fk) return EXPR(N, true, false);        \
      if (td) return EXPR(N, false, true);        \
    }                                             \
    return EXPR(N, false, false);                 \
  } while (0)
#define CDPR_PICK_CABLES(FN, ...)          \
  switch (n) {                             \
    case 1: return FN<1>(__VA_ARGS__);     \
    case 2: return FN<2>(__VA_ARGS__);     \
    case 3: return FN<3>(__VA_ARGS__);     \
    case 4: return FN<4>(__VA_ARGS__);     \
    case 5: return FN<5>(__VA_ARGS__);     \
    case 6: return FN<6>(__VA_ARGS__);     \
    case 7: return FN<7>(__VA_ARGS__);     \
    case 8: return FN<8>(__VA_ARGS__);     \
  }                                        \
  return nullptr

// nine to twelve cables (round 6: cube.yaml's `points` list is free-length): the first-generation kernel and the MPC rollout only
#define CDPR_PICK_CABLES12(FN, ...)        \
  switch (n) {                             \
    case 9: return FN<9>(__VA_ARGS__);     \
    case 10: return FN<10>(__VA_ARGS__);   \
    case 11: return FN<11>(__VA_ARGS__);   \
    case 12: return FN<12>(__VA_ARGS__);   \
  }                                        \
  CDPR_PICK_CABLES(FN, __VA_ARGS__)

// k_step.hip: first-generation kernel (cdpr_step_kernel.hpp): one step / several steps per launch, the low-register
// build (FK on, n >= 6), the MPC rollout, the PHYS instantiations (lumped legs, joint stop)
StepKernel pick_step_kernel(bool single, uint32_t n, bool fk, bool td);
StepKernel pick_lowreg_kernel(uint32_t n, bool td);
StepKernel pick_rollout_kernel(uint32_t n, bool fk, bool td);
enum PhysKind { kPhysStep, kPhysRollout };
StepKernel pick_phys_kernel(uint32_t n, bool fk, bool td, int kind);
// k_pr.hip: per-robot handles on the register-resident path (PR = true)
StepKernel pick_pr_kernel(bool single, bool rollout, bool lowreg, uint32_t n, bool fk, bool td);
// k_onestep.hip: second-generation one-step kernel and the role-split kernel (cdpr_onestep_kernel.hpp)
StepKernel pick_onestep_kernel(uint32_t n, bool fk, bool td);
StepKernel pick_onestep_persist_kernel(uint32_t n, bool fk, bool td);  // one wave per SIMD walking over blocks of 64 robots
StepKernel pick_split_kernel(uint32_t n);
StepKernel pick_pr_split_kernel(uint32_t n);
// k_pair.hip / k_cable.hip: the other two wavefront mappings
StepKernel pick_pair_kernel(bool single, uint32_t n, bool fk, bool td);
StepKernel pick_pair_stream_kernel(uint32_t n, bool vel);  // several steps per launch in the steady state (no FK / TD): see pair_stream_ok
StepKernel pick_cable_kernel(uint32_t n, bool fk, bool td);
// k_gen.hip: the general controller path (cdpr_general_step.hpp); long_window: derivative windows of 12 .. 32 samples
GenKernel pick_gen_kernel(uint32_t n, bool fk, bool td, bool rollout, bool long_window, bool single);
GenKernel pick_gen_lean11(uint32_t n);   // k_gen_split.hip: the same for larger batches: two waves per SIMD, the rare controller paths by call (cdpr_gen_lean_kernel)
GenKernel pick_gen_split11(uint32_t n);  // k_gen_split.hip: one step per launch, two waves per 64 robots split by role (FK + TD, n >= 6, windows <= 11)
// k_f64.hip: precision = 64
F64Kernel pick_f64_split_kernel(uint32_t n, bool lean);  // lean: nothing cached in LDS (four workgroups per CU)
F64Kernel pick_f64_hold_pr_kernel(uint32_t n, bool full);  // HOLD on per-robot handles
F64Kernel pick_f64_hold_long_kernel(uint32_t n, bool pr, bool tstop);  // HOLD = 2 over Pid records of 32 samples (k_f64_hold_long.hip)
F64Kernel pick_f64_tstop_kernel(uint32_t n, bool pr = false, int hold = 0);  // joint stop / lumped legs; pr: per-robot modes, hold: 0 | 1 | 2 the HOLD level (k_f64_phys.hip)
F64Kernel pick_f64_long_kernel(uint32_t n, bool pr = false, bool tstop = false);  // windows of 12 .. 32 samples (k_f64_long.hip); pr: per-robot modes, tstop: joint stop / lumped legs
F64Kernel pick_f64_split_hold_kernel(uint32_t n, bool lean, bool full);  // ... HOLD instantiations (the position-hold branch in double)
//  // one step per launch, two waves per 64 robots split by role (FK + TD, n >= 6)
F64Kernel pick_f64_pr_kernel(uint32_t n, bool ring_lds);  // per-robot handles (mode, call count and Pid per lane)
F64Kernel pick_f64_hold_kernel(uint32_t n, bool full);  // full: + biquad cascades, cmd_limit 0                 // velocityEpsilon >= 0: the position-hold branch live (both Pids of every cable)
F64Kernel pick_f64_kernel(uint32_t n, bool ring_lds, bool jcache);  // ring_lds: the derivative rings staged in LDS (small batches); jcache: and the structure-matrix rows (one workgroup per CU)

}  // namespace cdpr
