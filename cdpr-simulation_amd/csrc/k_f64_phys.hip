// precision = 64 with the optional physics (joint stop, lumped legs: the TSTOP instantiations of cdpr_step_kernel_f64) - alone, and
// (round 6) together with per-robot modes and / or the hold branch: the world step does not care who set the forces
#include "cdpr_kernels.hpp"
namespace cdpr {
namespace {
template <int N, bool PR, int HOLD> F64Kernel f64_tstop_n() { return cdpr_step_kernel_f64<N, false, false, PR, HOLD, true>; }
template <int N> F64Kernel f64_tstop_any(bool pr, int hold) {
  if (pr) return hold == 2 ? f64_tstop_n<N, true, 2>() : hold == 1 ? f64_tstop_n<N, true, 1>() : f64_tstop_n<N, true, 0>();
  return hold == 2 ? f64_tstop_n<N, false, 2>() : hold == 1 ? f64_tstop_n<N, false, 1>() : f64_tstop_n<N, false, 0>();
}
}  // namespace
F64Kernel pick_f64_tstop_kernel(uint32_t n, bool pr, int hold) { CDPR_PICK_CABLES(f64_tstop_any, pr, hold); }
}  // namespace cdpr
