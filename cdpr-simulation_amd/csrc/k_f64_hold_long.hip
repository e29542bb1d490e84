// precision = 64: the hold branch / cascades / cmd_limit 0 with derivative windows of 12 .. 32 samples (later in round 6): the HOLD = 2
// instantiations of the one-wave kernel over Pid records of 32 samples (HW = kHoldWinLong) - uniform and per-robot handles, with and
// without the optional physics
#include "cdpr_kernels.hpp"
namespace cdpr {
namespace {
template <int N, bool PR, bool TSTOP> F64Kernel f64_hold_long_n() { return cdpr_step_kernel_f64<N, false, false, PR, 2, TSTOP, kWin, kHoldWinLong>; }
template <int N> F64Kernel f64_hold_long_any(bool pr, bool tstop) {
  if (pr) return tstop ? f64_hold_long_n<N, true, true>() : f64_hold_long_n<N, true, false>();
  return tstop ? f64_hold_long_n<N, false, true>() : f64_hold_long_n<N, false, false>();
}
}  // namespace
F64Kernel pick_f64_hold_long_kernel(uint32_t n, bool pr, bool tstop) { CDPR_PICK_CABLES(f64_hold_long_any, pr, tstop); }
}  // namespace cdpr
