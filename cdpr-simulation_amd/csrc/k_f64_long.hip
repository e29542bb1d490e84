// precision = 64 with derivative windows of 12 .. 32 samples (W = kWinLong: a ring of 31 errors per cable on the one-wave kernel) - alone
// and (later in round 6) together with per-robot modes and / or the optional physics
#include "cdpr_kernels.hpp"
namespace cdpr {
namespace {
template <int N, bool PR, bool TSTOP> F64Kernel f64_long_n() { return cdpr_step_kernel_f64<N, false, false, PR, 0, TSTOP, kWinLong>; }
template <int N> F64Kernel f64_long_any(bool pr, bool tstop) {
  if (pr) return tstop ? f64_long_n<N, true, true>() : f64_long_n<N, true, false>();
  return tstop ? f64_long_n<N, false, true>() : f64_long_n<N, false, false>();
}
}  // namespace
F64Kernel pick_f64_long_kernel(uint32_t n, bool pr, bool tstop) { CDPR_PICK_CABLES(f64_long_any, pr, tstop); }
}  // namespace cdpr
