// two lanes per robot (cdpr_step_kernel_pair.hpp), n = 4 or 8
#include "cdpr_kernels.hpp"
#include "cdpr_step_kernel_pair.hpp"
namespace cdpr {
namespace {
#define K_PAIR(N, FK, TD) cdpr_step_kernel_pair<N, FK, TD, SINGLE>
template <int N, bool SINGLE> StepKernel stage(bool fk, bool td) { CDPR_PICK_STAGES(N, K_PAIR); }
}  // namespace
StepKernel pick_pair_kernel(bool single, uint32_t n, bool fk, bool td) {
  if (n == 4) return single ? stage<4, true>(fk, td) : stage<4, false>(fk, td);
  return single ? stage<8, true>(fk, td) : stage<8, false>(fk, td);
}
// the steady-state several-steps launch (cdpr_pair_stream_kernel): FK-less, TD-less handles; vel = the Pid sees the joint velocity
StepKernel pick_pair_stream_kernel(uint32_t n, bool vel) {
  if (n == 4) return vel ? cdpr_pair_stream_kernel<4, true> : cdpr_pair_stream_kernel<4, false>;
  if (n == 8) return vel ? cdpr_pair_stream_kernel<8, true> : cdpr_pair_stream_kernel<8, false>;
  return nullptr;
}
}  // namespace cdpr
