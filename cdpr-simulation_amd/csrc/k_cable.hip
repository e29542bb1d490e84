// one lane per cable (cdpr_step_kernel_cable.hpp): one kernel for any number of steps per launch
#include "cdpr_kernels.hpp"
#include "cdpr_step_kernel_cable.hpp"
namespace cdpr {
namespace {
#define K_CABLE(N, FK, TD) cdpr_step_kernel_cable<N, FK, TD>
template <int N> StepKernel cable_n(bool fk, bool td) { CDPR_PICK_STAGES(N, K_CABLE); }
}  // namespace
StepKernel pick_cable_kernel(uint32_t n, bool fk, bool td) { CDPR_PICK_CABLES(cable_n, fk, td); }
}  // namespace cdpr
