// precision = 64: the step in the reference's own precision (cdpr_step_kernel_f64.hpp)
#include "cdpr_kernels.hpp"
namespace cdpr {
namespace {
template <int N> F64Kernel f64_n(bool ring_lds) { return ring_lds ? cdpr_step_kernel_f64<N, true> : cdpr_step_kernel_f64<N, false>; }
}  // namespace
F64Kernel pick_f64_kernel(uint32_t n, bool ring_lds) { CDPR_PICK_CABLES(f64_n, ring_lds); }
}  // namespace cdpr
