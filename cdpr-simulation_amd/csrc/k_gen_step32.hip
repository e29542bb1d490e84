// general controller path in one launch per step (cdpr_general_step.hpp): stepping kernel, windows of 12 .. 32 samples
#include "cdpr_kernels.hpp"
#include "cdpr_general_step.hpp"
namespace cdpr {
namespace {
#define K_GEN(N, FK, TD) cdpr_gen_step_kernel<N, FK, TD, false, kGenMaxBuf, false>
template <int N> GenKernel gen_n(bool fk, bool td) { CDPR_PICK_STAGES(N, K_GEN); }
}  // namespace
GenKernel pick_gen_step32(uint32_t n, bool fk, bool td) { CDPR_PICK_CABLES(gen_n, fk, td); }
}  // namespace cdpr
