// second-generation one-step kernel (controller rows staged through LDS) and the role-split kernel, uniform and PR
#include "cdpr_kernels.hpp"
#include "cdpr_onestep_kernel.hpp"
namespace cdpr {
namespace {
#define K_ONE(N, FK, TD) cdpr_onestep_kernel<N, FK, TD>
#define K_ONE_PERSIST(N, FK, TD) cdpr_onestep_kernel<N, FK, TD, true>
template <int N> StepKernel one_n(bool fk, bool td) { CDPR_PICK_STAGES(N, K_ONE); }
template <int N> StepKernel one_persist_n(bool fk, bool td) { CDPR_PICK_STAGES(N, K_ONE_PERSIST); }
}  // namespace
StepKernel pick_onestep_kernel(uint32_t n, bool fk, bool td) { CDPR_PICK_CABLES(one_n, fk, td); }
StepKernel pick_onestep_persist_kernel(uint32_t n, bool fk, bool td) { CDPR_PICK_CABLES(one_persist_n, fk, td); }
StepKernel pick_split_kernel(uint32_t n) {
  switch (n) {
    case 6: return cdpr_split_kernel<6>;
    case 7: return cdpr_split_kernel<7>;
    case 8: return cdpr_split_kernel<8>;
  }
  return nullptr;
}
StepKernel pick_pr_split_kernel(uint32_t n) {
  switch (n) {
    case 6: return cdpr_split_kernel<6, true>;
    case 7: return cdpr_split_kernel<7, true>;
    case 8: return cdpr_split_kernel<8, true>;
  }
  return nullptr;
}
}  // namespace cdpr
