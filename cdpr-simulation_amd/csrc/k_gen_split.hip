// general controller path, one step per launch, role-split form (cdpr_general_split.hpp): FK + TD handles, windows up to 11 samples
#include "cdpr_kernels.hpp"
#include "cdpr_general_split.hpp"
namespace cdpr {
GenKernel pick_gen_split11(uint32_t n) {
  switch (n) {
    case 6: return cdpr_gen_split_kernel<6, 11>;
    case 7: return cdpr_gen_split_kernel<7, 11>;
    case 8: return cdpr_gen_split_kernel<8, 11>;
  }
  return nullptr;
}
GenKernel pick_gen_lean11(uint32_t n) {
  switch (n) {
    case 6: return cdpr_gen_lean_kernel<6>;
    case 7: return cdpr_gen_lean_kernel<7>;
    case 8: return cdpr_gen_lean_kernel<8>;
  }
  return nullptr;
}
}  // namespace cdpr
