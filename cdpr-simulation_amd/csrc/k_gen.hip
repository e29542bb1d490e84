// picks among the general-path kernel families (each instantiated in a translation unit of its own: k_gen_*.hip)
#include "cdpr_kernels.hpp"
namespace cdpr {
GenKernel pick_gen_one11(uint32_t n, bool fk, bool td);
GenKernel pick_gen_step11(uint32_t n, bool fk, bool td);
GenKernel pick_gen_roll11(uint32_t n, bool fk, bool td);
GenKernel pick_gen_step32(uint32_t n, bool fk, bool td);
GenKernel pick_gen_roll32(uint32_t n, bool fk, bool td);
GenKernel pick_gen_kernel(uint32_t n, bool fk, bool td, bool rollout, bool long_window, bool single) {
  if (long_window) return rollout ? pick_gen_roll32(n, fk, td) : pick_gen_step32(n, fk, td);
  if (rollout) return pick_gen_roll11(n, fk, td);
  return single ? pick_gen_one11(n, fk, td) : pick_gen_step11(n, fk, td);
}
}  // namespace cdpr
