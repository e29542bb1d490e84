// precision = 64: the step in the reference's own precision (cdpr_step_kernel_f64.hpp)
#include "cdpr_kernels.hpp"
namespace cdpr {
namespace {
template <int N> F64Kernel f64_pr_n(bool ring_lds) { return ring_lds ? cdpr_step_kernel_f64<N, true, false, true> : cdpr_step_kernel_f64<N, false, false, true>; }
template <int N> F64Kernel f64_n(bool ring_lds, bool jcache) {
  if (jcache) return cdpr_step_kernel_f64<N, true, true>;  // (112 KiB of LDS per wave at n = 8: one workgroup per CU)
  return ring_lds ? cdpr_step_kernel_f64<N, true> : cdpr_step_kernel_f64<N, false>;
}
template <int N> F64Kernel f64_hold_n(bool full) { return full ? cdpr_step_kernel_f64<N, false, false, false, 2> : cdpr_step_kernel_f64<N, false, false, false, 1>; }
template <int N> F64Kernel f64_hold_pr_n(bool full) { return full ? cdpr_step_kernel_f64<N, false, false, true, 2> : cdpr_step_kernel_f64<N, false, false, true, 1>; }
}  // namespace
F64Kernel pick_f64_hold_pr_kernel(uint32_t n, bool full) { CDPR_PICK_CABLES(f64_hold_pr_n, full); }  // ... on per-robot handles (the mode per lane)
F64Kernel pick_f64_hold_kernel(uint32_t n, bool full) { CDPR_PICK_CABLES(f64_hold_n, full); }  // the position-hold branch live (both Pids of every cable)
// nine to twelve cables (end of round 6): the plain one-wave kernel, rings in memory (the LDS variants' columns do not fit)
template <int N> F64Kernel f64_n12(bool ring_lds, bool jcache) {
  if constexpr (N > 8)
    return cdpr_step_kernel_f64<N, false>;
  else
    return f64_n<N>(ring_lds, jcache);
}
F64Kernel pick_f64_kernel(uint32_t n, bool ring_lds, bool jcache) { CDPR_PICK_CABLES12(f64_n12, ring_lds, jcache); }
F64Kernel pick_f64_pr_kernel(uint32_t n, bool ring_lds) { CDPR_PICK_CABLES(f64_pr_n, ring_lds); }  // per-robot modes (PR)
template <int H> F64Kernel f64_split_hold(uint32_t n, bool lean) {
  if (lean) {
    switch (n) {
      case 6: return cdpr_split_kernel_f64<6, true, H>;
      case 7: return cdpr_split_kernel_f64<7, true, H>;
      case 8: return cdpr_split_kernel_f64<8, true, H>;
    }
    return nullptr;
  }
  switch (n) {
    case 6: return cdpr_split_kernel_f64<6, false, H>;
    case 7: return cdpr_split_kernel_f64<7, false, H>;
    case 8: return cdpr_split_kernel_f64<8, false, H>;
  }
  return nullptr;
}
F64Kernel pick_f64_split_hold_kernel(uint32_t n, bool lean, bool full) { return full ? f64_split_hold<2>(n, lean) : f64_split_hold<1>(n, lean); }  // ... with the hold branch live
F64Kernel pick_f64_split_kernel(uint32_t n, bool lean) {
  if (lean) {
    switch (n) {
      case 6: return cdpr_split_kernel_f64<6, true>;
      case 7: return cdpr_split_kernel_f64<7, true>;
      case 8: return cdpr_split_kernel_f64<8, true>;
    }
    return nullptr;
  }
  switch (n) {
    case 6: return cdpr_split_kernel_f64<6>;
    case 7: return cdpr_split_kernel_f64<7>;
    case 8: return cdpr_split_kernel_f64<8>;
  }
  return nullptr;
}
}  // namespace cdpr
