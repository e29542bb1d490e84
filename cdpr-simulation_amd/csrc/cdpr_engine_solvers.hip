// cdpr_engine_solvers.hip - the one-shot solvers of the C-ABI (cdpr_solve_ik / fk / td) over cdpr_solver_kernel.
#include "cdpr_engine_internal.hpp"

namespace cdpr_host {

using SolveKernel = void (*)(const SolveArgs);

template <int N>
SolveKernel pick_solver_n(int op) {
  if (op == kSolveIk) return cdpr_solver_kernel<N, kSolveIk>;
  if constexpr (N >= 6) {
    if (op == kSolveFk) return cdpr_solver_kernel<N, kSolveFk>;
    if (op == kSolveTd) return cdpr_solver_kernel<N, kSolveTd>;
  }
  return nullptr;
}

SolveKernel pick_solver(uint32_t n, int op) {
  switch (n) {
    case 1: return pick_solver_n<1>(op);
    case 2: return pick_solver_n<2>(op);
    case 3: return pick_solver_n<3>(op);
    case 4: return pick_solver_n<4>(op);
    case 5: return pick_solver_n<5>(op);
    case 6: return pick_solver_n<6>(op);
    case 7: return pick_solver_n<7>(op);
    case 8: return pick_solver_n<8>(op);
    case 9: return pick_solver_n<9>(op);
    case 10: return pick_solver_n<10>(op);
    case 11: return pick_solver_n<11>(op);
    case 12: return pick_solver_n<12>(op);
  }
  return nullptr;
}

static int solver_prolog(cdpr_handle_t h, int op, const char* what, SolveKernel* k) {
  if (!h) return CDPR_ERR_INVALID;
  if (set_device(h) != CDPR_OK) return CDPR_ERR_DEVICE;
  *k = pick_solver(h->n, op);
  if (!*k) {
    h->err = std::string(what) + " needs at least 6 cables";
    return CDPR_ERR_UNSUPPORTED;
  }
  return CDPR_OK;
}

}  // namespace cdpr_host

#define UP(buf, src, count, T)                                                                              \
  do {                                                                                                       \
    HIP_TRY(h, (buf).alloc((size_t)(count) * sizeof(T)));                                                    \
    if (src) HIP_TRY(h, hipMemcpyAsync((buf).p, (src), (size_t)(count) * sizeof(T), hipMemcpyHostToDevice, h->stream)); \
  } while (0)

#define DOWN(dst, buf, count, T)                                                                             \
  do {                                                                                                       \
    if (dst) HIP_TRY(h, hipMemcpyAsync((dst), (buf).p, (size_t)(count) * sizeof(T), hipMemcpyDeviceToHost, h->stream)); \
  } while (0)

int cdpr_solve_ik(cdpr_handle_t h, const float* pose7, const float* twist6, float* q, float* qdot, float* jac) {
  SolveKernel k;
  int rc = solver_prolog(h, kSolveIk, "cdpr_solve_ik", &k);
  if (rc != CDPR_OK) return rc;
  if (!pose7) {
    h->err = "cdpr_solve_ik: pose7 is required";
    return CDPR_ERR_INVALID;
  }
  const size_t B = h->batch, n = h->n;
  DevBuf dp, dt, dq, dqd, dj;
  UP(dp, pose7, B * 7, float);
  UP(dt, twist6, B * 6, float);
  UP(dq, (const float*)nullptr, B * n, float);
  UP(dqd, (const float*)nullptr, B * n, float);
  UP(dj, (const float*)nullptr, B * n * 6, float);
  SolveArgs a{};
  a.geom = h->d_geom;
  a.batch = h->batch;
  a.pose7 = dp.as<float>();
  a.twist6 = twist6 ? dt.as<float>() : nullptr;
  a.q = dq.as<float>();
  a.qdot = dqd.as<float>();
  a.jac = dj.as<float>();
  hipLaunchKernelGGL(k, dim3((h->batch + 63u) / 64u), dim3(64), 0, h->stream, a);
  HIP_TRY(h, hipGetLastError());
  DOWN(q, dq, B * n, float);
  DOWN(qdot, dqd, B * n, float);
  DOWN(jac, dj, B * n * 6, float);
  HIP_TRY(h, wait_stream(h));
  return CDPR_OK;
}

int cdpr_solve_fk(cdpr_handle_t h, const float* lengths, const float* seed7, float* pose7, float* residual, int32_t* iterations) {
  SolveKernel k;
  int rc = solver_prolog(h, kSolveFk, "cdpr_solve_fk", &k);
  if (rc != CDPR_OK) return rc;
  if (!lengths || !seed7 || !pose7) {
    h->err = "cdpr_solve_fk: lengths, seed7 and pose7 are required";
    return CDPR_ERR_INVALID;
  }
  const size_t B = h->batch, n = h->n;
  DevBuf dl, ds, dp, dr, di;
  UP(dl, lengths, B * n, float);
  UP(ds, seed7, B * 7, float);
  UP(dp, (const float*)nullptr, B * 7, float);
  UP(dr, (const float*)nullptr, B, float);
  UP(di, (const int32_t*)nullptr, B, int32_t);
  SolveArgs a{};
  a.geom = h->d_geom;
  a.batch = h->batch;
  a.pose7 = ds.as<float>();
  a.lengths = dl.as<float>();
  a.pose_out = dp.as<float>();
  a.residual = dr.as<float>();
  a.iters = di.as<int32_t>();
  a.fk_lambda = h->base.fk_lambda;
  a.fk_tol = h->base.fk_tol;
  a.fk_iters = h->base.fk_iters > 0 ? h->base.fk_iters : 4;
  hipLaunchKernelGGL(k, dim3((h->batch + 63u) / 64u), dim3(64), 0, h->stream, a);
  HIP_TRY(h, hipGetLastError());
  DOWN(pose7, dp, B * 7, float);
  DOWN(residual, dr, B, float);
  DOWN(iterations, di, B, int32_t);
  HIP_TRY(h, wait_stream(h));
  return CDPR_OK;
}

int cdpr_solve_td(cdpr_handle_t h, const float* pose7, const float* wrench6, float* tension, int32_t* infeasible) {
  SolveKernel k;
  int rc = solver_prolog(h, kSolveTd, "cdpr_solve_td", &k);
  if (rc != CDPR_OK) return rc;
  if (!pose7 || !wrench6 || !tension) {
    h->err = "cdpr_solve_td: pose7, wrench6 and tension are required";
    return CDPR_ERR_INVALID;
  }
  if (!(h->cfg.td_f_max > h->cfg.td_f_min)) {
    h->err = "cdpr_solve_td: td_f_max must exceed td_f_min";
    return CDPR_ERR_INVALID;
  }
  const size_t B = h->batch, n = h->n;
  DevBuf dp, dw, dt, df;
  UP(dp, pose7, B * 7, float);
  UP(dw, wrench6, B * 6, float);
  UP(dt, (const float*)nullptr, B * n, float);
  UP(df, (const int32_t*)nullptr, B, int32_t);
  SolveArgs a{};
  a.geom = h->d_geom;
  a.batch = h->batch;
  a.pose7 = dp.as<float>();
  a.wrench6 = dw.as<float>();
  a.tension = dt.as<float>();
  a.flag = df.as<int32_t>();
  a.td_min = h->base.td_min;
  a.td_max = h->base.td_max;
  a.td_mid = h->base.td_mid;
  hipLaunchKernelGGL(k, dim3((h->batch + 63u) / 64u), dim3(64), 0, h->stream, a);
  HIP_TRY(h, hipGetLastError());
  DOWN(tension, dt, B * n, float);
  DOWN(infeasible, df, B, int32_t);
  HIP_TRY(h, wait_stream(h));
  return CDPR_OK;
}
