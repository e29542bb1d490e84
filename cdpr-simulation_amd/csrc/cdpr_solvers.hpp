// cdpr_solvers.hpp — one-shot batched kinematics on caller data (gfx950): the stages of the step
// kernel exposed on their own (cdpr_solve_ik / cdpr_solve_fk / cdpr_solve_td of include/cdpr.h).
// Lane-per-robot like the step kernel and built from the same device functions, but with the
// caller's robot-major arrays (float[B][n], float[B][7] ...) read and written directly.
#pragma once
#include "cdpr_step_kernel.hpp"

namespace cdpr {

struct SolveArgs {
  const float* geom;  // pair-interleaved cable geometry (as StepArgs.geom)
  uint32_t batch;
  // inputs (device, robot-major)
  const float* pose7;
  const float* twist6;
  const float* lengths;
  const float* wrench6;
  // outputs
  float* q;
  float* qdot;
  float* jac;       // [B][n][6]
  float* pose_out;  // [B][7]
  float* residual;  // [B]
  int32_t* iters;   // [B]
  float* tension;   // [B][n]
  int32_t* flag;    // [B]
  float fk_lambda, fk_tol;
  int fk_iters;
  float td_min, td_max, td_mid;
};

enum SolveOp : int { kSolveIk = 0, kSolveFk = 1, kSolveTd = 2 };

template <int N>
CDPR_DEV float pair_get(const v2f (&v)[cable_pairs(N)], int i) {
  return (i & 1) ? v[i / 2].y : v[i / 2].x;
}

template <int N, int OP>
__global__ __launch_bounds__(64) void cdpr_solver_kernel(const SolveArgs a) {
  constexpr int NP = cable_pairs(N);
  __shared__ __attribute__((aligned(16))) float lds[NP * kGeomFloatsPerPair];
  const uint32_t lane = threadIdx.x;
  for (uint32_t g = lane; g < (uint32_t)(NP * kGeomFloatsPerPair); g += 64u) lds[g] = a.geom[g];  // (96 floats at n = 12)
  __syncthreads();
  const uint32_t r = blockIdx.x * 64u + lane;
  if (r >= a.batch) return;
  const float* p = a.pose7 + (size_t)r * 7;
  float px = p[0], py = p[1], pz = p[2], qx = p[3], qy = p[4], qz = p[5], qw = p[6];
  v2f len[NP], jac[NP][6], l0[NP];

  if (OP == kSolveIk) {
    ik_pairs<N, true>(lds, px, py, pz, qx, qy, qz, qw, len, jac, l0);
    float tw[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (a.twist6) {
#pragma unroll
      for (int c = 0; c < 6; ++c) tw[c] = a.twist6[(size_t)r * 6 + c];
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int k = i / 2;
      float row[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) row[c] = (i & 1) ? jac[k][c].y : jac[k][c].x;
      if (a.q) a.q[(size_t)r * N + i] = pair_get<N>(l0, i) - pair_get<N>(len, i);
      if (a.qdot) {
        float s = row[0] * tw[0];
#pragma unroll
        for (int c = 1; c < 6; ++c) s = fmaf(row[c], tw[c], s);
        a.qdot[(size_t)r * N + i] = -s;
      }
      if (a.jac) {
#pragma unroll
        for (int c = 0; c < 6; ++c) a.jac[((size_t)r * N + i) * 6 + c] = row[c];
      }
    }
  } else if (OP == kSolveFk) {
    // Newton-Raphson FK ([NEW] SURVEY 8(a) row 14): pose7 is the seed, lengths the measurement
    v2f meas[NP], elen[NP], unused[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      meas[k].x = a.lengths[(size_t)r * N + 2 * k];
      meas[k].y = (2 * k + 1 < N) ? a.lengths[(size_t)r * N + 2 * k + 1] : 0.f;
    }
    bool active = true;
    int it_done = 0;
    for (int it = 0; it < a.fk_iters; ++it) {
      ik_pairs<N, false>(lds, px, py, pz, qx, qy, qz, qw, elen, jac, unused);
      v2f res[NP];
      v2f rm = splat(0.f);
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        res[k] = meas[k] - elen[k];
        rm = max2(rm, abs2(res[k]));
      }
      active = active && !(fmaxf(rm.x, rm.y) < a.fk_tol);
      float g[6];
      jt_times<NP>(jac, res, g);
      normal_solve<NP>(jac, a.fk_lambda, g);
      if (active) {
        px += g[0];
        py += g[1];
        pz += g[2];
        quat_apply_rotvec(qx, qy, qz, qw, g[3], g[4], g[5]);
        ++it_done;
      }
    }
    ik_pairs<N, false>(lds, px, py, pz, qx, qy, qz, qw, elen, jac, unused);
    v2f rm = splat(0.f);
#pragma unroll
    for (int k = 0; k < NP; ++k) rm = max2(rm, abs2(meas[k] - elen[k]));
    float* o = a.pose_out + (size_t)r * 7;
    o[0] = px; o[1] = py; o[2] = pz; o[3] = qx; o[4] = qy; o[5] = qz; o[6] = qw;
    if (a.residual) a.residual[r] = fmaxf(rm.x, rm.y);
    if (a.iters) a.iters[r] = it_done;
  } else {
    // tension distribution for an explicit wrench ([NEW] SURVEY 8(a) row 15):
    // A = -J^T, T = Tm 1 + A^T (A A^T)^-1 (w_d - A Tm 1) = Tm 1 - J (J^T J)^-1 (w_d + J^T Tm 1)
    ik_pairs<N, false>(lds, px, py, pz, qx, qy, qz, qw, len, jac, l0);
    v2f ones[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) ones[k] = splat(a.td_mid);
    float g[6];
    jt_times<NP>(jac, ones, g);
#pragma unroll
    for (int c = 0; c < 6; ++c) g[c] += a.wrench6[(size_t)r * 6 + c];
    normal_solve<NP, false>(jac, 0.f, g);
    int flag = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int k = i / 2;
      float t = a.td_mid;
#pragma unroll
      for (int c = 0; c < 6; ++c) t = fmaf(-((i & 1) ? jac[k][c].y : jac[k][c].x), g[c], t);
      const float tc = fmaxf(fminf(t, a.td_max), a.td_min);
      flag |= (tc != t) ? 1 : 0;
      a.tension[(size_t)r * N + i] = tc;
    }
    if (a.flag) a.flag[r] = flag;
  }
}

// Observable read-out: slot rows (float4 per robot per slot) -> robot-major float arrays as the C-ABI hands them out
// ([B][W]), on the device, so the host gets one contiguous copy instead of transposing B x W values in a scalar loop
// (at 524 288 robots: 50 MB of joint states).  Thread (r, j) picks component comp[j] of slot slot[j] of robot r.
struct UnpackArgs {
  const float4* rows;
  float* out;
  uint32_t stride, batch, width;
  uint8_t slot[24], comp[24];
  uint32_t as_int;  // bit j set: the value is converted to int32 (iteration counts, flags) before it is stored
};

static __global__ __launch_bounds__(256) void cdpr_unpack_kernel(const UnpackArgs a) {
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  if (t >= a.batch * a.width) return;
  const uint32_t r = t / a.width, j = t - r * a.width;
  const float v = comp4(a.rows[(size_t)a.slot[j] * a.stride + r], a.comp[j]);
  a.out[t] = ((a.as_int >> j) & 1u) ? __int_as_float((int)v) : v;
}

// One published step of every robot straight into the caller-visible (pinned, device-mapped) host buffer: five
// robot-major blocks [position | velocity | effort | pose7 | twist6] gathered from the observable rows by one launch,
// then a completion word written by the last workgroup after a system-scope fence.  The host spins on that word: no
// copy engine, no runtime call in the wait (cdpr_get_observables).
struct PublishArgs {
  const float4* rows;
  float* out;               // host-mapped
  uint64_t* done;           // host-mapped completion word
  uint32_t* arrivals;       // device counter, zero between launches
  uint64_t epoch;
  uint32_t stride, batch, n, width;  // width = 3 n + 13
  uint8_t slot[40], comp[40];
};

static __global__ __launch_bounds__(256) void cdpr_publish_kernel(const PublishArgs a) {
  // one thread per OUTPUT element, so that a wave writes 256 contiguous bytes (the destination may be host memory behind
  // PCIe); the output is five robot-major blocks: three joint blocks of n columns, then 7 pose and 6 twist columns
  const uint32_t o = blockIdx.x * 256u + threadIdx.x;
  const uint32_t bn = a.batch * a.n, n3 = 3u * a.n;
  if (o < a.batch * a.width) {
    uint32_t r, j;
    if (o < 3u * bn) {
      const uint32_t f = o / bn, rem = o - f * bn;
      r = rem / a.n;
      j = f * a.n + (rem - r * a.n);
    } else if (o < 3u * bn + 7u * a.batch) {
      const uint32_t rem = o - 3u * bn;
      r = rem / 7u;
      j = n3 + (rem - r * 7u);
    } else {
      const uint32_t rem = o - 3u * bn - 7u * a.batch;
      r = rem / 6u;
      j = n3 + 7u + (rem - r * 6u);
    }
    a.out[o] = comp4(a.rows[(size_t)a.slot[j] * a.stride + r], a.comp[j]);
  }
  if (!a.done) return;  // staged read-out: the copy engine and the stream wait finish the job
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t prev = atomicAdd(a.arrivals, 1u);
    if (prev == gridDim.x - 1u) {
      *a.arrivals = 0u;
      __threadfence_system();
      __hip_atomic_store(a.done, a.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

}  // namespace cdpr
