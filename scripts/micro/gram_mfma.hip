// gram_mfma.hip — measurement for BASELINE.json's north_star clause "MFMA used only for the batched J^T J / J^T w
// contractions where they are genuinely dense — choices evidenced by rocprof ... MFMA-busy against gfx950 peak".
//
// The only dense contraction of the step is the 6x6 normal matrix J^T J of one robot's 8x6 structure matrix (5 per step:
// 4 Newton iterations + the tension distribution).  Two ways to form it for 65 536 robots, same data, same launch
// geometry (1 024 single-wave workgroups = one wave per SIMD, as in the step kernel):
//
//   VALU   what the step kernel ships: one lane = one robot, cables in float2 pairs, 84 v_pk_fma_f32 + 21 adds per
//          robot (gram_partial + hsum of cdpr_step_kernel.hpp, included here unchanged);
//   MFMA   v_mfma_f32_4x4x1_16B_f32 (the batched f32 form: 16 independent 4x4 += 4x1 * 1x4 blocks per instruction):
//          four lanes = one robot, a 6x6 lower triangle = 3 tiles of 4x4, one rank-1 update per cable -> 24 MFMAs per 16
//          robots, 96 per 64 robots.  The MFMA side gets its operands ALREADY laid out element-per-lane for free; in the
//          step kernel the structure matrix lives robot-per-lane, so a real use would add a 64 x 48-value transpose
//          through LDS or DPP per Gram on top of what is measured here.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../cdpr-simulation_amd/csrc -o gram_mfma gram_mfma.hip
// Run:   ./gram_mfma            (prints one line per variant)
//        rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU -- ./gram_mfma
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "cdpr_step_kernel.hpp"

using namespace cdpr;

constexpr int kRobots = 65536;
constexpr int kReps = 512;  // Grams per robot per launch (in-kernel loop: launch overhead out of the picture)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// J of robot r, cable pair k, column c: jin[((k * 6 + c) * kRobots + r)] as float2 (coalesced per (k, c) row)
__global__ __launch_bounds__(64, 1) void gram_valu(const float2* __restrict__ jin, float* __restrict__ out, float eps) {
  const uint32_t r = blockIdx.x * 64u + threadIdx.x;
  v2f jac[4][6];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const float2 v = jin[(size_t)(k * 6 + c) * kRobots + r];
      jac[k][c] = (v2f){v.x, v.y};
    }
  float sum[21];
#pragma unroll
  for (int e = 0; e < 21; ++e) sum[e] = 0.f;
  for (int it = 0; it < kReps; ++it) {
    v2f acc[21];
    gram_partial<4>(jac, acc);  // the shipped code: 21 + 63 packed multiply-adds
#pragma unroll
    for (int e = 0; e < 21; ++e) sum[e] += hsum(acc[e]);  // 21 horizontal adds + keeping every Gram alive
#pragma unroll
    for (int c = 0; c < 6; ++c) jac[0][c] += splat(eps);  // the matrix changes between Grams, as between Newton iterations
  }
#pragma unroll
  for (int e = 0; e < 21; ++e) out[(size_t)e * kRobots + r] = sum[e];
}

// MFMA: lane l of a wave holds element (l & 3) of robot-group member (l >> 2): jm[((g * 8 + cable) * 2 + half) * 64 + l],
// g = group of 16 robots within the wave's 64 (4 groups), half 0 = columns 0..3, half 1 = columns 4,5,pad,pad.
__global__ __launch_bounds__(64, 1) void gram_mfma(const float* __restrict__ jm, float* __restrict__ out, float eps) {
  const uint32_t lane = threadIdx.x;
  const size_t base = (size_t)blockIdx.x * (4 * 8 * 2 * 64);
  float lo[4][8], hi[4][8];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      lo[g][c] = jm[base + ((size_t)(g * 8 + c) * 2 + 0) * 64 + lane];
      hi[g][c] = jm[base + ((size_t)(g * 8 + c) * 2 + 1) * 64 + lane];
    }
  f32x4 total[4][3];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int t = 0; t < 3; ++t) total[g][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < kReps; ++it) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 t00 = {0.f, 0.f, 0.f, 0.f}, t10 = t00, t11 = t00;
#pragma unroll
      for (int c = 0; c < 8; ++c) {  // one rank-1 update per cable and tile
        t00 = __builtin_amdgcn_mfma_f32_4x4x1f32(lo[g][c], lo[g][c], t00, 0, 0, 0);
        t10 = __builtin_amdgcn_mfma_f32_4x4x1f32(hi[g][c], lo[g][c], t10, 0, 0, 0);
        t11 = __builtin_amdgcn_mfma_f32_4x4x1f32(hi[g][c], hi[g][c], t11, 0, 0, 0);
      }
      total[g][0] += t00;
      total[g][1] += t10;
      total[g][2] += t11;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) lo[g][0] += eps;  // first term of every accumulation chain: nothing can be hoisted
  }
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) out[(((size_t)blockIdx.x * 4 + g) * 3 + t) * 256 + i * 64 + lane] = total[g][t][i];
}

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                     \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

int main() {
  std::vector<float> j((size_t)kRobots * 48);
  srand(7);
  for (auto& v : j) v = (float)rand() / (float)RAND_MAX - 0.5f;  // j[r][cable][col]
  // VALU layout
  std::vector<float2> jv((size_t)24 * kRobots);
  for (int r = 0; r < kRobots; ++r)
    for (int k = 0; k < 4; ++k)
      for (int c = 0; c < 6; ++c) jv[(size_t)(k * 6 + c) * kRobots + r] = make_float2(j[(size_t)r * 48 + (2 * k) * 6 + c], j[(size_t)r * 48 + (2 * k + 1) * 6 + c]);
  // MFMA layout
  std::vector<float> jmh((size_t)kRobots / 64 * 4 * 8 * 2 * 64);
  for (int wv = 0; wv < kRobots / 64; ++wv)
    for (int g = 0; g < 4; ++g)
      for (int c = 0; c < 8; ++c)
        for (int l = 0; l < 64; ++l) {
          const int r = wv * 64 + g * 16 + (l >> 2), i = l & 3;
          const size_t o = ((size_t)wv * 4 * 8 * 2 + (size_t)(g * 8 + c) * 2) * 64 + l;
          jmh[o] = j[(size_t)r * 48 + c * 6 + i];
          jmh[o + 64] = (i < 2) ? j[(size_t)r * 48 + c * 6 + 4 + i] : 0.f;
        }
  float2* d_jv;
  float *d_jm, *d_o1, *d_o2;
  CHECK(hipMalloc(&d_jv, jv.size() * sizeof(float2)));
  CHECK(hipMalloc(&d_jm, jmh.size() * sizeof(float)));
  CHECK(hipMalloc(&d_o1, (size_t)21 * kRobots * sizeof(float)));
  CHECK(hipMalloc(&d_o2, (size_t)kRobots / 64 * 4 * 3 * 256 * sizeof(float)));
  CHECK(hipMemcpy(d_jv, jv.data(), jv.size() * sizeof(float2), hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_jm, jmh.data(), jmh.size() * sizeof(float), hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const dim3 grid(kRobots / 64), block(64);
  for (int variant = 0; variant < 2; ++variant) {
    float best = 1e30f, sumt = 0.f;
    const int rounds = 12;
    for (int rnd = 0; rnd < rounds + 2; ++rnd) {
      CHECK(hipEventRecord(e0));
      if (variant == 0)
        hipLaunchKernelGGL(gram_valu, grid, block, 0, 0, d_jv, d_o1, 0.f);
      else
        hipLaunchKernelGGL(gram_mfma, grid, block, 0, 0, d_jm, d_o2, 0.f);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (rnd >= 2) {
        best = fminf(best, ms);
        sumt += ms;
      }
    }
    const double per_gram_ns = sumt / rounds * 1e6 / kReps;  // one Gram of all 65 536 robots
    const double flop = 2.0 * 21 * 8 * kRobots;                 // useful multiply-adds of the lower triangle
    printf("%s: %.3f us per Gram of 65536 robots (best %.3f), %.2f useful TFLOP/s\n", variant == 0 ? "VALU v_pk_fma_f32 (shipped)" : "MFMA 4x4x1_16B f32      ",
           per_gram_ns * 1e-3, best * 1e6 / kReps * 1e-3, flop / per_gram_ns * 1e-3);
  }
  // same numbers out of both (robot 5, entry (3,1)): VALU sum[e = 3*4/2+1 = 7]; MFMA: wave 0, group 0, tile t00, vgpr i = 3 (row), lane 4*5 + 1
  std::vector<float> o1((size_t)21 * kRobots), o2((size_t)kRobots / 64 * 4 * 3 * 256);
  CHECK(hipMemcpy(o1.data(), d_o1, o1.size() * sizeof(float), hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(o2.data(), d_o2, o2.size() * sizeof(float), hipMemcpyDeviceToHost));
  const float a = o1[(size_t)7 * kRobots + 5], b = o2[((size_t)0 * 3 + 0) * 256 + 3 * 64 + (4 * 5 + 1)];
  printf("check: robot 5, G[3][1] x %d: VALU %.6f  MFMA %.6f  %s\n", kReps, a, b, fabsf(a - b) <= 1e-3f * fabsf(a) + 1e-4f ? "agree" : "DIFFER");
  return 0;
}
