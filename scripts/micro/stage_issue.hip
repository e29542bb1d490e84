// stage_issue.hip — what does it cost ONE wave (one per SIMD, as the step kernels run) to bring 35 float4 record slots
// per lane in?  Three ways, each stamped with s_memrealtime (100 MHz) at entry / after the last issue / after the data
// is usable, reported as medians over the waves of a 256-block launch (B = 16 384 columns):
//   dma        35 x buffer_load_dwordx4 ... lds (what cdpr_gen_step_kernel's gen_stage_records does)
//   dma_valu   the same, each followed by ~40 dependent-free v_fma (is the issue cost a stall that arithmetic hides?)
//   regs       35 x buffer_load_dwordx4 into registers, then 35 x ds_write_b128
// Build: hipcc --offload-arch=gfx950 -O3 -o stage_issue stage_issue.hip ; run: ./stage_issue
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int kSlots = 35;
using lds_ptr = __attribute__((address_space(3))) void*;

__device__ inline void stamp(unsigned long long* st, int i) {
  __builtin_amdgcn_sched_barrier(0);
  if (threadIdx.x == 0) st[blockIdx.x * 4 + i] = __builtin_amdgcn_s_memrealtime();
  __builtin_amdgcn_sched_barrier(0);
}

template <int MODE>
__global__ __launch_bounds__(64, 1) void k(const float4* __restrict__ src, float* __restrict__ out, uint32_t B, uint32_t bytes, unsigned long long* st, float x) {
  __shared__ float4 stage[kSlots][64];
  const uint32_t lane = threadIdx.x, r = blockIdx.x * 64u + lane;
  stamp(st, 0);
  float acc = x;
  if (MODE == 2) {
    float4 v[kSlots];
#pragma unroll
    for (int i = 0; i < kSlots; ++i) v[i] = src[(size_t)i * B + r];
    stamp(st, 1);
#pragma unroll
    for (int i = 0; i < kSlots; ++i) stage[i][lane] = v[i];
  } else {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(src), 0, (int)bytes, 0x00020000);
    float f0 = x, f1 = x + 1.f, f2 = x + 2.f, f3 = x + 3.f;
#pragma unroll
    for (int i = 0; i < kSlots; ++i) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)&stage[i][0], 16, r * 16u, (uint32_t)i * B * 16u, 0, 0);
      if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 10; ++j) f0 = fmaf(f0, x, 1.f), f1 = fmaf(f1, x, 1.f), f2 = fmaf(f2, x, 1.f), f3 = fmaf(f3, x, 1.f);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    acc += (f0 + f1) + (f2 + f3);
    stamp(st, 1);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  stamp(st, 2);
#pragma unroll
  for (int i = 0; i < kSlots; ++i) acc += stage[i][(lane + 1) & 63].x;
  out[r] = acc;
  stamp(st, 3);
}

int main() {
  const uint32_t B = 16384;
  const size_t bytes = (size_t)kSlots * B * 16;
  float4* a; float* o; unsigned long long* st;
  const int nbuf = 40;  // 40 x 9.2 MB: every launch reads memory no earlier launch left in a cache
  CK(hipMalloc(&a, bytes * nbuf)); CK(hipMalloc(&o, (size_t)B * 4)); CK(hipMalloc(&st, (B / 64) * 4 * 8));
  CK(hipMemset(a, 0, bytes * nbuf));
  const char* names[3] = {"dma", "dma_valu", "regs"};
  for (int mode = 0; mode < 3; ++mode) {
    std::vector<double> issue, ready, total;
    for (int it = 0; it < 12; ++it) {
      float4* buf = a + (size_t)((it * 3 + mode) % nbuf) * (bytes / 16);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(B / 64), dim3(64), 0, 0, buf, o, B, (uint32_t)bytes, st, 0.5f);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(B / 64), dim3(64), 0, 0, buf, o, B, (uint32_t)bytes, st, 0.5f);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(B / 64), dim3(64), 0, 0, buf, o, B, (uint32_t)bytes, st, 0.5f);
      CK(hipDeviceSynchronize());
      std::vector<unsigned long long> h((B / 64) * 4);
      CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
      if (it < 2) continue;
      for (uint32_t w = 0; w < B / 64; ++w) {
        issue.push_back((h[w * 4 + 1] - h[w * 4]) * 0.01), ready.push_back((h[w * 4 + 2] - h[w * 4]) * 0.01), total.push_back((h[w * 4 + 3] - h[w * 4]) * 0.01);
      }
    }
    auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("%-9s issue done %.2f us   data usable %.2f us   read back + store %.2f us (medians over waves)\n", names[mode], med(issue), med(ready), med(total));
  }
  return 0;
}
