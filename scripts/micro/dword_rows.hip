// dword_rows.hip — calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the general controller path's access pattern
// (MI355X_MICROARCH.md: FETCH_SIZE is calibrated for 16 B per lane only, "other access widths are uncalibrated: calibrate
// on a known byte count in your own access pattern").  Four kernels over a [rows][B] array, B = 65 536 columns, each
// moving exactly 64 rows x B x 4 B = 16.8 MB per launch in one direction:
//   rd_dword      every lane loads 64 dword rows (buffer_load_dword), sums them, writes ONE dword (so the loads stay)
//   rd_dword_lds  the same rows global -> LDS by buffer_load_dword ... lds, read back from LDS
//   wr_dword      every lane stores 64 dword rows
//   rd_x4         16 float4 rows per lane (the calibrated pattern: the counter should read half)
// Run each under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes): ./dword_rows <kernel> <launches>
// Build: hipcc --offload-arch=gfx950 -O3 -o dword_rows dword_rows.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int kRows = 64;

__global__ __launch_bounds__(64) void rd_dword(const float* __restrict__ src, float* __restrict__ out, uint32_t B) {
  const uint32_t r = blockIdx.x * 64u + threadIdx.x;
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < kRows; ++i) acc += src[(size_t)i * B + r];
  out[r] = acc;
}
__global__ __launch_bounds__(64) void rd_dword_lds(const float* __restrict__ src, float* __restrict__ out, uint32_t B, uint32_t bytes) {
  __shared__ float stage[kRows][64];
  const uint32_t r = blockIdx.x * 64u + threadIdx.x;
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (int)bytes, 0x00020000);
#pragma unroll
  for (int i = 0; i < kRows; ++i)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)&stage[i][0], 4, r * 4u, (uint32_t)i * B * 4u, 0, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < kRows; ++i) acc += stage[i][threadIdx.x];
  out[r] = acc;
}
__global__ __launch_bounds__(64) void wr_dword(float* __restrict__ dst, uint32_t B, float v) {
  const uint32_t r = blockIdx.x * 64u + threadIdx.x;
#pragma unroll
  for (int i = 0; i < kRows; ++i) dst[(size_t)i * B + r] = v + (float)i;
}
__global__ __launch_bounds__(64) void rd_x4(const float4* __restrict__ src, float* __restrict__ out, uint32_t B) {
  const uint32_t r = blockIdx.x * 64u + threadIdx.x;
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < kRows / 4; ++i) {
    const float4 v = src[(size_t)i * B + r];
    acc += v.x + v.w;
  }
  out[r] = acc;
}

int main(int argc, char** argv) {
  const char* which = argc > 1 ? argv[1] : "rd_dword";
  const int launches = argc > 2 ? atoi(argv[2]) : 20;
  const uint32_t B = 65536;
  const size_t bytes = (size_t)kRows * B * 4;
  float *a, *o;
  // 20 buffers of 16.8 MB used round robin (336 MB: past the 256 MB Infinity Cache, so re-reads are real fetches)
  const int nbuf = 20;
  CK(hipMalloc(&a, bytes * nbuf)); CK(hipMalloc(&o, (size_t)B * 4)); CK(hipMemset(a, 0, bytes * nbuf));
  CK(hipDeviceSynchronize());
  for (int i = 0; i < launches; ++i) {
    float* buf = a + (size_t)(i % nbuf) * (bytes / 4);
    if (!strcmp(which, "rd_dword")) hipLaunchKernelGGL(rd_dword, dim3(B / 64), dim3(64), 0, 0, buf, o, B);
    else if (!strcmp(which, "rd_dword_lds")) hipLaunchKernelGGL(rd_dword_lds, dim3(B / 64), dim3(64), 0, 0, buf, o, B, (uint32_t)bytes);
    else if (!strcmp(which, "wr_dword")) hipLaunchKernelGGL(wr_dword, dim3(B / 64), dim3(64), 0, 0, buf, B, 1.f);
    else hipLaunchKernelGGL(rd_x4, dim3(B / 64), dim3(64), 0, 0, reinterpret_cast<const float4*>(buf), o, B);
  }
  CK(hipDeviceSynchronize());
  printf("%s: %d launches, %.2f MB per launch in the calibrated direction\n", which, launches, bytes / 1e6);
  return 0;
}
