// rowcopy.hip — memory floor of the step kernel's access pattern on MI355X: every lane reads R float4 rows and
// writes W float4 rows of a [rows][B] slot array (row stride B*16 bytes), one launch per "step", back to back.
// Build: hipcc --offload-arch=gfx950 -O3 -o rowcopy rowcopy.hip ; run: ./rowcopy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int R, int W, int NT>
__global__ __launch_bounds__(64) void rowcopy(const float4* __restrict__ src, float4* __restrict__ dst, float4* __restrict__ obs, uint32_t B) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= B) return;
  float4 v[R];
#pragma unroll
  for (int i = 0; i < R; ++i) v[i] = src[(size_t)i * B + r];
  float4 acc = v[0];
#pragma unroll
  for (int i = 1; i < R; ++i) { acc.x += v[i].x; acc.y += v[i].w; }
#pragma unroll
  for (int i = 0; i < W; ++i) {
    float4 o = v[i % R]; o.x += acc.x * 1e-9f;
    float4* p = (i < R ? dst : obs) + (size_t)(i < R ? i : i - R) * B + r;
    typedef float f4 __attribute__((ext_vector_type(4)));
    if (NT) __builtin_nontemporal_store((f4){o.x, o.y, o.z, o.w}, reinterpret_cast<f4*>(p)); else *p = o;
  }
}

template <int R, int W, int NT>
int run(const char* name, uint32_t B, int block) {
  float4 *a, *b, *o;
  CK(hipMalloc(&a, (size_t)R * B * 16)); CK(hipMalloc(&b, (size_t)R * B * 16)); CK(hipMalloc(&o, (size_t)(W - R + 1) * B * 16));
  CK(hipMemset(a, 0, (size_t)R * B * 16)); CK(hipMemset(b, 0, (size_t)R * B * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipStream_t s; CK(hipStreamCreate(&s));
  const int iters = 400;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) {
      hipLaunchKernelGGL((rowcopy<R, W, NT>), dim3((B + block - 1) / block), dim3(block), 0, s, a, a, o, B);  // in place like the engine
    }
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep == 2) {
      double us = ms * 1e3 / iters, bytes = (double)(R + W) * B * 16;
      printf("%-28s B=%u block=%d: %.2f us/launch, %.1f MB moved, %.2f TB/s\n", name, B, block, us, bytes / 1e6, bytes / us / 1e6);
    }
  }
  CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(o));
  return 0;
}

int main() {
  // step kernel at n = 8, FK+TD: reads 29 state rows + 2 command rows, writes 29 state + 10 observable rows
  run<31, 39, 0>("read31 write39 plain", 65536, 64);
  run<31, 39, 1>("read31 write39 nt", 65536, 64);
  run<31, 39, 0>("read31 write39 plain", 65536, 256);
  run<31, 31, 0>("read31 write31 plain", 65536, 64);
  run<31, 10, 0>("read31 write10 plain", 65536, 64);
  run<31, 1, 0>("read31 write1 plain", 65536, 64);
  run<8, 39, 0>("read8 write39 plain", 65536, 64);
  run<31, 39, 0>("read31 write39 plain", 524288, 64);
  run<31, 39, 1>("read31 write39 nt", 524288, 64);
  run<1, 1, 0>("read1 write1 (launch floor)", 65536, 64);
  return 0;
}
