// valu_issue.hip — what one wave's instruction stream costs on a gfx950 SIMD: cycles per instruction (s_memtime) of
// dependent chains and independent streams of the instructions the step kernels are made of, with one wave per SIMD
// (and, second table, two waves per SIMD running the same stream).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip      Run: ./valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>

typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int kIters = 512;
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ __launch_bounds__(512) void probe(uint64_t* out, float seed) {
  float a = seed + threadIdx.x, b = seed * 0.5f, c = 1.0001f, d = 0.3f;
  v2f pa = {a, a + 1.f}, pb = {b, b + 1.f}, pc = {c, c}, pd = {d, d};
  v2f q0 = pa, q1 = pb, q2 = pc, q3 = pd, q4 = pa + pb, q5 = pb + pc, q6 = pc + pd, q7 = pa + pd;
  float s0 = a, s1 = b, s2 = c, s3 = d, s4 = a + b, s5 = b + c, s6 = c + d, s7 = a + d;
  uint64_t t0 = 0, t1 = 0, r0 = 0, r1 = 0;
  for (int warm = 0; warm < 2; ++warm) {
    asm volatile("s_waitcnt lgkmcnt(0)\n s_memrealtime %1\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    for (int it = 0; it < kIters; ++it) {
      if (KIND == 0) { REP64(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s0) : "v"(c), "v"(d));) }  // dependent plain fma
      if (KIND == 1) { REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                                         : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7) : "v"(c), "v"(d));) }
      if (KIND == 2) { REP64(asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(q0) : "v"(pc), "v"(pd));) }  // dependent packed fma
      if (KIND == 3) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9"
                                         : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7) : "v"(pc), "v"(pd));) }
      if (KIND == 4) { REP64(asm volatile("v_rsq_f32 %0, %0" : "+v"(s0));) }                               // dependent rsq
      if (KIND == 5) { REP8(asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n v_rsq_f32 %4, %4\n v_rsq_f32 %5, %5\n v_rsq_f32 %6, %6\n v_rsq_f32 %7, %7"
                                         : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7));) }
      if (KIND == 6) { REP64(asm volatile("v_mul_f32 v200, v200, %1\n v_pk_fma_f32 %0, v[200:201], %2, %0 op_sel_hi:[0,1,1]" : "+v"(q0) : "v"(c), "v"(pd) : "v200", "v201");) }  // scalar -> broadcast packed; the scalar chain is the dependent one (128)
      if (KIND == 7) { REP64(asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(q0) : "v"(pc));) }              // dependent packed mul
      if (KIND == 8) { REP8(REP8(asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %4, %5" : "+v"(q0), "+v"(s0) : "v"(pc), "v"(pd), "v"(c), "v"(d));)) }  // two independent chains: one packed one plain (128)
      if (KIND == 9) { REP8(REP8(asm volatile("v_rsq_f32 %0, %0\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %4, %4, %2, %3" : "+v"(s0), "+v"(q0) : "v"(pc), "v"(pd), "v"(q1));)) }  // rsq chain + 2 independent packed (192)
      if (KIND == 10) { REP64(asm volatile("v_add_f32 v200, v202, v203\n v_pk_fma_f32 v[202:203], v[200:201], %0, v[202:203] op_sel_hi:[0,1,1]" ::"v"(pc) : "v200", "v201", "v202", "v203");) }  // hsum -> packed -> hsum ... fully dependent (128)
    }
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
  }
  float sink = s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7 + q0.x + q1.x + q2.x + q3.x + q4.x + q5.x + q6.x + q7.x + q0.y;
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = r1 - r0; }
  if (sink == 12345.678f) out[blockIdx.x * 2 + 1] = 1;
}

template <int KIND>
static void run(const char* name, int instr_per_it, uint64_t* d, int threads) {
  const int blocks = 256;
  hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(threads), 0, 0, d, 1.5f);
  hipDeviceSynchronize();
  std::vector<uint64_t> h(blocks * 2);
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  double sum = 0, rsum = 0;
  for (int i = 0; i < blocks; ++i) { sum += (double)h[2 * i]; rsum += (double)h[2 * i + 1]; }
  const double n = (double)kIters * instr_per_it;
  printf("  %-58s %6.2f s_memtime ticks/instr  %6.2f ns/instr\n", name, sum / blocks / n, rsum / blocks * 10.0 / n);
}

int main() {
  uint64_t* d;
  hipMalloc(&d, 8192 * 16);
  for (int threads : {64, 256, 512, 1024}) {
    printf("256 workgroups of %d threads (first wave of each timed; %s):\n", threads, threads == 64 ? "one wave per CU" : threads == 256 ? "one wave per SIMD" : threads == 512 ? "two waves per SIMD" : "four waves per SIMD");
    run<0>("v_fma_f32, dependent chain", 64, d, threads);
    run<1>("v_fma_f32, 8 independent chains", 64, d, threads);
    run<2>("v_pk_fma_f32, dependent chain", 64, d, threads);
    run<3>("v_pk_fma_f32, 8 independent chains", 64, d, threads);
    run<7>("v_pk_mul_f32, dependent chain", 64, d, threads);
    run<4>("v_rsq_f32, dependent chain", 64, d, threads);
    run<5>("v_rsq_f32, 8 independent chains", 64, d, threads);
    run<6>("v_mul_f32 -> v_pk_fma_f32 (op_sel broadcast), dependent", 128, d, threads);
    run<8>("v_pk_fma_f32 chain + v_fma_f32 chain interleaved", 128, d, threads);
    run<9>("v_rsq_f32 chain + 2 v_pk_fma_f32 (one dependent chain)", 192, d, threads);
    run<10>("v_add_f32 (halves) -> v_pk_fma_f32, dependent", 128, d, threads);
  }
  // whole-launch wall time of single-wave workgroups (events): 1 024 = one wave per SIMD, 2 048 = two, 4 096 = four
  {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("whole launch, single-wave workgroups, %d x 64 instructions per wave (hipEvent ms):\n", kIters);
    for (int kind = 0; kind < 3; ++kind)
      for (int blocks : {256, 1024, 2048, 4096}) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
          hipEventRecord(e0);
          if (kind == 0) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(64), 0, 0, d, 1.5f);
          if (kind == 1) hipLaunchKernelGGL(probe<3>, dim3(blocks), dim3(64), 0, 0, d, 1.5f);
          if (kind == 2) hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(64), 0, 0, d, 1.5f);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
          float ms;
          hipEventElapsedTime(&ms, e0, e1);
          best = ms < best ? ms : best;
        }
        printf("  %-34s %5d workgroups: %7.2f us\n", kind == 0 ? "v_fma_f32, 8 independent chains" : kind == 1 ? "v_pk_fma_f32, 8 independent chains" : "v_pk_fma_f32, dependent chain", blocks, best * 1e3f);
      }
  }
  return 0;
}
