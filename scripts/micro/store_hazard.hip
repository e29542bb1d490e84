// store_hazard.hip — does gfx950 need wait states between a 128-bit buffer store and a VALU write of its data registers
// when the store's soffset operand is an SGPR?
//
// The ISA manuals list "VMEM store of more than 64 bits followed by a VALU write of the VGPRs holding the write data" as a
// hazard that needs manual wait states, with the note that BUFFER_STORE_DWORDX3/X4 are exempt when soffset is an SGPR; LLVM's
// hazard recognizer (GCNHazardRecognizer::createsVALUHazard) follows the note and inserts no s_nop in that case.  Round 2's
// experiment "one buffer descriptor per buffer, the row as the scalar offset" (cdpr_onestep_kernel.hpp) made the FK kernels
// fail bit-identity; round 5 traced the deviation to 64-byte granules (lanes 12-15 of every 16) of ring-row stores whose
// data registers the next v_cndmask overwrites with zero wait states.  This program measures the
// hazard in isolation: store v[10:13] = A with an SGPR (or immediate) soffset, `NOPS` wait states, then v_mov v11 = B;
// every lane whose y component arrives as B saw the overwrite.
//
//   hipcc --offload-arch=gfx950 -O2 -o store_hazard scripts/micro/store_hazard.hip && ./store_hazard
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define NOP0 ""
#define NOP1 "s_nop 0\n"
#define NOP2 "s_nop 1\n"
#define NOP3 "s_nop 2\n"

// PRE: 128-bit stores issued right before the measured one (to other rows): how busy the store path is
template <int NOPS, bool SGPR_OFF, int PRE, bool NT>
__global__ __launch_bounds__(64) void probe(unsigned int* out, unsigned int* sink, unsigned int stride_bytes) {
  const unsigned long long base = (unsigned long long)out;
  const u32x4 rsrc = {(unsigned)base, (unsigned)(base >> 32) & 0xffffu, 0x7fffffffu, 0x00020000u};
  const __amdgpu_buffer_rsrc_t rsink = __builtin_amdgcn_make_buffer_rsrc(sink, 0, 0x7fffffff, 0x00020000);
  const unsigned off = threadIdx.x * 16u;
  const unsigned a = 0xA0000000u | threadIdx.x, b = 0xB0000000u | threadIdx.x;
  const unsigned soff = blockIdx.x * stride_bytes;  // one row per workgroup: wave-uniform, an SGPR
  if (PRE > 0) {
    for (int i = 0; i < PRE; ++i) {
      const u32x4 d = {a, a, a, a};
      __builtin_amdgcn_raw_buffer_store_b128(d, rsink, off, (blockIdx.x * 64u + (unsigned)i) * 1024u, 2);
    }
  }
#define BODY(NOPSTR, STORE)                                                                                  \
  asm volatile("v_mov_b32 v10, %[a]\n v_mov_b32 v11, %[a]\n v_mov_b32 v12, %[a]\n v_mov_b32 v13, %[a]\n"      \
               "s_nop 4\n" STORE NOPSTR "v_mov_b32 v11, %[b]\n s_waitcnt vmcnt(0)\n"                          \
               :                                                                                             \
               : [a] "v"(a), [b] "v"(b), [off] "v"(off), [rsrc] "s"(rsrc), [soff] "s"(soff), [offi] "v"(off + soff) \
               : "v10", "v11", "v12", "v13", "memory")
#define ST_S_NT "buffer_store_dwordx4 v[10:13], %[off], %[rsrc], %[soff] offen nt\n"
#define ST_S "buffer_store_dwordx4 v[10:13], %[off], %[rsrc], %[soff] offen\n"
#define ST_I_NT "buffer_store_dwordx4 v[10:13], %[offi], %[rsrc], 0 offen nt\n"
#define ST_I "buffer_store_dwordx4 v[10:13], %[offi], %[rsrc], 0 offen\n"
#define PICK(NOPSTR)                          \
  do {                                        \
    if (SGPR_OFF && NT) BODY(NOPSTR, ST_S_NT); \
    else if (SGPR_OFF) BODY(NOPSTR, ST_S);     \
    else if (NT) BODY(NOPSTR, ST_I_NT);        \
    else BODY(NOPSTR, ST_I);                   \
  } while (0)
  if (NOPS == 0) PICK(NOP0);
  else if (NOPS == 1) PICK(NOP1);
  else if (NOPS == 2) PICK(NOP2);
  else PICK(NOP3);
}

template <int NOPS, bool SGPR_OFF, int PRE, bool NT>
static int run(unsigned int* d_out, unsigned int* d_sink, int blocks, int reps) {
  const unsigned stride = 1024u;
  std::vector<unsigned> h((size_t)blocks * 256);
  long bad = 0, total = 0;
  unsigned long long lane_mask = 0ull;
  for (int r = 0; r < reps; ++r) {
    (void)hipMemset(d_out, 0, (size_t)blocks * stride);
    hipLaunchKernelGGL((probe<NOPS, SGPR_OFF, PRE, NT>), dim3(blocks), dim3(64), 0, 0, d_out, d_sink, stride);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), d_out, (size_t)blocks * stride, hipMemcpyDeviceToHost);
    for (int w = 0; w < blocks; ++w)
      for (int l = 0; l < 64; ++l) {
        const unsigned y = h[(size_t)w * 256 + l * 4 + 1];
        ++total;
        if (y != (0xA0000000u | (unsigned)l)) {
          ++bad;
          lane_mask |= 1ull << l;
        }
      }
  }
  printf("soffset %-4s %-3s  wait states %d  stores before %2d : %8ld of %ld lanes saw the overwrite  (lanes 0x%016llx)\n", SGPR_OFF ? "SGPR" : "imm", NT ? "nt" : "",
         NOPS, PRE, bad, total, lane_mask);
  return bad != 0;
}

int main() {
  const int blocks = 4096, reps = 3;
  unsigned int *d_out = nullptr, *d_sink = nullptr;
  if (hipMalloc(&d_out, (size_t)blocks * 1024) != hipSuccess || hipMalloc(&d_sink, (size_t)blocks * 64 * 1024) != hipSuccess) {
    fprintf(stderr, "no GPU / out of memory\n");
    return 2;
  }
  run<0, true, 0, true>(d_out, d_sink, blocks, reps);
  run<1, true, 0, true>(d_out, d_sink, blocks, reps);
  run<2, true, 0, true>(d_out, d_sink, blocks, reps);
  run<3, true, 0, true>(d_out, d_sink, blocks, reps);
  run<0, true, 4, true>(d_out, d_sink, blocks, reps);
  run<1, true, 4, true>(d_out, d_sink, blocks, reps);
  run<2, true, 4, true>(d_out, d_sink, blocks, reps);
  run<0, true, 16, true>(d_out, d_sink, blocks, reps);
  run<1, true, 16, true>(d_out, d_sink, blocks, reps);
  run<2, true, 16, true>(d_out, d_sink, blocks, reps);
  run<0, true, 16, false>(d_out, d_sink, blocks, reps);
  run<1, true, 16, false>(d_out, d_sink, blocks, reps);
  run<0, false, 0, true>(d_out, d_sink, blocks, reps);
  run<1, false, 0, true>(d_out, d_sink, blocks, reps);
  run<2, false, 0, true>(d_out, d_sink, blocks, reps);
  run<0, false, 16, true>(d_out, d_sink, blocks, reps);
  run<1, false, 16, true>(d_out, d_sink, blocks, reps);
  run<2, false, 16, true>(d_out, d_sink, blocks, reps);
  (void)hipFree(d_out);
  (void)hipFree(d_sink);
  return 0;
}
