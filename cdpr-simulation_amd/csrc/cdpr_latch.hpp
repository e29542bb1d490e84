// cdpr_latch.hpp — per-robot command arrival on the register-resident path (cdpr_set_*_command_masked on handles created
// with per_robot_commands; PLG.cpp:206-219 runs per model).  The general controller path has its own latch kernel
// (cdpr_gen_latch_kernel, cdpr_general_step.hpp).
#pragma once
#include "cdpr_step_kernel.hpp"

namespace cdpr {

// A Joy batch reaching some robots of a per-robot handle (every robot keeps one Pid record, the active mode's:
// cdpr_step_kernel.hpp, StepArgs::meta).  One thread per robot: the Joy's row becomes the robot's active target,
// and entering the mode from the other one resets its Pid (JFC.cpp:101-103,113-115): integral 0 and call count 0, so the
// next Pid::update is the "first" and returns 0 (Pid.cpp:123-126).  The derivative ring needs no clearing: derive()
// returns 0 until nbuf new samples are in, and by then they have overwritten every slot the weights reach.
struct LatchFastArgs {
  const uint8_t* mask;   // uint8[B], or nullptr = every robot
  uint8_t* meta;         // StepArgs::meta
  const float* pending;  // float[B][n]
  float* target;         // float[B][n]: the robots' active target rows (StepArgs::cmd of the PR kernels)
  float4* hot;           // first integral row of the state: slot P + 5 * pairs
  size_t stride;
  uint32_t batch, n, hot_rows;
  uint32_t new_mode;     // kMetaPosition / kMetaVelocity
};

static __global__ __launch_bounds__(256) void cdpr_latch_fast_kernel(const LatchFastArgs a) {
  const uint32_t r = blockIdx.x * 256u + threadIdx.x;
  if (r >= a.batch) return;
  if (a.mask && !a.mask[r]) return;
  for (uint32_t i = 0; i < a.n; ++i) a.target[(size_t)r * a.n + i] = a.pending[(size_t)r * a.n + i];
  uint32_t m = a.meta[r];
  if (a.new_mode == kMetaForce) {  // setForce (JFC.h:92-95) resets nothing; leaving Force mode resets the Pid entered
    m = (m & ~kMetaModeMask) | kMetaForce;
  } else if ((m & kMetaModeMask) != a.new_mode) {
    for (uint32_t g = 0; g < a.hot_rows; ++g) a.hot[(size_t)g * a.stride + r] = make_float4(0.f, 0.f, 0.f, 0.f);
    m = a.new_mode;  // call count 0
  }
  a.meta[r] = (uint8_t)m;
}

// cdpr_update_scheduled_kind served as a chain of launches: one thread waits for the mailbox word of the next batch in front
// of the batch's latch (what sched_wait does inside a launch that carries the schedule itself).
static __global__ void cdpr_mailbox_wait_kernel(const uint32_t* word, uint32_t* fault) { mailbox_wait(word, fault); }

}  // namespace cdpr
