"""Host in the loop, one world step at a time — what a Gazebo-style caller pays per step (the reference's own use:
ConnectWorldUpdateBegin -> update() every 1 ms step, messages published every step).  Two levels:
  C-ABI   cdpr_update(1) + cdpr_get_observables (one gather launch into a pinned host image, host spins on its completion
          word), against the two separate getters (five gather / copy / wait rounds) and against no read-out at all
  facade  CdprGazeboPlugin.update() with a subscriber on jointStates / platformPose and a 100 Hz sine publisher
for the reference's robot (1 x 4 cables) and for batches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cdpr_simulation_amd as pkg

def c_abi(batch, n, steps=3000):
    model = pkg.cube_model() if n == 4 else pkg.eight_cable_model()
    eng = pkg.Engine(pkg.Config(model=model, batch=batch, stages=0 if n == 4 else 3), 0)
    eng.set_velocity_command(np.full(n, 0.01, dtype=np.float32))
    for _ in range(200):
        eng.update(1); eng.joint_states(); eng.platform_state()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.update(1); eng.joint_states(); eng.platform_state()
        ts.append((time.perf_counter() - t0) / steps * 1e6)
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.update(1); eng.synchronize()
    only = (time.perf_counter() - t0) / steps * 1e6
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.update(1); eng.observables()
    one = (time.perf_counter() - t0) / steps * 1e6
    eng.close()
    return float(np.median(ts)), only, one

def facade(steps=3000):
    from cdpr_simulation_amd.messages import Joy
    bus = pkg.TopicBus(); got = [0]
    plug = pkg.CdprGazeboPlugin(bus); plug.Load(pkg.Config())
    bus.subscribe("jointStates", lambda m: got.__setitem__(0, got[0] + 1)); bus.subscribe("platformPose", lambda m: None)
    pub = bus.advertise("jointVelocities")
    for k in range(200): plug.update()
    t0 = time.perf_counter()
    for k in range(steps):
        if k % 10 == 0: pub(Joy(axes=[np.float32(0.05 * np.sin(2 * np.pi * 0.1 * k * 1e-3))] * 4))
        plug.update()
    return (time.perf_counter() - t0) / steps * 1e6, got[0]

if __name__ == "__main__":
    for batch, n in ((1, 4), (1, 8), (4096, 4), (65536, 8)):
        full, only, one = c_abi(batch, n, 3000 if batch <= 4096 else 600)
        print(f"C-ABI  {batch:6d} x {n}: update(1) + cdpr_get_observables {one:7.1f} us/step (real-time factor of a 1 ms step {1000.0 / one:6.1f})   "
              f"update(1) + get_joint_states + get_platform_state {full:7.1f}   update(1) + synchronize only {only:6.1f}")
    us, n_msgs = facade()
    print(f"facade      1 x 4: CdprGazeboPlugin.update() with subscribers and a 100 Hz sine publisher {us:8.1f} us/step ({n_msgs} JointState messages)   real-time factor {1000.0 / us:7.1f}")
