"""Does splitting one batch into independent chains on separate HIP streams hide the kernel-to-kernel boundary?
One handle of 65 536 robots (one launch per step) against K handles of 65 536 / K robots on the same GPU, each on its
own stream (robots are independent, so the chains never synchronise).  Wall clock over `steps` steps, interleaved
rounds, median + min."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cdpr_simulation_amd as pkg
import bench

os.environ["CDPR_MAPPING"] = "1"
B, n, steps = int(os.environ.get("PROBE_B", 65536)), 8, 400
model, pose, command, n_cmd = bench.make_workload(pkg, B, n, 1235, 10)
cfg = pkg.Config(model=model, batch=B, stages=3, mapping=pkg._abi.MAP_LANE_PER_ROBOT)
variants = {}
for k in (1, 2, 4, 8):
    e = pkg.Engine(cfg, 0) if k == 1 else pkg.ShardedEngine(cfg, devices=[0] * k)
    e.set_platform_state(pose7=pose)
    e.set_velocity_command(command(0))
    e.update(100)
    e.synchronize()
    variants[k] = e
res = {k: [] for k in variants}
for rnd in range(9):
    for k, e in variants.items():
        t0 = time.perf_counter()
        if k == 1:
            e.update(steps)
        else:
            for s in range(0, steps, 10):  # interleave the chains 10 steps at a time, as a real driver would
                e.update(10)
        e.synchronize()
        res[k].append((time.perf_counter() - t0) / steps * 1e6)
for k, v in res.items():
    print(f"B={B} chains={k}: {np.median(v):.2f} us/step (min {min(v):.2f})  {B / np.median(v) * 1e6:.3e} state-steps/s", flush=True)
