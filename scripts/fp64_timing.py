"""precision = 64 beside fp32: us per world step (HIP events, one launch per step and ten steps per launch) for config 1
(one 4-cable robot), 4 096 x 4 and 4 096 x 8 with FK + TD."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
os.environ["CDPR_NO_GRAPH"] = "1"
import cdpr_simulation_amd as pkg, bench
for B, n, stages in ((1, 4, 0), (4096, 4, 0), (1, 8, 3), (4096, 8, 3), (65536, 8, 3)):
    model, pose, command, _ = bench.make_workload(pkg, B, n, 1235, 10)
    for prec in (32, 64):
        eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=stages, precision=prec), 0)
        eng.set_platform_state(pose7=pose); eng.set_velocity_command(command(0)); eng.update(50); eng.synchronize()
        out = []
        for spl in (1, 10):
            ts = []
            for rnd in range(5):
                eng.profile_begin(); eng.update(200, spl); ms, nl = eng.profile_end(); ts.append(ms / 200 * 1e3)
            out.append(f"spl={spl}: {np.median(ts):.2f} us/step")
        print(f"B={B} n={n} stages={stages} fp{prec} ({eng.mapping}): " + ", ".join(out), flush=True)
        eng.close()

# The hold branch (velocityEpsilon >= 0) in double beside the fp32 general path: steady commands above epsilon, then a third of
# the cables at or below it (their windows non-uniform: fits).
for B, n, stages in ((1, 8, 3), (4096, 8, 3), (65536, 8, 3)):
    model, pose, command, _ = bench.make_workload(pkg, B, n, 1235, 10)
    rng = np.random.default_rng(5)
    for prec in (32, 64):
        eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=stages, precision=prec, velocityEpsilon=0.001), 0)
        eng.set_platform_state(pose7=pose); eng.set_velocity_command(command(0)); eng.update(120); eng.synchronize()
        out = []
        for label in ("steady", "holding"):
            if label == "holding":
                cmd = command(0).copy(); low = rng.random(cmd.shape) < 0.33; cmd[low] = (0.0005 * rng.uniform(-1, 1, int(low.sum()))).astype(np.float32)
                eng.set_velocity_command(cmd); eng.update(120)
            ts = []
            for rnd in range(5):
                eng.profile_begin(); eng.update(200, 1); ms, nl = eng.profile_end(); ts.append(ms / 200 * 1e3)
            out.append(f"{label}: {np.median(ts):.2f} us/step")
        print(f"hold branch B={B} n={n} stages={stages} fp{prec} ({eng.mapping}): " + ", ".join(out), flush=True)
        eng.close()
