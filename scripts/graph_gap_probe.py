"""Does the launch-to-launch gap of a one-launch-per-step run depend on hipGraph replay?  Fast path at 65 536 x 8 with and without
CDPR_NO_GRAPH, general path (no graphs there) beside it: us per step by HIP events over 1000 steps."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
import cdpr_simulation_amd as pkg, bench
B = 65536
for eps, label in ((-1.0, "fast path"), (0.001, "general path")):
    model, pose, command, _ = bench.make_workload(pkg, B, 8, 1235, 10)
    eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=3, velocityEpsilon=eps), 0)
    eng.set_platform_state(pose7=pose); eng.set_velocity_command(command(0)); eng.update(200); eng.synchronize()
    ts = []
    for rnd in range(5):
        eng.profile_begin(); eng.update(1000); ms, nl = eng.profile_end(); ts.append(ms / 1000 * 1e3)
    print(f"CDPR_NO_GRAPH={os.environ.get('CDPR_NO_GRAPH', '0')} {label} ({eng.kernel_name}): {np.median(ts):.2f} us/step", flush=True)
    eng.close()
''' % ROOT
for rep in range(2):
    for ng in ("0", "1"):
        subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CDPR_NO_GRAPH=ng))
