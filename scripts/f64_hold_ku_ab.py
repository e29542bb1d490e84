"""fp64 hold branch, role-split kernel (LDS build): cables per pass of the controller wave's loops (build variants, CDPR_LIB)."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
os.environ["CDPR_NO_GRAPH"] = "1"
import cdpr_simulation_amd as pkg, bench
for B in (1, 4096, 16384):
    model, pose, command, _ = bench.make_workload(pkg, B, 8, 1235, 10)
    eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=3, precision=64, velocityEpsilon=0.001), 0)
    eng.set_platform_state(pose7=pose); eng.set_velocity_command(command(0)); eng.update(120); eng.synchronize()
    ts = []
    for rnd in range(5):
        eng.profile_begin(); eng.update(200, 1); ms, nl = eng.profile_end(); ts.append(ms / 200 * 1e3)
    print(f"  B={B}: {np.median(ts):.2f} us/step", flush=True)
    eng.close()
''' % ROOT
for lib in ("libcdpr_hip.so", "libcdpr_hip_ku4.so", "libcdpr_hip_ku2.so"):
    print(lib, flush=True)
    env = dict(os.environ, CDPR_LIB=os.path.join(ROOT, "cdpr-simulation_amd", lib))
    subprocess.run([sys.executable, "-c", CHILD], env=env)
