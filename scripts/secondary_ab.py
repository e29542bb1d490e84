"""A/B of library builds (CDPR_LIB) on the secondary workloads: general path steady (65 536 x 8, hold branch live), fp64 (65 536 x 8,
plain and with the hold branch), 524 288 x 8 - HIP-event medians, interleaved subprocesses on one box."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
import cdpr_simulation_amd as pkg, bench
B = 65536
model, pose, command, n_cmd = bench.make_workload(pkg, B, 8, 1235, 10)
def run(label, warm, steps, **kw):
    eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=3, **kw), 0)
    eng.set_platform_state(pose7=pose); eng.set_velocity_command(command(0)); eng.update(warm); eng.synchronize()
    ts = []
    for rnd in range(5):
        eng.profile_begin(); eng.update(steps); ms, nl = eng.profile_end(); ts.append(ms / steps * 1e3)
    print(os.environ.get("CDPR_LIB"), f"{label}: {np.median(ts):.2f} us/step (min {min(ts):.2f})", flush=True)
    eng.close()
run("general steady", 120, 300, velocityEpsilon=0.001)
run("fp64 plain    ", 100, 200, precision=64)
run("fp64 hold     ", 100, 200, precision=64, velocityEpsilon=0.001)
''' % ROOT
for rep in range(2):
    for lib in sys.argv[1:]:
        subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CDPR_LIB=lib))
