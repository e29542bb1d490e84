"""A/B of the headline workload (65 536 x 8, every stage, one launch per step) between builds of the library (CDPR_LIB): HIP-event
medians of 2 000-step runs, interleaved subprocesses on one box, with a digest of the state."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
import cdpr_simulation_amd as pkg, bench
B = 65536
model, pose, command, n_cmd = bench.make_workload(pkg, B, 8, 1235, 10)
eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=3), 0)
eng.set_platform_state(pose7=pose); eng.set_velocity_command(command(0)); eng.update(500); eng.synchronize()
ts = []
for rnd in range(7):
    eng.profile_begin(); eng.update(2000); ms, nl = eng.profile_end(); ts.append(ms / 2000 * 1e3)
p, t = eng.platform_state()
print(os.environ.get("CDPR_LIB"), f"{np.median(ts):.3f} us/step (min {min(ts):.3f})  digest {float(np.abs(p).sum() + np.abs(t).sum())!r}", flush=True)
''' % ROOT
for rep in range(3):
    for lib in sys.argv[1:]:
        subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CDPR_LIB=lib))
