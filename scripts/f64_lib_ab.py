"""A/B of library builds on the precision = 64 workloads of bench.py (one subprocess per build and repetition, same box, interleaved):
65 536 x 8 and 1 x 8; plain, hold branch live (every cable on its velocity Pid), hold branch with a third of the cables held.
  python scripts/f64_lib_ab.py libA.so libB.so ..."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:] or ["libcdpr_hip.so"]
code = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
import cdpr_simulation_amd as pkg, bench
out = []
for B in (65536, 1):
    for eps, third, label in ((-1.0, False, "plain"), (0.001, False, "hold"), (0.001, True, "hold3")):
        model, pose, command, n_cmd = bench.make_workload(pkg, B, 8, 1235, 10)
        eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=3, precision=64, velocityEpsilon=eps), 0)
        cmd = command(0).copy()
        if third:
            held = (np.arange(B * 8).reshape(B, 8) %% 3) == 0
            cmd[held] = 0.0
            cmd[~held & (np.abs(cmd) <= eps)] = 0.02
        eng.set_platform_state(pose7=pose); eng.set_velocity_command(cmd); eng.update(100); eng.synchronize()
        ts = []
        for rnd in range(5):
            eng.profile_begin(); eng.update(300, 1); ms, nl = eng.profile_end(); ts.append(ms / 300 * 1e3)
        out.append(f"{label}@{B} {np.median(ts):.2f}")
        eng.close()
print(os.environ.get("CDPR_LIB"), " | ".join(out), flush=True)
''' % ROOT
for rep in range(int(os.environ.get("AB_REPS", "3"))):
    for lib in libs:
        subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CDPR_LIB=lib))
