"""Batches between one and two robots per hardware lane (65 536 < B <= 131 072), one-step launches, n = 8 with FK + TD:
the role-split kernel in two rounds (CDPR_LOWREG=0) against the low-register kernel with two identical waves per SIMD
(CDPR_LOWREG=1).  One subprocess per variant, interleaved, same box."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
import cdpr_simulation_amd as pkg, bench
os.environ["CDPR_MAPPING"] = "1"
for B in (65536, 73728, 81920, 90112, 98304, 114688, 131072, 196608):
    model, pose, command, n_cmd = bench.make_workload(pkg, B, 8, 1235, 10)
    eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=3), 0)
    eng.set_platform_state(pose7=pose); eng.set_velocity_command(command(0)); eng.update(50); eng.synchronize()
    ts = []
    for rnd in range(7):
        eng.profile_begin(); eng.update(150); ms, nl = eng.profile_end(); ts.append(ms / 150 * 1e3)
    print(os.environ.get("LABEL"), f"B={B}: {np.median(ts):.2f} us/step (min {min(ts):.2f}) = {B / np.median(ts) * 1e6:.3e} state-steps/s", flush=True)
    eng.close()
''' % ROOT
for rep in range(2):
    for label, env in (("split (two rounds)", {"CDPR_LOWREG": "0"}), ("low-register (2 waves/SIMD)", {"CDPR_LOWREG": "1"})):
        subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LABEL=label, **env))
