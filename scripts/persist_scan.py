"""Batches above one robot per hardware lane, one-step launches, n = 8 with FK + TD: the persistent one-wave kernel
(CDPR_PERSIST=1: one wave per SIMD walks over blocks of 64 robots, the next block's rows in flight under the current
block's arithmetic) against what CDPR_MAP_AUTO did before (role-split kernel up to 90 112 robots, low-register kernel
beyond), interleaved subprocesses on one box; the first variant also checks that the two give the same bits after 60
steps.  -> profiles/r04_persist_scan.txt"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZES = (65536, 73728, 98304, 131072, 196608, 262144, 524288)
code = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
import cdpr_simulation_amd as pkg, bench
os.environ["CDPR_MAPPING"] = "1"
for B in %r:
    model, pose, command, n_cmd = bench.make_workload(pkg, B, 8, 1235, 10)
    eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=3), 0)
    eng.set_platform_state(pose7=pose); eng.set_velocity_command(command(0)); eng.update(50); eng.synchronize()
    ts = []
    for rnd in range(7):
        eng.profile_begin(); eng.update(100); ms, nl = eng.profile_end(); ts.append(ms / 100 * 1e3)
    p, t = eng.platform_state()
    digest = float(np.abs(p).sum() + np.abs(t).sum() + np.abs(eng.joint_states()[2]).sum())
    print(os.environ.get("LABEL"), f"B={B}: {np.median(ts):.2f} us/step (min {min(ts):.2f}) = {B / np.median(ts) * 1e6:.3e} state-steps/s  digest {digest!r}", flush=True)
    eng.close()
''' % (ROOT, SIZES)
variants = [("persistent", {"CDPR_PERSIST": "1"}), ("auto before (split <= 90 112 < lowreg)", {"CDPR_PERSIST": "0"})]
for rep in range(2):
    for label, env in variants:
        subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LABEL=label, **env))
