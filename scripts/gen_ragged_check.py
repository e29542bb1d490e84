"""Check: the lean general kernel on ragged large batches with cables switching Pids (per-robot sines refreshed every 10 steps, velocityEpsilon
0.004) - the first 40 and the last 70 robots of 100 003 and of 40 001 replayed on the oracle over 300 steps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import cdpr_simulation_amd as pkg, bench, oracle
for B in (100003, 40001):
    n, refresh = 8, 10
    model, pose, command, _ = bench.make_workload(pkg, B, n, 4321, 400, refresh)
    cfg = pkg.Config(model=model, batch=B, stages=3, velocityEpsilon=0.004)
    eng = pkg.Engine(cfg, 0)
    eng.set_platform_state(pose7=pose)
    sl = np.r_[0:40, B - 70:B]
    osim = oracle.OracleSim(pkg.Config(model=model, batch=len(sl), stages=3, velocityEpsilon=0.004).to_struct(), oracle.DERIV_EXACT)
    osim.set_platform_state(pose7=pose[sl].astype(np.float64))
    for j in range(30):
        c = command(j)
        eng.set_velocity_command(c); osim.set_velocity_command(c[sl])
        eng.update(refresh); osim.update(refresh)
    gp, ge = eng.platform_state()[0][sl], eng.joint_states()[2][sl]
    dp, de = np.abs(gp - osim.platform_state()[0]).max(), np.abs(ge - osim.joint_states()[2]).max()
    print(B, eng.kernel_name, "pose", dp, "effort", de, "ok" if dp < 1e-4 and de < 5e-2 else "FAIL", flush=True)
    eng.close(); osim.close()
