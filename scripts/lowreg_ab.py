"""A/B of two library builds (CDPR_LIB) over the batch sizes where CDPR_MAP_AUTO picks the low-register one-step kernel
(two waves per SIMD: beyond 90 112 robots), n = 8 with FK + TD, one launch per step, with a digest of the state (same bits).
  python scripts/lowreg_ab.py libcdpr_hip_prev.so libcdpr_hip.so      -> profiles/r05_lowreg_scratch_ab.txt"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, hashlib
sys.path.insert(0, %r)
import numpy as np
import cdpr_simulation_amd as pkg, bench
os.environ["CDPR_MAPPING"] = "1"
for B in (98304, 131072, 196608, 262144, 524288):
    for pr in (False, True):
        model, pose, command, n_cmd = bench.make_workload(pkg, B, 8, 1235, 10)
        eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=3, perRobotCommands=pr), 0)
        eng.set_platform_state(pose7=pose); eng.set_velocity_command(command(0)); eng.update(50); eng.synchronize()
        ts = []
        for rnd in range(7):
            eng.profile_begin(); eng.update(100); ms, nl = eng.profile_end(); ts.append(ms / 100 * 1e3)
        h = hashlib.sha256()
        for a in eng.raw_state() + eng.joint_states(): h.update(a.tobytes())
        print(os.environ.get("CDPR_LIB"), f"B={B} per_robot={int(pr)}: {np.median(ts):.2f} us/step (min {min(ts):.2f}) = {B / np.median(ts) * 1e6:.3e} state-steps/s  digest {h.hexdigest()[:12]}", flush=True)
        eng.close()
''' % ROOT
for rep in range(2):
    for lib in sys.argv[1:]:
        subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CDPR_LIB=lib))
