"""A/B of experimental library builds (CDPR_LIB), one subprocess per build, same box."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:] or ["libcdpr_hip.so"]
code = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
import cdpr_simulation_amd as pkg, bench
os.environ["CDPR_MAPPING"] = "1"
for (B, stages) in ((65536, 3), (65536, 0)):
    model, pose, command, n_cmd = bench.make_workload(pkg, B, 8, 1235, 10)
    eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=stages), 0)
    eng.set_platform_state(pose7=pose); eng.set_velocity_command(command(0)); eng.update(50); eng.synchronize()
    for spl in (1, 10):
        ts = []
        for rnd in range(7):
            eng.profile_begin(); eng.update(200, spl); ms, nl = eng.profile_end(); ts.append(ms / 200 * 1e3)
        print(os.environ.get("CDPR_LIB"), f"B={B} stages={stages} spl={spl}: {np.median(ts):.2f} us/step (min {min(ts):.2f})", flush=True)
    eng.close()
''' % ROOT
for rep in range(2):
    for lib in libs:
        subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CDPR_LIB=lib))
