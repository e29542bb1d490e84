import os, subprocess, sys
code = r'''
import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
os.environ["CDPR_NO_GRAPH"] = "1"
import cdpr_simulation_amd as pkg, bench
B, n = 65536, 8
model, pose, command, _ = bench.make_workload(pkg, B, n, 1235, 10)
eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=3, precision=64), 0)
eng.set_platform_state(pose7=pose); eng.set_velocity_command(command(0)); eng.update(100); eng.synchronize()
ts = []
for rnd in range(7):
    eng.profile_begin(); eng.update(300, 1); ms, nl = eng.profile_end(); ts.append(ms / 300 * 1e3)
print(os.environ.get("CDPR_LIB"), f"fp64 65536x8: {np.median(ts):.2f} us/step (min {min(ts):.2f})", flush=True)
'''
for rep in range(3):
    for lib in sys.argv[1:]:
        subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CDPR_LIB=lib))
