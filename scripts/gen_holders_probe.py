"""Does a wave pay for lanes that call different Pids?  General path at 65 536 x 8, velocityEpsilon = 0.004, one held Joy: nobody holding,
a tenth of the robots holding in one block (their waves hold entirely), a tenth holding scattered (every tenth robot: every wave
has holders and movers side by side).  us per step by HIP events."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cdpr_simulation_amd as pkg, bench
B, n = 65536, 8
model, pose, command, _ = bench.make_workload(pkg, B, n, 1235, 10)
base = command(0).copy()
base[np.abs(base) <= 0.004] = 0.02
for label, held in (("nobody holds", np.zeros(B, bool)), ("a tenth holds, one block", np.arange(B) < B // 10), ("a tenth holds, every tenth robot", np.arange(B) % 10 == 0)):
    eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=3, velocityEpsilon=0.004), 0)
    eng.set_platform_state(pose7=pose)
    cmd = base.copy(); cmd[held] = 0.0
    d = eng.device_upload(cmd); eng.bind_velocity_command_device(d, B * n); eng.update(150); eng.synchronize()
    ts = []
    for rnd in range(5):
        eng.profile_begin(); eng.update(300); ms, nl = eng.profile_end(); ts.append(ms / 300 * 1e3)
    print(f"{label}: {np.median(ts):.2f} us per step ({eng.kernel_name})", flush=True)
    eng.close()
