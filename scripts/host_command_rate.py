"""PCIe-inclusive rate: every command batch comes from a HOST buffer (cdpr_set_velocity_command, 65 536 x 8 floats = 2 MiB
per batch), ten steps per batch, no synchronisation inside the loop — against the same loop with the batches resident in
HBM (cdpr_bind_velocity_command_device).  Usage: host_command_rate.py [lib.so ...]"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np
import cdpr_simulation_amd as pkg, bench
B, n, periods = 65536, 8, 300
model, pose, command, n_cmd = bench.make_workload(pkg, B, n, 1235, 10)
eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=3), 0)
eng.set_platform_state(pose7=pose); eng.set_velocity_command(command(0)); eng.update(50); eng.synchronize()
batches = [np.ascontiguousarray(command(j)) for j in range(8)]
res = {}
for label in ("host", "resident"):
    if label == "resident":
        dptrs = [eng.device_upload(b) for b in batches]
    ts = []
    for rnd in range(5):
        eng.synchronize(); t0 = time.perf_counter()
        for p in range(periods):
            if label == "host": eng.set_velocity_command(batches[p %% 8])
            else: eng.bind_velocity_command_device(dptrs[p %% 8], B * n)
            eng.update(10)
        eng.synchronize(); ts.append((time.perf_counter() - t0) / (periods * 10) * 1e6)
    res[label] = float(np.median(ts))
print(os.environ.get("CDPR_LIB", "default"), f"host batches {res['host']:.2f} us/step = {B / res['host'] * 1e6:.3e} state-steps/s   resident batches {res['resident']:.2f} us/step = {B / res['resident'] * 1e6:.3e}   ratio {res['resident'] / res['host']:.3f}", flush=True)
''' % ROOT
for rep in range(2):
    for lib in (sys.argv[1:] or ["libcdpr_hip.so"]):
        subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CDPR_LIB=lib))
