"""What does cdpr_create's warm launch buy?  Time (a) cdpr_create, (b) the FIRST cdpr_update(1) + synchronize of a fresh
handle, (c) the tenth, with and without the create-time launch (CDPR_NO_WARM_LAUNCH=1), one subprocess per variant so
that every run starts with no code object loaded; then bench.py's short invocation (--steps 20 --warmup 5) either way."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np
import cdpr_simulation_amd as pkg, bench
B, n = 65536, 8
model, pose, command, _ = bench.make_workload(pkg, B, n, 1235, 10)
t0 = time.perf_counter(); eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=3), 0); t_create = time.perf_counter() - t0
eng.set_platform_state(pose7=pose); eng.synchronize()
ts = []
for k in range(10):
    t0 = time.perf_counter(); eng.update(1); eng.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
print(os.environ.get("LABEL"), f"create {t_create * 1e3:.1f} ms | first update+sync {ts[0]:.0f} us | second {ts[1]:.0f} us | tenth {ts[9]:.0f} us", flush=True)
''' % ROOT
for rep in range(3):
    for label, env in (("warm launch at create", {}), ("no warm launch", {"CDPR_NO_WARM_LAUNCH": "1"})):
        subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LABEL=label, **env))
for label, env in (("warm launch at create", {}), ("no warm launch", {"CDPR_NO_WARM_LAUNCH": "1"})):
    for rep in range(3):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-secondary", "--no-parity-check"],
                           env=dict(os.environ, **env), capture_output=True, text=True)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print(f"bench --steps 20 --warmup 5, {label}: {d['value']:.3e} state-steps/s, {d['ms_per_step'] * 1e3:.2f} us/step, kernel {d['roofline']['kernel_us']:.2f} us", flush=True)
