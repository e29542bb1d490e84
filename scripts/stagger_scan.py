"""Low-register kernel (two co-resident waves per SIMD) over more robots than hardware lanes, with the workgroups of the
second wave slot delayed by CDPR_STAGGER x ~1 us: do out-of-phase waves overlap memory and compute?  One subprocess per
variant, interleaved, same box."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
import cdpr_simulation_amd as pkg, bench
os.environ["CDPR_MAPPING"] = "1"
for B in (73728, 98304, 131072, 163840, 196608, 262144, 524288):
    model, pose, command, n_cmd = bench.make_workload(pkg, B, 8, 1235, 10)
    eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=3), 0)
    eng.set_platform_state(pose7=pose); eng.set_velocity_command(command(0)); eng.update(50); eng.synchronize()
    ts = []
    for rnd in range(5):
        eng.profile_begin(); eng.update(100); ms, nl = eng.profile_end(); ts.append(ms / 100 * 1e3)
    print(os.environ.get("LABEL"), f"B={B}: {np.median(ts):.2f} us/step (min {min(ts):.2f}) = {B / np.median(ts) * 1e6:.3e} state-steps/s", flush=True)
    eng.close()
''' % ROOT
variants = [(f"lowreg stagger {s} period {p}", {"CDPR_CHUNK": "0", "CDPR_LOWREG": "1", "CDPR_STAGGER": str(s), "CDPR_STAGGER_PERIOD": str(p)})
            for s, p in ((0, 1024), (2, 1024), (4, 1024), (6, 1024), (4, 1), (4, 256), (4, 128))]
for rep in range(2):
    for label, env in variants:
        subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LABEL=label, **env))
