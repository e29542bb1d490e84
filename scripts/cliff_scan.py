"""Batches above one robot per hardware lane (65 536 < B <= 524 288), one-step launches, n = 8 with FK + TD: what
CDPR_MAP_AUTO does since round 4 (blocks of <= 65 536 robots launched back to back, each at the role-split kernel's
operating point: CDPR_CHUNK unset) against one launch of the role-split kernel over the whole batch (CDPR_CHUNK=0
CDPR_LOWREG=0) and one launch of the low-register kernel (two identical waves per SIMD: CDPR_CHUNK=0 CDPR_LOWREG=1).
One subprocess per variant, interleaved, same box.  -> profiles/r04_cliff_scan.txt"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
import cdpr_simulation_amd as pkg, bench
os.environ["CDPR_MAPPING"] = "1"
for B in (65536, 73728, 81920, 98304, 114688, 131072, 163840, 196608, 262144, 393216, 524288):
    model, pose, command, n_cmd = bench.make_workload(pkg, B, 8, 1235, 10)
    eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=3), 0)
    eng.set_platform_state(pose7=pose); eng.set_velocity_command(command(0)); eng.update(50); eng.synchronize()
    ts = []
    for rnd in range(7):
        eng.profile_begin(); eng.update(100); ms, nl = eng.profile_end(); ts.append(ms / 100 * 1e3)
    print(os.environ.get("LABEL"), f"B={B}: {np.median(ts):.2f} us/step (min {min(ts):.2f}; {nl // 100} launch(es) per step) = {B / np.median(ts) * 1e6:.3e} state-steps/s", flush=True)
    eng.close()
''' % ROOT
variants = [("auto (round 4: blocks <= 65 536)", {}), ("one launch, role-split kernel", {"CDPR_CHUNK": "0", "CDPR_LOWREG": "0"}),
            ("one launch, low-register kernel", {"CDPR_CHUNK": "0", "CDPR_LOWREG": "1"}), ("blocks <= 65 536 at every size", {"CDPR_CHUNK": "65536"}),
            ("blocks <= 32 768 at every size", {"CDPR_CHUNK": "32768"})]
for rep in range(2):
    for label, env in variants:
        subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LABEL=label, **env))
