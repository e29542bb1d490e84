"""The fp64 kernels' variants beside each other on non-hold handles (us per world step, HIP events, one launch per step): the role-split
kernel AUTO picks, the one-wave kernel with rings / rows in LDS, and the plain one-wave variant (rings in memory) the HOLD
instantiations are built from."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
os.environ["CDPR_NO_GRAPH"] = "1"
import cdpr_simulation_amd as pkg, bench
for B in (1, 4096, 65536):
    model, pose, command, _ = bench.make_workload(pkg, B, 8, 1235, 10)
    for label, env in (("auto", {}), ("one-wave, LDS rings + rows", {"CDPR_F64_SPLIT": "0", "CDPR_F64_RING_LDS": "1", "CDPR_F64_JCACHE": "1"}),
                       ("one-wave, LDS rings", {"CDPR_F64_SPLIT": "0", "CDPR_F64_RING_LDS": "1", "CDPR_F64_JCACHE": "0"}),
                       ("one-wave, plain", {"CDPR_F64_SPLIT": "0", "CDPR_F64_RING_LDS": "0"})):
        for k in ("CDPR_F64_SPLIT", "CDPR_F64_RING_LDS", "CDPR_F64_JCACHE"):
            os.environ.pop(k, None)
        os.environ.update(env)
        eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=3, precision=64), 0)
        eng.set_platform_state(pose7=pose); eng.set_velocity_command(command(0)); eng.update(50); eng.synchronize()
        ts = []
        for rnd in range(5):
            eng.profile_begin(); eng.update(200, 1); ms, nl = eng.profile_end(); ts.append(ms / 200 * 1e3)
        print(f"B={B} {label}: {np.median(ts):.2f} us/step", flush=True)
        eng.close()
