"""A/B of library builds on the fused launches: 10 steps per launch into a trajectory record (every step's observables kept)
and 10 steps per launch overwriting one image.  Usage: record_ab.py libA.so libB.so ...  (interleaved subprocesses)"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
import cdpr_simulation_amd as pkg, bench
B, n = 65536, 8
model, pose, command, n_cmd = bench.make_workload(pkg, B, n, 1235, 10)
eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=3), 0)
eng.set_platform_state(pose7=pose); eng.set_velocity_command(command(0)); eng.update(50); eng.synchronize()
image = eng.observable_image_bytes(); dptr = eng.device_alloc(image * 10)
for label, fn in (("record", lambda: eng.update_record_device(10, 10, dptr, image * 10)), ("overwrite", lambda: eng.update(10, 10))):
    ts = []
    for rnd in range(7):
        eng.profile_begin()
        for _ in range(30): fn()
        ms, nl = eng.profile_end(); ts.append(ms / 300 * 1e3)
    print(os.environ.get("CDPR_LIB", "default"), f"{label}: {np.median(ts):.2f} us/step (min {min(ts):.2f})", flush=True)
''' % ROOT
for rep in range(2):
    for lib in sys.argv[1:]:
        subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CDPR_LIB=lib, CDPR_MAPPING="1"))
