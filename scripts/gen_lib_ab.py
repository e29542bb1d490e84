"""A/B of library builds on the general controller path at 65 536 x 8 (bench.py's general_path leg: steady, and with the cables switching
between their two Pids - per-robot sines refreshed every 10 steps, velocityEpsilon 0.004), one subprocess per build and repetition,
same box, interleaved.   python scripts/gen_lib_ab.py libA.so libB.so ..."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:] or ["libcdpr_hip.so"]
code = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
import cdpr_simulation_amd as pkg, bench
B, n, refresh = 65536, 8, 10
out = []
model, pose, command, _ = bench.make_workload(pkg, B, n, 1235, refresh)
eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=3, velocityEpsilon=0.001), 0)
eng.set_platform_state(pose7=pose)
d = eng.device_upload(command(0)); eng.bind_velocity_command_device(d, B * n); eng.update(120); eng.synchronize()
ts = []
for rnd in range(5):
    eng.profile_begin(); eng.update(300); ms, nl = eng.profile_end(); ts.append(ms / 300 * 1e3)
out.append(f"steady {np.median(ts):.2f}")
eng.close()
warm, periods = 120, 30
model, pose, command, _ = bench.make_workload(pkg, B, n, 1235, warm + 3 * periods * refresh, refresh)
eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=3, velocityEpsilon=0.004), 0)
eng.set_platform_state(pose7=pose)
sched = [eng.device_upload(command(j)) for j in range(warm // refresh + 3 * periods)]
for j in range(warm // refresh):
    eng.bind_velocity_command_device(sched[j], B * n); eng.update(refresh)
eng.synchronize()
ts = []
for rnd in range(3):
    eng.profile_begin()
    for j in range(warm // refresh + rnd * periods, warm // refresh + (rnd + 1) * periods):
        eng.bind_velocity_command_device(sched[j], B * n); eng.update(refresh)
    ms, nl = eng.profile_end(); ts.append(ms / nl * 1e3)
out.append(f"switching {np.median(ts):.2f} ({min(ts):.2f}..{max(ts):.2f})")
eng.close()
print(os.environ.get("CDPR_LIB"), " | ".join(out), flush=True)
''' % ROOT
for rep in range(int(os.environ.get("AB_REPS", "3"))):
    for lib in libs:
        subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CDPR_LIB=lib))
