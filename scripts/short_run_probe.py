"""Where do the microseconds of a 20-step timed region go?  (the driver's `--steps 20 --warmup 5` invocation)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cdpr_simulation_amd as pkg, bench

B, n = 65536, 8
model, pose, command, n_cmd = bench.make_workload(pkg, B, n, 1235, 400)
def fresh():
    eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=3), 0)
    eng.set_platform_state(pose7=pose)
    sched = [eng.device_upload(command(j)) for j in range(6)]
    return eng, sched
def run(eng, sched, first, nsteps, events):
    if events: eng.profile_begin()
    t0 = time.perf_counter()
    done = 0
    while done < nsteps:
        s = first + done
        if s % 10 == 0: eng.bind_velocity_command_device(sched[(s // 10) % 6], B * n)
        k = min(10 - s % 10, nsteps - done)
        eng.update(k); done += k
    t1 = time.perf_counter()
    ev = eng.profile_end() if events else None
    t2 = time.perf_counter()
    eng.synchronize()
    t3 = time.perf_counter()
    return (t3 - t0) * 1e6, (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (ev[0] * 1e3 if ev else float("nan"))
for events in (True, False):
    for warm in (5, 500):
        rows = []
        for rep in range(5):
            eng, sched = fresh()
            eng.update(warm - warm % 10 if warm >= 10 else warm); eng.synchronize()
            rows.append(run(eng, sched, 10 if warm >= 10 else 5, 20, events))
            eng.close()
        r = np.median(np.array(rows), axis=0)
        print(f"events={events} warmup={warm}: wall {r[0]:.1f} us = {r[0]/20:.2f} us/step | host enqueue {r[1]:.1f} | event sync {r[2]:.1f} | stream sync {r[3]:.1f} | HIP events {r[4]:.1f}", flush=True)
