import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cdpr_simulation_amd as pkg, bench
B, n = 4096, 4
model, pose, command, n_cmd = bench.make_workload(pkg, B, n, 1234, 5000)
def run(label, mode):
    eng = pkg.Engine(pkg.Config(model=model, batch=B), 0)
    eng.set_platform_state(pose7=pose)
    sched = [eng.device_upload(command(j)) for j in range(50)]
    eng.bind_velocity_command_device(sched[0], B * n); eng.update(500); eng.synchronize()
    eng.profile_begin(); t0 = time.perf_counter()
    if mode == "bench":
        for j in range(500):
            eng.bind_velocity_command_device(sched[j % 50], B * n); eng.update(10)
    elif mode == "one":
        eng.update(5000)
    t1 = time.perf_counter()
    ms, nl = eng.profile_end(); t2 = time.perf_counter()
    print(f"{label}: mapping {eng.mapping}: events {ms*1e3/nl:.2f} us/launch, host enqueue {(t1-t0)*1e6/nl:.2f} us/launch, wall {(t2-t0)*1e6/nl:.2f}", flush=True)
    eng.close()
run("bench loop (bind + update(10))", "bench")
run("update(5000) one call (graph replays)", "one")
os.environ["CDPR_NO_GRAPH"] = "1"
run("update(5000) one call, eager", "one")
run("bench loop, eager", "bench")
