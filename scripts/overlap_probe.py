"""Does splitting the batch over independent streams overlap the load/compute/store phases? (no code change: k engines)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cdpr_simulation_amd as pkg
import bench
B, n, stages = 65536, 8, 3
for parts in (1, 2, 4, 8):
    engs = []
    for p in range(parts):
        model, pose, command, n_cmd = bench.make_workload(pkg, B // parts, n, 1235 + p, 10)
        e = pkg.Engine(pkg.Config(model=model, batch=B // parts, stages=stages), 0)
        e.set_platform_state(pose7=pose); e.set_velocity_command(command(0)); e.update(20); e.synchronize()
        engs.append(e)
    for spl in (1,):
        best = 1e9
        for rnd in range(5):
            for e in engs: e.synchronize()
            t0 = time.perf_counter()
            for k in range(300):
                for e in engs: e.update(1, spl)
            for e in engs: e.synchronize()
            best = min(best, (time.perf_counter() - t0) / 300 * 1e6)
        print(f"parts={parts}: {best:.2f} us per whole-batch step -> {B/best*1e6:.3e} st/s", flush=True)
    for e in engs: e.close()
