import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cdpr_simulation_amd as pkg, bench
for (B, n, stages) in ((4096, 4, 0), (65536, 8, 3)):
    for ng in ("1", "0"):
        os.environ["CDPR_NO_GRAPH"] = ng
        model, pose, command, n_cmd = bench.make_workload(pkg, B, n, 1235, 10)
        eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=stages), 0)
        eng.set_platform_state(pose7=pose); eng.set_velocity_command(command(0)); eng.update(64); eng.synchronize()
        for spl in (1, 8):
            best = 1e9
            for rnd in range(5):
                eng.synchronize(); t0 = time.perf_counter(); eng.update(1600, spl); eng.synchronize()
                best = min(best, (time.perf_counter() - t0) / 1600 * 1e6)
            print(f"B={B} n={n} no_graph={ng} spl={spl}: wall {best:.2f} us/step  {B/best*1e6:.3e} st/s", flush=True)
        eng.close()
