"""The three wavefront mappings side by side (VERDICT r02 task 4): lane-per-robot (1), lane-pair (2), lane-per-cable (3),
one-step launches, config 2 (n = 4, IK + PID + dynamics) and config 3 (n = 8, all stages) at B in {1, 512, 4096, 16384,
65536}.  Run it under rocprofv3 (--kernel-trace for the durations, --pmc SQ_INSTS_VALU SQ_WAVES for the instruction
counts) and feed the CSVs to scripts/mapping_scan_summary.py; without a profiler it prints HIP-event timings.
Every (mapping, config, batch) runs the same number of launches, so dispatches are attributed by kernel name + grid size."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
os.environ["CDPR_NO_GRAPH"] = "1"  # eager launches: every dispatch is a traced kernel of its own
import cdpr_simulation_amd as pkg
import bench

LAUNCHES = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for n, stages in ((8, 3), (4, 0)):
    for B in (1, 512, 4096, 16384, 65536):
        model, pose, command, _ = bench.make_workload(pkg, B, n, 1235, 10)
        for mapping in (1, 2, 3):
            os.environ["CDPR_MAPPING"] = str(mapping)
            eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=stages), 0)
            eng.set_platform_state(pose7=pose); eng.set_velocity_command(command(0)); eng.update(50); eng.synchronize()
            ts = []
            for rnd in range(5):
                eng.profile_begin(); eng.update(LAUNCHES // 5); ms, nl = eng.profile_end(); ts.append(ms * 1e3 / max(nl, 1))
            print(f"n={n} stages={stages} B={B} mapping={eng.mapping}: {np.median(ts):.2f} us/step by HIP events (min {min(ts):.2f})", flush=True)
            eng.close()
