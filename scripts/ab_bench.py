"""A/B timing of the wavefront mappings in ONE process (interleaved rounds, median + min; cdna guide rule 24)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cdpr_simulation_amd as pkg
import bench

def make(B, n, stages, mapping):
    os.environ["CDPR_MAPPING"] = mapping
    model, pose, command, n_cmd = bench.make_workload(pkg, B, n, 1235, 10)
    eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=stages), 0)
    eng.set_platform_state(pose7=pose)
    eng.set_velocity_command(command(0))
    eng.update(50); eng.synchronize()
    return eng

def t(eng, steps, spl):
    eng.profile_begin(); eng.update(steps, spl); ms, nl = eng.profile_end()
    return ms / steps * 1e3

variants = sys.argv[1].split(",") if len(sys.argv) > 1 else ["1", "2"]
cases = [(65536, 8, 3), (65536, 8, 0), (4096, 4, 0), (16384, 8, 3), (131072, 8, 3), (262144, 8, 3), (524288, 8, 3)]
if os.environ.get("AB_CASES"):
    cases = [tuple(int(x) for x in c.split("x")) for c in os.environ["AB_CASES"].split(",")]
for (B, n, stages) in cases:
    engs = {v: make(B, n, stages, v) for v in variants}
    for spl in (1, 10):
        res = {v: [] for v in variants}
        for rnd in range(7):
            for v in variants:
                res[v].append(t(engs[v], 200, spl))
        line = f"B={B} n={n} stages={stages} spl={spl}: "
        for v in variants:
            med, mn = np.median(res[v]), np.min(res[v])
            line += f" map{v}: {med:.2f} us/step (min {mn:.2f}) {B/med*1e6:.3e} st/s |"
        print(line, flush=True)
    for e in engs.values(): e.close()
