"""A/B timing of kernel variants in ONE process (interleaved rounds, median + min), per cdna guide rule 24."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cdpr_simulation_amd as pkg
import bench

def make(B, n, stages, kernel):
    os.environ["CDPR_KERNEL"] = kernel
    model, pose, command, n_cmd = bench.make_workload(pkg, B, n, 1235, 10)
    eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=stages), 0)
    eng.set_platform_state(pose7=pose)
    eng.set_velocity_command(command(0))
    eng.update(50); eng.synchronize()
    return eng

def t(eng, steps, spl):
    eng.profile_begin(); eng.update(steps, spl); ms, nl = eng.profile_end()
    return ms / steps * 1e3

variants = sys.argv[1].split(",") if len(sys.argv) > 1 else ["cur"]
cases = [(65536, 8, 3), (65536, 8, 0), (4096, 4, 0), (524288, 8, 3)]
for (B, n, stages) in cases:
    engs = {v: make(B, n, stages, v) for v in variants}
    for spl in (1, 8):
        res = {v: [] for v in variants}
        for rnd in range(7):
            for v in variants:
                res[v].append(t(engs[v], 200, spl))
        line = f"B={B} n={n} stages={stages} spl={spl}: "
        for v in variants:
            med, mn = np.median(res[v]), np.min(res[v])
            line += f" k{v}: {med:.2f} us/step (min {mn:.2f}) {B/med*1e6:.3e} st/s alg {B*engs[v].bytes_per_state_step()/med*1e6/1e12:.2f} TB/s |"
        print(line, flush=True)
    for e in engs.values(): e.close()
