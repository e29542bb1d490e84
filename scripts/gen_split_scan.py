"""General controller path, one step per launch, n = 8 with FK + TD, hold branch live: the role-split kernel
(cdpr_gen_split_kernel, CDPR_GEN_SPLIT=1) against the one-wave kernel (CDPR_GEN_SPLIT=0) over the batch size, HIP-event
medians, interleaved subprocesses on one box, with a digest of the state after 150 steps (the two give the same bits).
-> profiles/r04_gen_split_scan.txt"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
import cdpr_simulation_amd as pkg, bench
for B in (64, 1024, 4096, 16384, 32768, 49152, 65536):
    for eps, label in ((0.001, "steady"), (0.004, "switching")):
        model, pose, command, n_cmd = bench.make_workload(pkg, B, 8, 1235, 10)
        eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=3, velocityEpsilon=eps), 0)
        eng.set_platform_state(pose7=pose); eng.set_velocity_command(command(0)); eng.update(50); eng.synchronize()
        ts = []
        for rnd in range(5):
            eng.profile_begin()
            for j in range(10):
                if label == "switching": eng.set_velocity_command(command(rnd * 10 + j + 1))
                eng.update(10)
            ms, nl = eng.profile_end(); ts.append(ms / 100 * 1e3)
        p, t = eng.platform_state()
        digest = float(np.abs(p).sum() + np.abs(t).sum() + np.abs(eng.joint_states()[2]).sum())
        print(os.environ.get("LABEL"), f"B={B} {label}: {np.median(ts):.2f} us/step (min {min(ts):.2f})  digest {digest!r}", flush=True)
        eng.close()
''' % ROOT
for rep in range(2):
    for label, env in (("role-split", {"CDPR_GEN_SPLIT": "1"}), ("one wave  ", {"CDPR_GEN_SPLIT": "0"})):
        subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LABEL=label, **env))
