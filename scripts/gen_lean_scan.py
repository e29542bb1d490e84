"""General controller path, one step per launch, n = 8 with FK + TD, hold branch live, batches beyond the role-split
kernel's reach: the lean role-split kernel (two waves per SIMD, the rare controller paths by call; CDPR_GEN_LEAN=1) against the
one-wave kernel alone (CDPR_GEN_LEAN=0), HIP-event medians over the batch size, interleaved subprocesses on one box, with a
digest of the state (same bits).  "steady": one held Joy, every window a uniform grid; "switching": epsilon = 0.004 and the
bench's sines refreshed every 10 steps, cables keep switching Pids.  -> profiles/r05_gen_lean_scan.txt"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
import cdpr_simulation_amd as pkg, bench
for B in [int(x) for x in os.environ.get("SCAN_B", "16384,32768,49152,65536,131072").split(",")]:
    for eps, label in ((0.001, "steady"), (0.004, "switching")):
        model, pose, command, n_cmd = bench.make_workload(pkg, B, 8, 1235, 10)
        eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=3, velocityEpsilon=eps), 0)
        eng.set_platform_state(pose7=pose); eng.set_velocity_command(command(0)); eng.update(50); eng.synchronize()
        sched = [eng.device_upload(command(j + 1)) for j in range(50)]
        ts, per = [], []
        for rnd in range(5):
            eng.profile_begin()
            for j in range(10):
                if label == "switching": eng.bind_velocity_command_device(sched[rnd * 10 + j], B * 8)
                eng.update(10)
            ms, nl = eng.profile_end(); ts.append(ms / 100 * 1e3)
        p, t = eng.platform_state()
        digest = float(np.abs(p).sum() + np.abs(t).sum() + np.abs(eng.joint_states()[2]).sum())
        print(os.environ.get("LABEL"), f"B={B} {label}: {np.median(ts):.2f} us/step (min {min(ts):.2f}, max {max(ts):.2f})  digest {digest!r}", flush=True)
        eng.close()
''' % ROOT
for rep in range(2):
    for label, env in (("lean role-split", {"CDPR_GEN_LEAN": "1"}), ("one wave     ", {"CDPR_GEN_LEAN": "0", "CDPR_GEN_SPLIT": "0"}), ("AUTO         ", {})):
        subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LABEL=label, **env))
