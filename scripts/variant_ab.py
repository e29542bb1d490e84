"""A/B of kernel variants selected by environment (CDPR_ONESTEP, CDPR_LIB, CDPR_LOWREG ...), one subprocess per
variant, variants interleaved and repeated, same box.  Usage: variant_ab.py "label:K=V,K=V" "label2:K=V" ...
AB_CASES=BxNxSTAGES,... picks the cases (default 65536x8x3, 65536x8x0, 4096x4x0)."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variants = []
for spec in sys.argv[1:]:
    label, _, kv = spec.partition(":")
    variants.append((label, dict(x.split("=", 1) for x in kv.split(",") if x)))
code = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
import cdpr_simulation_amd as pkg, bench
os.environ.setdefault("CDPR_MAPPING", "1")
cases = [tuple(int(x) for x in c.split("x")) for c in os.environ.get("AB_CASES", "65536x8x3,65536x8x0,4096x4x0").split(",")]
for (B, n, stages) in cases:
    model, pose, command, n_cmd = bench.make_workload(pkg, B, n, 1235, 10)
    eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=stages), 0)
    eng.set_platform_state(pose7=pose); eng.set_velocity_command(command(0)); eng.update(50); eng.synchronize()
    for spl in (1, 10):
        ts = []
        for rnd in range(7):
            eng.profile_begin(); eng.update(300, spl); ms, nl = eng.profile_end(); ts.append(ms / 300 * 1e3)
        print(os.environ.get("AB_LABEL"), f"B={B} n={n} stages={stages} spl={spl}: {np.median(ts):.2f} us/step (min {min(ts):.2f})", flush=True)
    eng.close()
if os.environ.get("AB_ROLLOUT", "1") == "1":
    Br, S, H = bench.ROLLOUT_SHAPE
    model, pose, command, _ = bench.make_workload(pkg, Br, 8, 1235, 10)
    er = pkg.Engine(pkg.Config(model=model, batch=Br, stages=3), 0)
    er.set_platform_state(pose7=pose); er.update(20)
    dptr = er.device_upload(bench.make_rollout_commands(Br, H, S, 8))
    d_ref, d_cost = er.device_upload(pose[:, :3].copy()), er.device_alloc(Br * S * 4)
    er.rollout_velocity_device(dptr, S, H, d_ref, d_cost); er.synchronize()
    ts = []
    for rnd in range(7):
        er.profile_begin()
        for _ in range(5): er.rollout_velocity_device(dptr, S, H, d_ref, d_cost)
        ms, nl = er.profile_end(); ts.append(ms / 5 * 1e3)
    print(os.environ.get("AB_LABEL"), f"rollout {Br}x{S}x{H}: {np.median(ts):.1f} us (min {min(ts):.1f}) = {np.median(ts) / H:.2f} us/step", flush=True)
    er.close()
''' % ROOT
for rep in range(2):
    for label, env in variants:
        subprocess.run([sys.executable, "-c", code], env=dict(os.environ, AB_LABEL=label, **env))
