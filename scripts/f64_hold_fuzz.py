"""Fuzz: the role-split fp64 kernels' hold controller (passes of cables, straight-line path for steady cables, per-cable code for the rest)
against the one-wave kernel (per-cable code only) on random command sequences - random refresh intervals, commands scattered around
epsilon, occasional Position / Force mode, saturating commands, world resets; bit for bit, every refresh.  argv: seeds..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from dataclasses import replace
import cdpr_simulation_amd as pkg

def run(seed, build, cables, B, steps_total):
    os.environ["CDPR_F64_SPLIT"] = build
    rng = np.random.default_rng(seed)
    full = pkg.eight_cable_model()
    model = replace(full, frame_anchors=full.frame_anchors[:cables], platform_anchors=full.platform_anchors[:cables])
    eps = float(rng.choice([0.002, 0.004, 0.01]))
    cfg = pkg.Config(model=model, batch=B, stages=3, precision=64, velocityEpsilon=eps)
    cfg.velocityController.dBufferLength = int(rng.integers(3, 12)); cfg.velocityController.dDegree = int(rng.integers(1, min(4, cfg.velocityController.dBufferLength - 1) + 1))
    cfg.positionController.dBufferLength = int(rng.integers(3, 12)); cfg.positionController.dDegree = int(rng.integers(1, min(4, cfg.positionController.dBufferLength - 1) + 1))
    eng = pkg.Engine(cfg, 0)
    pose = np.tile(model.home_pose(), (B, 1)); pose[:, :3] += rng.uniform(-0.02, 0.02, (B, 3))
    eng.set_platform_state_f64(pose7=pose)
    out, done = [], 0
    while done < steps_total:
        kind = rng.choice(["vel", "vel", "vel", "vel", "pos", "frc", "reset", "big"])
        k = int(rng.integers(1, 40))
        if kind == "reset":
            eng.reset(); eng.set_platform_state_f64(pose7=pose)
        elif kind == "pos":
            eng.set_position_command(rng.uniform(-0.004, 0.004, (B, cables)).astype(np.float32))
        elif kind == "frc":
            eng.set_force_command(rng.uniform(5.0, 20.0, (B, cables)).astype(np.float32))
        else:
            amp = 1.0 if kind == "vel" else 30.0
            c = (amp * rng.choice([0.0, 0.5 * eps, 0.99 * eps, 1.01 * eps, 3 * eps, -2 * eps, 8 * eps], (B, cables)) * rng.choice([1.0, 1.0, -1.0], (B, cables))).astype(np.float32)
            eng.set_velocity_command(c)
        for _ in range(k):
            eng.update(1)
        done += k
        out.append(eng.observables_f64() + eng.raw_state_f64())
    name = eng.kernel_name
    eng.close()
    return out, name

bad = 0
for seed in [int(x) for x in (sys.argv[1:] or ["1", "2", "3"])]:
    cables, B = [(8, 137), (6, 75), (7, 200)][seed % 3]
    ref, n0 = run(seed, "0", cables, B, 1500)
    for build in ("1", "2"):
        got, n1 = run(seed, build, cables, B, 1500)
        worst = max(float(np.abs(x.astype(np.float64) - y.astype(np.float64)).max()) for a, b in zip(ref, got) for x, y in zip(a, b))
        nan = any(not np.isfinite(x).all() for a in got for x in a)
        print(f"seed {seed} n={cables} B={B}: {n1} vs {n0}: {len(ref)} refreshes, worst difference {worst:.3e}{' NaN!' if nan else ''}", flush=True)
        bad += worst != 0.0 or nan
print("FAILED" if bad else "all identical")
sys.exit(1 if bad else 0)
