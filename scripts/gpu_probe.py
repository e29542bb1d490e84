"""First-contact probe on the GPU box: parity magnitudes vs the oracle and raw step timing."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import cdpr_simulation_amd as pkg
from cdpr_simulation_amd import _abi
import oracle

def run_pair(cfg, pose, script, label):
    eng = pkg.Engine(cfg, 0); ora = oracle.OracleSim(cfg.to_struct())
    if pose is not None:
        eng.set_platform_state(pose7=pose.astype(np.float32)); ora.set_platform_state(pose7=pose.astype(np.float32).astype(np.float64))
    worst = {}
    for k, (kind, val) in enumerate(script):
        for sim in (eng, ora):
            if kind == "vel": sim.set_velocity_command(val)
            elif kind == "pos": sim.set_position_command(val)
            elif kind == "run": sim.update(val)
        if kind == "run":
            gp, gt = eng.platform_state(); op, ot = ora.platform_state()
            gq, gqd, ge = eng.joint_states(); oq, oqd, oe = ora.joint_states()
            for name, g, o in (("pose", gp, op), ("twist", gt, ot), ("q", gq, oq), ("qd", gqd, oqd), ("eff", ge, oe)):
                worst[name] = max(worst.get(name, 0.0), float(np.abs(g - o).max()))
    print(label, {k: f"{v:.3e}" for k, v in worst.items()}, "finite", bool(np.isfinite(gp).all()))
    return eng, ora

# config 1: single 4-cable robot, sine velocity, 3000 steps
cfg = pkg.Config(batch=3)
gen = pkg.stimulus.sine_velocity(4)
script = []
for k in range(300):
    script.append(("vel", next(gen))); script.append(("run", 10))
run_pair(cfg, None, script, "config1 sine 3000 steps:")
run_pair(pkg.Config(batch=2), None, [("run", 1000)], "hold from load 1000 steps:")

# 8-cable FK+TD
cfg8 = pkg.Config(model=pkg.eight_cable_model(), batch=130, stages=3)
rng = np.random.default_rng(5)
pose = np.tile(cfg8.model.home_pose(), (130, 1)); pose[:, :3] += rng.uniform(-0.05, 0.05, (130, 3))
script = [("run", 50)]
for k in range(30):
    script.append(("vel", rng.uniform(-0.05, 0.05, (130, 8)).astype(np.float32))); script.append(("run", 10))
eng, ora = run_pair(cfg8, pose, script, "config3-like 8 cable FK+TD:")
print("fk", eng.fk_state()[1].max(), ora.fk_state()[1].max(), eng.fk_state()[2][:4], "td flags", eng.td_state()[1].sum(), ora.td_state()[1].sum())

# timing
for (B, n, stages) in ((4096, 4, 0), (65536, 8, 3), (65536, 8, 0), (524288, 8, 3)):
    model = pkg.cube_model() if n == 4 else pkg.eight_cable_model()
    cfg = pkg.Config(model=model, batch=B, stages=stages)
    eng = pkg.Engine(cfg, 0)
    eng.set_velocity_command(np.full(n, 0.01, dtype=np.float32))
    eng.update(100); eng.synchronize()
    for spl in (1, 8):
        eng.profile_begin(); t0 = time.perf_counter()
        eng.update(400, spl); ms, nl = eng.profile_end(); t1 = time.perf_counter()
        print(f"B={B} n={n} stages={stages} spl={spl}: {ms/400*1e3:.2f} us/step (events), wall {((t1-t0)/400)*1e6:.2f} us/step, "
              f"{B*400/(ms*1e-3):.3e} steps/s, alg {B*400*eng.bytes_per_state_step()/(ms*1e-3)/1e12:.3f} TB/s")
    eng.close()
