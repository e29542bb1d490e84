"""Bit-level digests of what the library computes, one line of JSON: {scenario: sha256}.

Run once per build of libcdpr_hip.so (CDPR_LIB=<file> selects the build, scripts/build_variants.sh makes them); two builds
of the same sources must print the same digests (tests/test_gpu_build_variants.py).  Every kernel family the engine can
select is driven for ~200 world steps through the C-ABI with seeded inputs: the role-split, one-wave, low-register,
multi-step, lane-pair and lane-per-cable kernels of the register-resident path, per-robot handles, the general controller
path (hold branch with cables switching Pids, cascades, long windows; one-wave and role-split), the optional physics,
the fp64 kernels, the trajectory record, the schedule-in-one-launch form and the MPC rollout.  Batches are not multiples
of 64 (a partly filled last wavefront) and span several wavefronts.

  python scripts/variant_digest.py [--only name,name] [--dump DIR]

--dump DIR additionally writes the raw arrays of every scenario to DIR/<scenario>.npz (to locate a difference)."""
import argparse
import hashlib
import json
import os
import sys
from dataclasses import replace

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def poses(model, B, rng, dp=0.02, dr=0.05):
    from scipy.spatial.transform import Rotation

    pose = np.tile(model.home_pose(), (B, 1))
    pose[:, :3] += rng.uniform(-dp, dp, (B, 3))
    pose[:, 3:] = Rotation.from_rotvec(rng.uniform(-dr, dr, (B, 3))).as_quat()
    return pose.astype(np.float32)


def hold_commands(rng, B, n, eps, share=0.35):
    cmd = rng.uniform(-0.04, 0.04, (B, n)).astype(np.float32)
    cmd[np.abs(cmd) <= 2 * eps] = np.float32(3 * eps)
    low = rng.random((B, n)) < share
    cmd[low] = (rng.uniform(-1.0, 1.0, int(low.sum())) * eps).astype(np.float32)
    return cmd


class Collector:
    def __init__(self):
        self.arrays = []

    def add(self, *arrs):
        for a in arrs:
            self.arrays.append(np.ascontiguousarray(a))

    def snap(self, eng, f64=False):
        if f64:
            self.add(*eng.observables_f64(), *eng.raw_state_f64())
        else:
            self.add(*eng.platform_state(), *eng.joint_states(), *eng.raw_state())
            if eng.config.stages & 1:
                self.add(*eng.fk_state())
            if eng.config.stages & 2:
                self.add(*eng.td_state())

    def digest(self):
        h = hashlib.sha256()
        for a in self.arrays:
            h.update(a.tobytes())
        return h.hexdigest()


def env(**kv):
    """Environment overrides read by cdpr_create; returns the undo function."""
    old = {k: os.environ.get(k) for k in kv}
    for k, v in kv.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)

    def undo():
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

    return undo


def velocity_script(pkg, c, cfg, seed, f64=False, spl=1, rounds=(13, 40, 7, 61, 29, 50), position_at=4, force_at=None, record=False):
    rng = np.random.default_rng(seed)
    B, n = cfg.batch, cfg.n_cables
    eng = pkg.Engine(cfg, 0)
    eng.set_platform_state(pose7=poses(cfg.model, B, rng))
    eng.update(9, spl)
    for j, k in enumerate(rounds):
        if j == position_at:
            eng.set_position_command(rng.uniform(-0.004, 0.004, (B, n)).astype(np.float32))
        elif force_at is not None and j == force_at:
            eng.set_force_command(rng.uniform(6.0, 12.0, (B, n)).astype(np.float32))
        else:
            eng.set_velocity_command(rng.uniform(-0.03, 0.03, (B, n)).astype(np.float32))
        if record:
            r = eng.update_record(k, spl)
            c.add(*[r[key] for key in ("position", "velocity", "effort", "pose", "twist")])
        else:
            eng.update(k, spl)
        c.snap(eng, f64)
    eng.close()


def sc_config3(pkg, c, spl=1, record=False, B=4133, **envs):
    undo = env(CDPR_MAPPING=1, **envs)
    try:
        velocity_script(pkg, c, pkg.Config(model=pkg.eight_cable_model(), batch=B, stages=3), 11, spl=spl, record=record, force_at=2)
    finally:
        undo()


def sc_stage_mix(pkg, c):
    """the other stage combinations and cable counts of the one-step / multi-step kernels"""
    undo = env(CDPR_MAPPING=1)
    try:
        full = pkg.eight_cable_model()
        for cables, stages, B in ((8, 1, 517), (8, 2, 517), (8, 0, 33000), (7, 3, 300), (6, 3, 300), (4, 0, 4133), (5, 0, 200)):
            model = pkg.cube_model() if cables == 4 else replace(full, frame_anchors=full.frame_anchors[:cables], platform_anchors=full.platform_anchors[:cables])
            for spl in (1, 6):
                velocity_script(pkg, c, pkg.Config(model=model, batch=B, stages=stages), 20 + cables, spl=spl, rounds=(13, 31, 17, 40))
    finally:
        undo()


def sc_mapping(pkg, c, mapping):
    undo = env(CDPR_MAPPING=mapping)
    try:
        for cables, stages, B in ((8, 3, 1000), (8, 0, 1000), (4, 0, 4133)):
            model = pkg.eight_cable_model() if cables == 8 else pkg.cube_model()
            for spl in (1, 5):
                velocity_script(pkg, c, pkg.Config(model=model, batch=B, stages=stages), 30 + cables, spl=spl, rounds=(13, 31, 17, 40))
    finally:
        undo()


def sc_hold(pkg, c, split, per_robot=False, B=1500, lean=None):
    undo = env(CDPR_GEN_SPLIT=split, CDPR_GEN_LEAN=lean)
    try:
        eps, n = 0.004, 8
        rng = np.random.default_rng(41)
        model = pkg.eight_cable_model()
        cfg = pkg.Config(model=model, batch=B, stages=3, velocityEpsilon=eps, perRobotCommands=per_robot)
        eng = pkg.Engine(cfg, 0)
        eng.set_platform_state(pose7=poses(model, B, rng))
        eng.update(14)
        for j, k in enumerate((3, 11, 17, 6, 25, 9, 12, 31, 10, 40, 22)):
            mask = (rng.random(B) < 0.6).astype(np.uint8) if per_robot else None
            if j == 6:
                eng.set_position_command(rng.uniform(-0.004, 0.004, (B, n)).astype(np.float32), mask)
            elif per_robot and j == 8:
                eng.set_force_command(rng.uniform(6.0, 12.0, (B, n)).astype(np.float32), mask)
            else:
                eng.set_velocity_command(hold_commands(rng, B, n, eps), mask)
            eng.update(k, 1 if j % 3 else 4)
            c.snap(eng)
        eng.close()
    finally:
        undo()


def sc_general_kinds(pkg, c):
    model = pkg.eight_cable_model()
    for kind in ("cascade", "long_window", "no_clamp", "n4_hold"):
        B = 333
        cfg = pkg.Config(model=model, batch=B, stages=3)
        if kind == "cascade":
            for f in (cfg.velocityController.pFilter, cfg.velocityController.dFilter):
                f.cascade, f.relCutoff, f.quality = 1, 0.05, 0.5
            cfg.velocityController.pGain, cfg.velocityController.iGain, cfg.velocityController.dGain = 4.0, 40.0, 0.01
        if kind == "long_window":
            for p in (cfg.velocityController, cfg.positionController):
                p.dBufferLength, p.dDegree = 21, 3
        if kind == "no_clamp":
            cfg.velocityController.cmdLimit = 0.0
        if kind == "n4_hold":
            cfg = pkg.Config(batch=B, velocityEpsilon=0.004)
        velocity_script(pkg, c, cfg, 50, rounds=(13, 31, 17, 40, 21), position_at=3)
        velocity_script(pkg, c, cfg, 51, spl=5, rounds=(13, 31, 17), position_at=1)


def sc_per_robot(pkg, c, B=2100):
    rng = np.random.default_rng(61)
    model = pkg.eight_cable_model()
    cfg = pkg.Config(model=model, batch=B, stages=3, perRobotCommands=True)
    eng = pkg.Engine(cfg, 0)
    eng.set_platform_state(pose7=poses(model, B, rng))
    eng.update(7)
    for j, k in enumerate((5, 13, 30, 8, 41, 17, 33, 26)):
        mask = (rng.random(B) < 0.5).astype(np.uint8)
        kind = j % 3
        if kind == 0:
            eng.set_velocity_command(rng.uniform(-0.03, 0.03, (B, 8)).astype(np.float32), mask)
        elif kind == 1:
            eng.set_position_command(rng.uniform(-0.004, 0.004, (B, 8)).astype(np.float32), mask)
        else:
            eng.set_force_command(rng.uniform(6.0, 12.0, (B, 8)).astype(np.float32), mask)
        eng.update(k, 1 if j % 2 else 6)
        c.snap(eng)
    eng.close()


def sc_phys(pkg, c):
    full = pkg.eight_cable_model()
    lumped = replace(full, passive_damping=0.01, leg_inertia=0.004, cable_axial_mass=0.001, anchor_point_mass=0.002, anchor_inertia=0.001,
                     velocity_limit=10.0, unilateral_cables=True, travel_lower=-0.004, travel_upper=0.004, travel_stop=4)
    for stages in (3, 0):
        for spl in (1, 5):
            velocity_script(pkg, c, pkg.Config(model=lumped, batch=300, stages=stages), 70 + stages, spl=spl, rounds=(13, 31, 17, 40))


def sc_fp64(pkg, c, split, B=130):
    undo = env(CDPR_F64_SPLIT=split)
    try:
        cfg = pkg.Config(model=pkg.eight_cable_model(), batch=B, stages=3, precision=64)
        velocity_script(pkg, c, cfg, 81, f64=True, rounds=(13, 31, 17, 40, 21), force_at=2)
        velocity_script(pkg, c, replace(cfg, stages=0, model=pkg.cube_model()), 82, f64=True, spl=4, rounds=(13, 31, 17, 40))
    finally:
        undo()


def sc_scheduled(pkg, c, B=4096):
    rng = np.random.default_rng(91)
    for cables, stages in ((4, 0), (8, 3)):
        model = pkg.eight_cable_model() if cables == 8 else pkg.cube_model()
        Bc = B if cables == 4 else 700
        eng = pkg.Engine(pkg.Config(model=model, batch=Bc, stages=stages), 0)
        eng.set_platform_state(pose7=poses(model, Bc, rng))
        eng.update(5)
        T, refresh = 137, 10
        nb = (T + refresh - 1) // refresh
        sched = rng.uniform(-0.03, 0.03, (nb, Bc, cables)).astype(np.float32)
        d_sched = eng.device_upload(sched)
        image = eng.observable_image_bytes()
        d_rec = eng.device_alloc(image * T)
        eng.update_scheduled(T, refresh, d_sched, d_rec, image * T)
        c.add(eng.device_download(d_rec, (image * T,), np.uint8))
        c.snap(eng)
        eng.device_free(d_rec)
        eng.close()  # (the schedule buffer dies with the process)


def sc_rollout(pkg, c):
    rng = np.random.default_rng(95)
    model = pkg.eight_cable_model()
    for kw in (dict(), dict(velocityEpsilon=0.004), dict(perRobotCommands=True)):
        B, S, H = 70, 6, 24
        eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=3, **kw), 0)
        eng.set_platform_state(pose7=poses(model, B, rng))
        eng.update(25)
        eng.set_velocity_command(rng.uniform(-0.03, 0.03, (B, 8)).astype(np.float32))
        eng.update(15)
        cmds = (rng.uniform(-0.03, 0.03, (B, H, 1, 8)) + rng.normal(0.0, 0.01, (B, H, S, 8))).astype(np.float32)
        ref = eng.raw_state()[0][:, :3] + np.float32([0.0, 0.0, 0.005])
        c.add(eng.rollout_velocity(cmds, ref))
        c.snap(eng)
        eng.close()


SCENARIOS = {
    "config3_role_split": lambda pkg, c: sc_config3(pkg, c),
    "config3_one_wave": lambda pkg, c: sc_config3(pkg, c, CDPR_SPLIT=0),
    "config3_first_generation": lambda pkg, c: sc_config3(pkg, c, CDPR_ONESTEP=1, CDPR_SPLIT=0),
    "config3_low_register": lambda pkg, c: sc_config3(pkg, c, CDPR_LOWREG=1),
    "config3_persistent": lambda pkg, c: sc_config3(pkg, c, CDPR_PERSIST=1, CDPR_PERSIST_GRID=5),
    "config3_fused": lambda pkg, c: sc_config3(pkg, c, spl=7),
    "config3_record": lambda pkg, c: sc_config3(pkg, c, spl=5, record=True, B=517),
    "config3_one_robot": lambda pkg, c: sc_config3(pkg, c, B=1),
    "stage_mix": sc_stage_mix,
    "lane_pair": lambda pkg, c: sc_mapping(pkg, c, 2),
    "lane_per_cable": lambda pkg, c: sc_mapping(pkg, c, 3),
    "hold_one_wave": lambda pkg, c: sc_hold(pkg, c, 0),
    "hold_role_split": lambda pkg, c: sc_hold(pkg, c, 1),
    "hold_lean": lambda pkg, c: sc_hold(pkg, c, None, lean=1),                      # lean role-split kernel: the rare controller paths by call
    "hold_per_robot_lean": lambda pkg, c: sc_hold(pkg, c, None, per_robot=True, lean=1),
    "hold_per_robot_one_wave": lambda pkg, c: sc_hold(pkg, c, 0, per_robot=True),
    "hold_per_robot_role_split": lambda pkg, c: sc_hold(pkg, c, 1, per_robot=True),
    "general_kinds": sc_general_kinds,
    "per_robot": sc_per_robot,
    "optional_physics": sc_phys,
    "fp64_one_wave": lambda pkg, c: sc_fp64(pkg, c, 0),
    "fp64_role_split": lambda pkg, c: sc_fp64(pkg, c, 1),
    "fp64_role_split_lean": lambda pkg, c: sc_fp64(pkg, c, 2),
    "scheduled": sc_scheduled,
    "rollout": sc_rollout,
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--dump", default="")
    ap.add_argument("--compare", default="", help="directory of another build's --dump: report where the arrays differ")
    a = ap.parse_args()
    import cdpr_simulation_amd as pkg

    names = [s for s in a.only.split(",") if s] or list(SCENARIOS)
    out = {}
    for name in names:
        c = Collector()
        try:
            SCENARIOS[name](pkg, c)
            out[name] = c.digest()
        except Exception as exc:  # noqa: BLE001  (a variant that cannot run a scenario is a finding, not a crash)
            out[name] = f"error: {type(exc).__name__}: {exc}"
        if a.dump:
            os.makedirs(a.dump, exist_ok=True)
            np.savez_compressed(os.path.join(a.dump, name + ".npz"), *c.arrays)
        if a.compare and os.path.exists(os.path.join(a.compare, name + ".npz")):
            ref = np.load(os.path.join(a.compare, name + ".npz"))
            for i, arr in enumerate(c.arrays):
                other = ref[f"arr_{i}"] if f"arr_{i}" in ref else None
                if other is None or other.shape != arr.shape:
                    print(f"[compare] {name}: array {i} missing or reshaped", file=sys.stderr)
                    break
                same = (arr.view(np.uint8) == other.view(np.uint8)) if arr.dtype == other.dtype else np.zeros(1, bool)
                if not same.all():
                    bad = np.argwhere(arr != other) if arr.dtype.kind != "f" else np.argwhere(~((arr == other) | (np.isnan(arr) & np.isnan(other))))
                    rows = sorted(set(int(b[-2] if arr.ndim >= 2 else b[0]) for b in bad[:5000]))
                    err = float(np.nanmax(np.abs(arr.astype(np.float64) - other.astype(np.float64)))) if arr.dtype.kind == "f" else -1.0
                    print(f"[compare] {name}: first difference in array {i} shape {arr.shape}: {len(bad)} elements, max |diff| {err:.3e}, "
                          f"robots {rows[:24]}{' ...' if len(rows) > 24 else ''} (rows mod 64: {sorted(set(x % 64 for x in rows))[:20]})", file=sys.stderr)
                    break
    print(json.dumps({"lib": os.environ.get("CDPR_LIB", "libcdpr_hip.so"), "digests": out}))


if __name__ == "__main__":
    main()
