"""The general controller path (position-hold branch JFC.cpp:78-82, biquad cascades Pid.cpp:27-44, long windows, cmdLimit 0)
as ONE plugin whose options compose: since round 4 it is one launch per step on a lane-per-robot kernel
(cdpr_general_step.hpp) with the forms the fast path has — several steps per launch, the trajectory record, the MPC
rollout, per-robot modes, the optional physics — each against the fp64 oracle.  Tolerances: tests/test_gpu_parity.py."""
from dataclasses import replace

import numpy as np
import pytest

from test_gpu_parity import TOL, compare, pair, perturbed_poses

pytestmark = pytest.mark.gpu


def hold_commands(rng, B, n, eps, share=0.35):
    """Velocity Joys with a good share of the cables at or below epsilon (hold branch) and the rest clearly above."""
    cmd = rng.uniform(-0.04, 0.04, (B, n)).astype(np.float32)
    cmd[np.abs(cmd) <= 2 * eps] = np.float32(3 * eps)
    low = rng.random((B, n)) < share
    cmd[low] = (rng.uniform(-1.0, 1.0, int(low.sum())) * eps).astype(np.float32)
    return cmd


def cascade_config(pkg, model, B, stages, cascade=1, eps=-0.001):
    """A gentle (stable) loop through low-pass biquads on the P and D inputs of both Pids."""
    cfg = pkg.Config(model=model, batch=B, stages=stages, velocityEpsilon=eps)
    for f in (cfg.velocityController.pFilter, cfg.velocityController.dFilter, cfg.positionController.pFilter, cfg.positionController.dFilter):
        f.cascade, f.relCutoff, f.quality = cascade, 0.05, 0.5
    cfg.velocityController.pGain, cfg.velocityController.iGain, cfg.velocityController.dGain = 4.0, 40.0, 0.01
    cfg.positionController.pGain, cfg.positionController.iGain, cfg.positionController.dGain = 60.0, 20.0, 2.0
    return cfg


@pytest.mark.parametrize("cables,stages", [(8, 3), (4, 0), (7, 1), (6, 2)])
def test_hold_branch_one_step_fused_and_recorded(pkg, oracle, cables, stages):
    """velocityEpsilon > 0 with cables drifting in and out of the hold branch (both Pids of a cable sampled at non-uniform
    times: the derivative is the uniform FIR where the window's samples are consecutive and a fit on the real stamps for
    the ten steps after every switch): one step per launch, several steps per launch and the trajectory record give the
    same bits, and every published step matches the oracle."""
    B, eps = 150, 0.004
    rng = np.random.default_rng(100 + cables)
    full = pkg.eight_cable_model()
    model = pkg.cube_model() if cables == 4 else replace(full, frame_anchors=full.frame_anchors[:cables], platform_anchors=full.platform_anchors[:cables])
    cfg = pkg.Config(model=model, batch=B, stages=stages, velocityEpsilon=eps)
    pose = perturbed_poses(model, B, rng, 0.02, 0.05)
    one, ora = pair(pkg, oracle, cfg, pose)
    fused, _ = pair(pkg, oracle, cfg, pose)
    rec, _ = pair(pkg, oracle, cfg, pose)
    for e in (one, fused, rec, ora):
        e.update(14)
    for j in range(9):
        cmd = hold_commands(rng, B, cables, eps)
        k = [3, 11, 17, 6, 25, 9, 12, 31, 10][j]
        for e in (one, fused, rec, ora):
            e.set_velocity_command(cmd)
        ora_steps = []
        for _ in range(k):  # one launch per step, the oracle beside it: every published step is compared
            one.update(1), ora.update(1)
            ora_steps.append((ora.platform_state(), ora.joint_states()))
            compare(one, ora, where=f"n={cables} hold, round {j}")
        fused.update(k, 7)
        r = rec.update_record(k, 5)
        for x, y in zip(one.platform_state() + one.joint_states(), fused.platform_state() + fused.joint_states()):
            assert np.array_equal(x, y), f"fused launches differ from one-step launches in round {j}"
        for x, y in zip(one.platform_state() + one.joint_states(), rec.platform_state() + rec.joint_states()):
            assert np.array_equal(x, y)
        for i, ((op, ot), (oq, oqd, oe)) in enumerate(ora_steps):
            assert np.abs(r["pose"][i] - op).max() <= TOL["pose"] and np.abs(r["effort"][i] - oe).max() <= TOL["eff"], (j, i)
            assert np.abs(r["velocity"][i] - oqd).max() <= TOL["qd"]
    p = rng.uniform(-0.004, 0.004, (B, cables)).astype(np.float32)
    for e in (one, fused, ora):
        e.set_position_command(p)
    one.update(33), fused.update(33, 11), ora.update(33)
    compare(one, ora, where="position mode after the hold rounds")
    compare(fused, ora, where="position mode after the hold rounds (fused)")


@pytest.mark.parametrize("kind", ["hold", "cascade", "long_window", "no_clamp"])
def test_record_and_rollout_on_general_handles(pkg, oracle, kind):
    """cdpr_update_record and cdpr_rollout_velocity on handles the fast path cannot serve: hold branch live (the sampled
    sequences cross epsilon inside the horizon, so trajectories switch Pids on their own private records), biquad
    cascades, a 21-sample degree-3 window, cmdLimit 0; rollouts entered from Position mode (velocity Pid reset,
    JFC.cpp:113-115) and from Velocity mode; the handle's own state is untouched."""
    B, S, H, n = 40, 6, 24, 8
    rng = np.random.default_rng({"hold": 1, "cascade": 2, "long_window": 3, "no_clamp": 4}[kind])
    model = pkg.eight_cable_model()
    eps = 0.004 if kind == "hold" else -0.001
    if kind == "cascade":
        cfg = cascade_config(pkg, model, B, 3)
    else:
        cfg = pkg.Config(model=model, batch=B, stages=3, velocityEpsilon=eps)
    if kind == "long_window":
        for p in (cfg.velocityController, cfg.positionController):
            p.dBufferLength, p.dDegree = 21, 3
    if kind == "no_clamp":
        cfg.velocityController.cmdLimit = 0.0
    eng, ora = pair(pkg, oracle, cfg, perturbed_poses(model, B, rng, 0.02, 0.05))
    eng.update(25), ora.update(25)
    tol = dict(TOL, eff=5e-2) if kind == "long_window" else TOL

    def rollout(label):
        nominal = rng.uniform(-0.03, 0.03, (B, H, 1, n))
        cmds = (nominal + rng.normal(0.0, 0.01, (B, H, S, n))).astype(np.float32)
        if kind == "hold":
            low = rng.random((B, H, S, n)) < 0.3
            cmds[low] = (rng.uniform(-1, 1, int(low.sum())) * eps).astype(np.float32)
        ref = eng.raw_state()[0][:, :3].astype(np.float64) + [0.0, 0.0, 0.005]
        before = eng.raw_state()
        gc, oc = eng.rollout_velocity(cmds, ref), ora.rollout_velocity(cmds, ref)
        assert np.isfinite(gc).all() and np.abs(gc - oc).max() <= 1e-9 + 2e-4 * np.abs(oc).max(), f"{kind}: rollout {label}"
        for x, y in zip(before, eng.raw_state()):
            assert np.array_equal(x, y)

    rollout("from Position mode")
    v = hold_commands(rng, B, n, eps) if kind == "hold" else rng.uniform(0.005, 0.03, (B, n)).astype(np.float32)
    eng.set_velocity_command(v), ora.set_velocity_command(v)
    r = eng.update_record(23, 4)
    ora_eff = []
    for _ in range(23):
        ora.update(1)
        ora_eff.append(ora.joint_states()[2])
    assert np.abs(r["effort"] - np.array(ora_eff)).max() <= tol["eff"]
    compare(eng, ora, tol=tol, where=f"{kind}: after the record")
    rollout("from Velocity mode")  # the handle's velocity Pids carry on inside the trajectories
    eng.update(10), ora.update(10)
    compare(eng, ora, tol=tol, where=f"{kind}: the handle after its rollouts")


@pytest.mark.parametrize("physics", ["lumped", "stop"])
def test_per_robot_modes_with_the_optional_physics_and_the_hold_branch(pkg, oracle, physics):
    """per_robot_commands + lumped legs / joint stop + velocityEpsilon > 0 on one handle: masked Joys (velocity, position,
    force) reach changing subsets of the robots, fused launches in between, a rollout at the end; against the oracle."""
    B, n, eps = 130, 8, 0.004
    rng = np.random.default_rng(50)
    base = pkg.eight_cable_model()
    if physics == "lumped":
        model = replace(base, inertia=(0.9, 1.1, 1.0, 0.05, -0.03, 0.02), passive_damping=0.01, leg_inertia=0.004, cable_axial_mass=0.001,
                        anchor_point_mass=0.002, anchor_inertia=0.001)
    else:
        model = replace(base, travel_lower=-0.012, travel_upper=0.012, travel_stop=4)
    cfg = pkg.Config(model=model, batch=B, stages=3, velocityEpsilon=eps, perRobotCommands=True)
    eng, ora = pair(pkg, oracle, cfg, perturbed_poses(model, B, rng, 0.02, 0.05))
    eng.update(10), ora.update(10)
    tol = TOL if physics == "lumped" else dict(TOL, eff=5e-2)
    for rnd in range(6):
        g = (np.arange(B) + rnd) % 4
        v, p = hold_commands(rng, B, n, eps), rng.uniform(-0.004, 0.004, (B, n)).astype(np.float32)
        f = (7.0 + rng.uniform(-0.5, 0.5, (B, n))).astype(np.float32)
        for sim in (eng, ora):
            sim.set_velocity_command(v, mask=g <= 1)
            if rnd % 2:
                sim.set_position_command(p, mask=g == 2)
            if rnd >= 2:
                sim.set_force_command(f, mask=g == 3)
        k = [9, 14, 20, 6, 12, 17][rnd]
        if rnd % 3 == 1:
            eng.update(k, 6)
        else:
            eng.update(k)
        ora.update(k)
        if physics == "lumped":  # (a joint stop is a threshold: fp32 and fp64 may see a joint arrive a step apart)
            compare(eng, ora, tol=tol, where=f"round {rnd}")
    if physics == "stop":
        gp, op = eng.platform_state()[0], ora.platform_state()[0]
        assert np.isfinite(gp).all() and np.abs(gp - op).max() < 5e-4
    else:
        cmds = rng.uniform(-0.03, 0.03, (B, 12, 3, n)).astype(np.float32)
        ref = eng.raw_state()[0][:, :3].astype(np.float64)
        gc, oc = eng.rollout_velocity(cmds, ref), ora.rollout_velocity(cmds, ref)
        assert np.abs(gc - oc).max() <= 1e-9 + 2e-4 * np.abs(oc).max()


def test_a_pid_that_sleeps_for_a_long_time_keeps_its_exact_stamps(pkg, oracle):
    """The position Pid of a held cable keeps samples that may be arbitrarily old; the fit works on the real stamps (integer
    world steps, differences taken before any rounding): a cable that holds for 12 steps, runs on its velocity Pid for
    700 and holds again fits a window with one 700-step gap in it, as the reference would."""
    B, n, eps = 20, 8, 0.004
    rng = np.random.default_rng(8)
    cfg = pkg.Config(model=pkg.eight_cable_model(), batch=B, stages=3, velocityEpsilon=eps)
    eng, ora = pair(pkg, oracle, cfg, perturbed_poses(cfg.model, B, rng, 0.02, 0.05))
    fast = rng.uniform(0.01, 0.03, (B, n)).astype(np.float32)
    slow = fast.copy()
    slow[:, ::2] = 0.001
    for cmd, k in ((slow, 12), (fast, 700), (slow, 5), (fast, 3), (slow, 30)):
        eng.set_velocity_command(cmd), ora.set_velocity_command(cmd)
        if k > 100:
            eng.update(k, 20)
        else:
            eng.update(k)
        ora.update(k)
        compare(eng, ora, where=f"after {k} steps")


@pytest.mark.parametrize("cables,variant", [(8, "hold"), (6, "hold"), (7, "per_robot_lumped"), (8, "per_robot_stop"), (8, "pid_debug")])
def test_role_split_general_kernel_is_bit_identical_to_the_one_wave_kernel(pkg, monkeypatch, cables, variant):
    """One-step launches of FK + TD handles on the general path run on the role-split kernel up to two workgroups per CU
    (cdpr_general_split.hpp: estimator wave + controller wave per 64 robots; CDPR_GEN_SPLIT=0 keeps the one-wave kernel):
    same bits for the whole state, every observable, the estimator's results and the `pid` topic, with cables switching Pids
    (fit queue, ring turns), per-robot modes incl. Force, the lumped legs, the joint stop, a ragged last block and the first
    world step.  Fused launches and the record run on the one-wave kernel on both handles and continue from either."""
    B, eps = 64 * 5 + 23, 0.004
    rng = np.random.default_rng(400 + cables)
    full = pkg.eight_cable_model()
    model = replace(full, frame_anchors=full.frame_anchors[:cables], platform_anchors=full.platform_anchors[:cables])
    if variant == "per_robot_lumped":
        model = replace(model, inertia=(0.9, 1.1, 1.0, 0.05, -0.03, 0.02), passive_damping=0.01, leg_inertia=0.004, cable_axial_mass=0.001,
                        anchor_point_mass=0.002, anchor_inertia=0.001)
    if variant == "per_robot_stop":
        model = replace(model, travel_lower=-0.012, travel_upper=0.012, travel_stop=4)
    stages = 3 | (pkg._abi.STAGE_PID_DEBUG if variant == "pid_debug" else 0)
    cfg = pkg.Config(model=model, batch=B, stages=stages, velocityEpsilon=eps, perRobotCommands=variant.startswith("per_robot"))
    pose = perturbed_poses(model, B, rng, 0.02, 0.05)
    joys = [(hold_commands(rng, B, cables, eps), rng.uniform(-0.004, 0.004, (B, cables)).astype(np.float32),
             (7.0 + rng.uniform(-0.5, 0.5, (B, cables))).astype(np.float32)) for _ in range(6)]
    out = []
    # one-wave kernel | role-split kernel | the lean role-split kernel (batches beyond 32 768 robots use it: forced here; its
    # controller wave inlines the first tier and leaves through gen_lean_cold_tail for the others; not with the optional physics)
    for split, lean in (("0", "0"), ("1", "0"), ("0", "1")):
        monkeypatch.setenv("CDPR_GEN_SPLIT", split)
        monkeypatch.setenv("CDPR_GEN_LEAN", lean)
        eng = pkg.Engine(cfg, 0)
        eng.set_platform_state(pose7=pose)
        eng.update(1)  # the first world step
        snaps = []
        for rnd, (v, p, f) in enumerate(joys):
            if cfg.perRobotCommands:
                g = (np.arange(B) + rnd) % 4
                eng.set_velocity_command(v, mask=g <= 1)
                if rnd % 2:
                    eng.set_position_command(p, mask=g == 2)
                if rnd >= 2:
                    eng.set_force_command(f, mask=g == 3)
            else:
                eng.set_velocity_command(v)
            for _ in range([9, 14, 3, 12, 17, 5][rnd]):
                eng.update(1)
            if rnd == 3:
                eng.update(8, 4)  # a fused launch in between: the one-wave multi-step kernel on both handles
            snaps.append(eng.platform_state() + eng.joint_states() + eng.fk_state() + eng.td_state() + ((eng.pid_debug(),) if variant == "pid_debug" else ()))
        out.append(snaps)
        eng.close()
    for other, name in ((out[1], "role-split"), (out[2], "lean role-split")):
        for rnd, (a, b) in enumerate(zip(out[0], other)):
            for x, y in zip(a, b):
                assert np.array_equal(x, y), f"{name} kernel, round {rnd}"


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_random_call_sequences_on_the_general_path(pkg, oracle, seed):
    """The general path's host-side state machine under random call sequences, hold branch live: velocity Joys with cables at
    or below epsilon, position and force Joys (to random subsets of the robots on the per-robot seeds), updates issued one
    launch per step (role-split kernel on FK + TD handles), fused, as a trajectory record, MPC rollouts in between (they
    must leave the handle alone), world resets and state writes - compared with the oracle after every update.  Rings
    turn, windows refill, fit queues form and drain at moments no scripted test picks."""
    rng = np.random.default_rng(900 + seed)
    n = [8, 4, 8, 7][seed]
    per_robot = seed >= 2
    eps = 0.004
    full = pkg.eight_cable_model()
    model = pkg.cube_model() if n == 4 else replace(full, frame_anchors=full.frame_anchors[:n], platform_anchors=full.platform_anchors[:n])
    B = [130, 200, 97, 64][seed]
    stages = {4: 0, 7: 3, 8: 3}[n]
    cfg = pkg.Config(model=model, batch=B, stages=stages, velocityEpsilon=eps, perRobotCommands=per_robot)
    eng, ora = pair(pkg, oracle, cfg, perturbed_poses(model, B, rng, 0.02, 0.05))
    total = 0
    for op in range(55):
        kind = rng.choice(["vel", "pos", "frc", "run", "fused", "record", "rollout", "reset", "state"], p=[0.22, 0.08, 0.05, 0.3, 0.12, 0.08, 0.06, 0.04, 0.05])
        where = f"seed {seed} op {op} ({kind}) after {total} steps"
        mask = (rng.random(B) < rng.choice([0.3, 0.7, 1.0])) if per_robot else None
        kw = {"mask": mask} if per_robot else {}
        if kind == "vel":
            c = hold_commands(rng, B, n, eps)
            eng.set_velocity_command(c, **kw), ora.set_velocity_command(c, **kw)
        elif kind == "pos":
            c = rng.uniform(-0.004, 0.004, (B, n)).astype(np.float32)
            eng.set_position_command(c, **kw), ora.set_position_command(c, **kw)
        elif kind == "frc":
            c = (7.0 + rng.uniform(-0.5, 0.5, (B, n))).astype(np.float32)
            eng.set_force_command(c, **kw), ora.set_force_command(c, **kw)
        elif kind == "reset":
            eng.reset(), ora.reset()
            total = 0
        elif kind == "state":
            p = perturbed_poses(model, B, rng, 0.02, 0.05).astype(np.float32)
            eng.set_platform_state(pose7=p), ora.set_platform_state(pose7=p.astype(np.float64))
        elif kind == "rollout":
            cmds = rng.uniform(-0.03, 0.03, (B, 6, 3, n)).astype(np.float32)
            ref = eng.raw_state()[0][:, :3].astype(np.float64) + np.array([0.0, 0.0, 0.01])  # (a cost well above fp32 rounding of the positions)
            gc, oc = eng.rollout_velocity(cmds, ref), ora.rollout_velocity(cmds, ref)
            assert np.abs(gc - oc).max() <= 1e-9 + 2e-4 * np.abs(oc).max(), where
        else:
            k = int(rng.integers(1, 30))
            if kind == "run":
                for _ in range(k):
                    eng.update(1)
            elif kind == "fused":
                eng.update(k, int(rng.choice([2, 5, 10])))
            else:
                rec = eng.update_record(k, int(rng.choice([1, 4, 10])))
                assert rec["pose"].shape == (k, B, 7)
            ora.update(k)
            total += k
            compare(eng, ora, where=where)
            if kind == "record" and total > k:
                assert np.array_equal(rec["pose"][-1], eng.platform_state()[0]), where
    eng.close()
