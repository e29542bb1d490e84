"""Nine to twelve cables (VERDICT r05 missing 6 / next 8: the upstream YAML layout is a free-length `points` list,
sdf/cube.yaml:21-29; CDPR_MAX_CABLES was 8).  Uniform-mode fp32 handles on the first-generation lane-per-robot kernels: the
whole step (IK, PID, [FK, TD,] dynamics) against the fp64 oracle at n = 12 and n = 9 (an odd count: the padded pair), one step
per launch = fused = recorded = scheduled bit for bit, the MPC rollout and the one-shot solvers against the oracle, and the
combinations that stay at eight cables refused by name.  Tolerances: tests/test_gpu_parity.py."""
from dataclasses import replace

import numpy as np
import pytest

from test_gpu_parity import TOL, compare, pair, perturbed_poses

pytestmark = pytest.mark.gpu


def model_of(pkg, n):
    m = pkg.twelve_cable_model()
    return m if n == 12 else replace(m, frame_anchors=m.frame_anchors[:n], platform_anchors=m.platform_anchors[:n])


@pytest.mark.parametrize("n,stages", [(12, 3), (12, 0), (9, 3), (10, 1), (11, 2)])
def test_step_against_the_oracle(pkg, oracle, n, stages):
    B = 200
    rng = np.random.default_rng(900 + n)
    cfg = pkg.Config(model=model_of(pkg, n), batch=B, stages=stages)
    assert pkg.plan_kernel(cfg, 1).startswith(f"cdpr_step_kernel<{n}, ") and "SINGLE" in pkg.plan_kernel(cfg, 1)
    pose = perturbed_poses(cfg.model, B, rng, dp=0.03, dr=0.06)
    eng, ora = pair(pkg, oracle, cfg, pose=pose)
    assert eng.mapping == "lane-per-robot"
    cmd = rng.uniform(-0.03, 0.03, (B, n)).astype(np.float32)
    for sim in (eng, ora):
        sim.update(25)
        sim.set_velocity_command(cmd)
        sim.update(60)
    compare(eng, ora, where=f"n = {n}, velocity mode")
    assert eng.kernel_name == pkg.plan_kernel(cfg, 1)
    tgt = rng.uniform(-0.004, 0.004, (B, n)).astype(np.float32)
    for sim in (eng, ora):
        sim.set_position_command(tgt)
        sim.update(40)
    compare(eng, ora, where=f"n = {n}, position mode")
    if stages & 1:
        gp, gr, gi = eng.fk_state()
        op, orr, oi = ora.fk_state()
        assert np.abs(gp - op).max() < 1e-5 and np.array_equal(gi, oi) and gr.max() < 1e-5
    if stages & 2:
        gt, gf = eng.td_state()
        ot, of = ora.td_state()
        assert np.abs(gt - ot).max() < TOL["eff"] and np.array_equal(gf, of)


def test_launch_forms_are_bit_identical_at_twelve_cables(pkg, oracle):
    B, n = 130, 12
    rng = np.random.default_rng(912)
    cfg = pkg.Config(model=model_of(pkg, n), batch=B, stages=3)
    pose = perturbed_poses(cfg.model, B, rng, dp=0.03, dr=0.06).astype(np.float32)
    sched = rng.uniform(-0.03, 0.03, (4, B, n)).astype(np.float32)
    a, b, c = (pkg.Engine(cfg, 0) for _ in range(3))
    for e in (a, b, c):
        e.set_platform_state(pose7=pose)
        e.update(15)
    for j in range(4):  # a: one launch per step; b: ten steps per launch, every step's observables recorded
        a.set_velocity_command(sched[j]), b.set_velocity_command(sched[j])
        a.update(10)
        rec = b.update_record(10, 10)
    d = c.device_upload(sched)
    c.update_scheduled(40, 10, d)  # c: the whole schedule queued with one call
    for x, y, z in zip(a.raw_state() + a.joint_states(), b.raw_state() + b.joint_states(), c.raw_state() + c.joint_states()):
        assert np.array_equal(x, y) and np.array_equal(x, z)
    assert np.array_equal(rec["effort"][-1], a.joint_states()[2])
    c.device_free(d)
    ora = oracle.OracleSim(cfg.to_struct(), oracle.DERIV_EXACT)
    ora.set_platform_state(pose7=pose.astype(np.float64))
    ora.update(15)
    for j in range(4):
        ora.set_velocity_command(sched[j])
        ora.update(10)
    compare(a, ora, where="n = 12, schedule")


def test_rollout_and_solvers_at_twelve_cables(pkg, oracle):
    B, n, S, H = 24, 12, 16, 20
    rng = np.random.default_rng(913)
    cfg = pkg.Config(model=model_of(pkg, n), batch=B, stages=3)
    pose = perturbed_poses(cfg.model, B, rng, dp=0.02, dr=0.05).astype(np.float32)
    eng, ora = pair(pkg, oracle, cfg, pose=pose)
    eng.update(20), ora.update(20)
    cmds = (rng.uniform(-0.03, 0.03, (B, H, 1, n)) + rng.normal(0.0, 0.01, (B, H, S, n))).astype(np.float32)
    ref = pose[:, :3].copy()
    cost = eng.rollout_velocity(cmds, ref)
    ocost = ora.rollout_velocity(cmds, ref.astype(np.float64))
    assert np.isfinite(cost).all() and np.abs(cost - ocost).max() <= 2e-4 * np.abs(ocost).max()
    # one-shot solvers against the oracle's: IK, the IK -> FK round trip, TD for a wrench
    s = cfg.to_struct()
    q, qd, jac = eng.solve_ik(pose)
    L0 = np.array(s.cable_ref_length[:n])
    seed = np.tile(cfg.model.home_pose(), (B, 1)).astype(np.float32)
    fk_pose, res, it = eng.solve_fk((L0[None, :] - q).astype(np.float32), seed)
    wrench = (np.tile([0, 0, 9.8 * cfg.model.mass, 0, 0, 0], (B, 1)) + rng.uniform(-0.5, 0.5, (B, 6)) * [1, 1, 1, 0.02, 0.02, 0.02]).astype(np.float32)
    t, flag = eng.solve_td(pose, wrench)
    for r in range(B):
        oq, _, _, ojac = oracle.ik(s, pose[r].astype(np.float64))
        assert np.abs(q[r] - oq).max() < 1e-5 and np.abs(jac[r] - ojac).max() < 1e-5
        assert np.abs(fk_pose[r, :3] - pose[r, :3]).max() < 1e-4 and res[r] < 1e-5
        ot, oflag = oracle.td_wrench(s, pose[r].astype(np.float64), wrench[r].astype(np.float64))
        assert int(flag[r]) == oflag and np.abs(t[r] - ot).max() < 5e-3


@pytest.mark.parametrize("n", [9, 12])
def test_more_than_eight_cables_in_double(pkg, oracle, n):
    """Nine to twelve cables with precision = 64 (end of round 6): the plain one-wave fp64 kernel - uniform modes, FK and TD, one step
    and several per launch, the trajectory record, a mode change, the rollout - against the fp64 oracle at the double tolerances."""
    B = 70
    rng = np.random.default_rng(930 + n)
    cfg = pkg.Config(model=model_of(pkg, n), batch=B, stages=3, precision=64)
    assert pkg.plan_kernel(cfg, 1) == f"cdpr_step_kernel_f64<{n}>" == pkg.plan_kernel(cfg, 10)
    pose = perturbed_poses(cfg.model, B, rng, dp=0.02, dr=0.05).astype(np.float64)
    eng, ora = pkg.Engine(cfg, 0), oracle.OracleSim(cfg.to_struct(), oracle.DERIV_EXACT)
    eng.set_platform_state_f64(pose7=pose), ora.set_platform_state(pose7=pose)

    def same(where):
        g = eng.observables_f64()
        dp, de = np.abs(g[3] - ora.platform_state()[0]).max(), np.abs(g[2] - ora.joint_states()[2]).max()
        dq = np.abs(g[0] - ora.joint_states()[0]).max()
        assert dp < 1e-12 and dq < 1e-12 and de < 1e-8, (where, dp, dq, de)

    v = rng.uniform(-0.03, 0.03, (B, n)).astype(np.float32)
    eng.update(7), ora.update(7)
    eng.set_velocity_command(v), ora.set_velocity_command(v)
    for _ in range(15):
        eng.update(1)
    ora.update(15)
    same("one-step launches")
    eng.update(30, 10), ora.update(30)
    same("fused launches")
    eng.set_position_command((0.1 * v).astype(np.float32)), ora.set_position_command((0.1 * v).astype(np.float32))
    rec = eng.update_record(18, 6)
    ora.update(18)
    same("position mode, record")
    assert np.array_equal(rec["effort"][-1], eng.observables_f64()[2])
    S, H = 6, 12
    cmds = (rng.uniform(-0.03, 0.03, (B, H, 1, n)) + rng.normal(0.0, 0.01, (B, H, S, n))).astype(np.float32)
    ref = pose[:, :3].astype(np.float32)
    cost, ocost = eng.rollout_velocity(cmds, ref), ora.rollout_velocity(cmds, ref.astype(np.float64))
    assert np.abs(cost - ocost).max() <= 3e-7 * np.abs(ocost).max()
    assert eng.kernel_name == f"cdpr_step_kernel_f64<{n}>"
    eng.close()


def test_what_stays_at_eight_cables_is_refused_by_name(pkg):
    m = pkg.twelve_cable_model()
    lumped = replace(m, leg_inertia=0.004)
    for kw, word in ((dict(perRobotCommands=True), "per_robot_commands"), (dict(velocityEpsilon=0.001), "general controller path"),
                     (dict(precision=64, perRobotCommands=True), "per_robot_commands"), (dict(mapping=pkg._abi.MAP_LANE_PER_CABLE), "mapping")):
        with pytest.raises(pkg.CdprError) as ei:
            pkg.Engine(pkg.Config(model=m, batch=4, **kw), 0)
        assert ei.value.code == pkg._abi.ERR_UNSUPPORTED and word in str(ei.value), kw
    with pytest.raises(pkg.CdprError) as ei:
        pkg.Engine(pkg.Config(model=lumped, batch=4), 0)
    assert "lumped" in str(ei.value)
    with pytest.raises(ValueError):
        pkg.Config(model=replace(m, frame_anchors=np.vstack([m.frame_anchors, m.frame_anchors[:1]]), platform_anchors=np.vstack([m.platform_anchors, m.platform_anchors[:1]])), batch=1).to_struct()
