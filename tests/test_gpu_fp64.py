"""cdpr_config_t.precision = 64: the step in the reference's own precision (cdpr_step_kernel_f64.hpp), through the C-ABI,
against the fp64 oracle.  Both sides compute in double; they differ in the order of the sums and in the derivative's
formulation (closed-form end-point weights here, a least-squares fit in centred time there), so the agreement is that of
two double implementations.  Measured on MI355X over this module: pose 8.3e-16, twist 5.1e-14, joint position 6.7e-16,
joint velocity 3.9e-14, effort 1.2e-11 N (the position Pid's D gain times 1/dt = 8e4 N s/m amplifies the last bits of the
1 ms window); config 1 over 3 000 steps: pose 1.7e-16, effort 7.3e-14.  Tolerances: two orders above that."""
import numpy as np
import pytest

from test_gpu_parity import perturbed_poses

pytestmark = pytest.mark.gpu

TOL64 = {"pose": 1e-13, "twist": 5e-12, "q": 1e-13, "qd": 5e-12, "eff": 1e-9}
WORST = {}


def compare64(eng, ora, where, tol=TOL64):
    gq, gqd, ge, gp, gt = eng.observables_f64()
    op, ot = ora.platform_state()
    oq, oqd, oe = ora.joint_states()
    worst = {}
    for name, g, o in (("pose", gp, op), ("twist", gt, ot), ("q", gq, oq), ("qd", gqd, oqd), ("eff", ge, oe)):
        assert g.dtype == np.float64 and np.isfinite(g).all(), where
        worst[name] = float(np.abs(g - o).max())
        assert worst[name] <= tol[name], f"{where}: {name} differs from the oracle by {worst[name]:.3e} (tolerance {tol[name]:.1e})"
        WORST[name] = max(WORST.get(name, 0.0), worst[name])
    return worst


def pair64(pkg, oracle, cfg, pose=None):
    eng, ora = pkg.Engine(cfg, 0), oracle.OracleSim(cfg.to_struct(), oracle.DERIV_EXACT)
    if pose is not None:
        eng.set_platform_state_f64(pose7=pose), ora.set_platform_state(pose7=pose)
    return eng, ora


def test_config1_in_the_reference_precision(pkg, oracle):
    """BASELINE config 1 (the shipped 4-cable robot under sinevelocitytest, 3 000 steps of 1 ms) in double: the facade's
    drop-in case.  Five orders of magnitude closer to the oracle than the fp32 kernels (pose 6e-7 there)."""
    cfg = pkg.Config(batch=1, precision=64)
    eng, ora = pair64(pkg, oracle, cfg)
    gen = pkg.stimulus.sine_velocity(4)
    worst = {}
    for k in range(300):
        cmd = next(gen)
        eng.set_velocity_command(cmd), ora.set_velocity_command(cmd)
        eng.update(10), ora.update(10)
        if k % 20 == 19:
            w = compare64(eng, ora, f"step {10 * (k + 1)}")
            worst = {n: max(worst.get(n, 0.0), v) for n, v in w.items()}
    assert eng.step_count == 3000
    print("fp64 config 1, worst over the trajectory:", worst)
    # the float getters of the same handle round the doubles
    q32, qd32, e32 = eng.joint_states()
    q64, qd64, e64, p64, t64 = eng.observables_f64()
    assert np.array_equal(q32, q64.astype(np.float32)) and np.array_equal(e32, e64.astype(np.float32))
    assert np.array_equal(eng.platform_state()[0], p64.astype(np.float32))


@pytest.mark.parametrize("stages", [0, 3])
def test_eight_cable_batch_modes_and_fused_launches(pkg, oracle, stages):
    """8 cables, with and without FK + TD, ragged batch, both modes with Pid resets in between, position commands,
    fused launches (bit-identical to one-step launches: same kernel, state in registers / LDS between the steps)."""
    B = 70
    rng = np.random.default_rng(81 + stages)
    model = pkg.eight_cable_model()
    cfg = pkg.Config(model=model, batch=B, stages=stages, precision=64)
    pose = perturbed_poses(model, B, rng, 0.03, 0.05)
    eng, ora = pair64(pkg, oracle, cfg, pose)
    twin = pkg.Engine(cfg, 0)
    twin.set_platform_state_f64(pose7=pose)
    v = rng.uniform(-0.03, 0.03, (B, 8)).astype(np.float32)
    p = rng.uniform(-0.003, 0.003, (B, 8)).astype(np.float32)
    for e in (eng, ora, twin):
        e.update(12)
        e.set_velocity_command(v)
    eng.update(60), ora.update(60), twin.update(60, 15)
    compare64(eng, ora, "velocity mode")
    for e in (eng, ora, twin):
        e.set_position_command(p)
    eng.update(45), ora.update(45), twin.update(45, 9)
    compare64(eng, ora, "position mode")
    for e in (eng, ora, twin):
        e.set_velocity_command(-v)
        e.set_position_command(p[0])  # both kinds before one update; a broadcast row
    eng.update(30), ora.update(30), twin.update(30, 30)
    compare64(eng, ora, "both kinds in one update")
    for x, y in zip(eng.observables_f64() + eng.raw_state_f64(), twin.observables_f64() + twin.raw_state_f64()):
        assert np.array_equal(x, y)
    if stages & 1:
        gp, gr, gi = eng.fk_state()
        op, orr, oi = ora.fk_state()
        assert np.array_equal(gi, oi) and np.abs(gp - op).max() < 1e-6 and gr.max() < 1e-12  # fk_state reads out as float
    if stages & 2:
        gt, gf = eng.td_state()
        ot, of = ora.td_state()
        assert np.array_equal(gf, of) and np.abs(gt - ot).max() < 1e-4
    eng.reset(), ora.reset()
    eng.update(25), ora.update(25)
    compare64(eng, ora, "after reset")


def test_fp64_limits_decimation_and_debug_topic(pkg, oracle):
    """Travel-limit flags, publishPeriod decimation, the `pid` debug topic, velocity limit and unilateral cables on the
    fp64 kernel."""
    rng = np.random.default_rng(83)
    model = pkg.eight_cable_model()
    model.travel_lower, model.travel_upper = -0.003, 0.003
    model.velocity_limit, model.unilateral_cables = 0.03, True
    B = 40
    cfg = pkg.Config(model=model, batch=B, stages=3 | pkg._abi.STAGE_PID_DEBUG, precision=64, publishPeriod=0.0035)
    eng, ora = pair64(pkg, oracle, cfg, perturbed_poses(model, B, rng, 0.01, 0.03))
    jac = oracle.ik(cfg.to_struct(), model.home_pose())[3]
    tw = np.concatenate([rng.uniform(-0.05, 0.05, (B, 3)), rng.uniform(-0.2, 0.2, (B, 3))], axis=1)
    cmd = (-(jac @ tw.T).T).astype(np.float32)
    for e in (eng, ora):
        e.update(9)
        e.set_velocity_command(cmd)
    for k in range(10):
        eng.update(17), ora.update(17)
        compare64(eng, ora, f"block {k}")
        q = ora.joint_states()[0]
        near = (np.abs(np.abs(q) - 0.003) < 1e-9).any(axis=1)
        assert np.array_equal(eng.limit_state()[~near], ora.limit_state()[~near])
        assert np.abs(eng.pid_debug() - ora.pid_debug()).max() < 1e-4  # the topic is float32 (sensor_msgs/Joy)
    assert (eng.limit_state() != 0).any()


def test_fp64_facade_publishes_doubles(pkg, oracle):
    """The drop-in surface with precision = 64: jointStates / platformPose carry float64 arrays (as the ROS messages of the
    reference do, PLG.cpp:248-280) that match the oracle to double accuracy."""
    cfg = pkg.Config(batch=1, precision=64)
    plug = pkg.CdprGazeboPlugin()
    plug.Load(cfg)
    got = {"joint": [], "platform": []}
    plug.bus.subscribe("jointStates", got["joint"].append)
    plug.bus.subscribe("platformPose", got["platform"].append)
    ora = oracle.OracleSim(cfg.to_struct())
    gen = pkg.stimulus.sine_velocity(4)
    for k in range(20):
        cmd = next(gen)
        plug.bus.publish("jointVelocities", pkg.Joy(axes=cmd))
        ora.set_velocity_command(cmd)
        plug.update(10), ora.update(10)
    assert len(got["joint"]) == 199
    js, ps = got["joint"][-1], got["platform"][-1]
    assert js.position.dtype == np.float64 and ps.pose.position.dtype == np.float64
    oq, oqd, oe = ora.joint_states()
    op, ot = ora.platform_state()
    assert np.abs(js.effort - oe).max() < TOL64["eff"] and np.abs(js.position - oq).max() < TOL64["q"]
    assert np.abs(ps.pose.position - op[:, :3]).max() < TOL64["pose"] and np.abs(ps.velocity.angular - ot[:, 3:]).max() < TOL64["twist"]


@pytest.mark.parametrize("cables,stages,B", [(8, 3, 130), (4, 0, 70), (8, 3, 40000)])
def test_fp64_per_robot_modes(pkg, oracle, cables, stages, B):
    """precision = 64 with per_robot_commands (round 5): every robot has its own JointForceCalculator mode and Pid call
    history (PLG.cpp:206-219 runs per model) - velocity, position and setForce commands reaching subsets of the robots, mode
    changes resetting the Pid of the robots they reach, one step and several steps per launch - against the oracle at fp64
    tolerances; 40 000 robots: the large-batch variant of the kernel (rings read from HBM)."""
    rng = np.random.default_rng(84 + cables)
    model = pkg.eight_cable_model() if cables == 8 else pkg.cube_model()
    cfg = pkg.Config(model=model, batch=B, stages=stages, precision=64, perRobotCommands=True)
    eng, ora = pair64(pkg, oracle, cfg, perturbed_poses(model, B, rng, 0.02, 0.05))
    eng.update(11), ora.update(11)
    compare64(eng, ora, "position mode from Load")
    grp = np.arange(B) % 4
    steps = iter([13, 1, 24, 7, 30, 12, 19])
    for rnd in range(2):
        v = rng.uniform(-0.03, 0.03, (B, cables)).astype(np.float32)
        f = ((7.0 if cables == 8 else 3.97) + rng.uniform(-0.5, 0.5, (B, cables))).astype(np.float32)
        p = rng.uniform(-0.004, 0.004, (B, cables)).astype(np.float32)
        for e in (eng, ora):
            e.set_velocity_command(v, mask=(grp <= 1).astype(np.uint8))       # groups 0, 1 -> Velocity
            e.set_force_command(f, mask=(grp == 2).astype(np.uint8))          # group 2 -> Force; group 3 stays in Position mode
        k = next(steps)
        eng.update(k, 1), ora.update(k)
        compare64(eng, ora, f"round {rnd}: velocity + force subsets")
        for e in (eng, ora):
            e.set_position_command(p, mask=(grp == 1).astype(np.uint8))       # group 1 back to Position: its Pid is reset
        k = next(steps)
        eng.update(k, 5), ora.update(k)
        compare64(eng, ora, f"round {rnd}: position subset (fused launches)")
        for e in (eng, ora):
            e.set_velocity_command(v[::-1].copy(), mask=(grp >= 2).astype(np.uint8))  # Force -> Velocity resets the velocity Pid
        k = next(steps)
        eng.update(k, 1), ora.update(k)
        compare64(eng, ora, f"round {rnd}: leaving Force mode")
    # an unmasked Joy reaches everybody
    v = rng.uniform(-0.02, 0.02, (B, cables)).astype(np.float32)
    eng.set_velocity_command(v), ora.set_velocity_command(v)
    eng.update(15, 3), ora.update(15)
    compare64(eng, ora, "unmasked Joy at the end")


@pytest.mark.parametrize("cables,stages,B", [(8, 3, 200), (4, 0, 90)])
def test_fp64_per_robot_modes_with_the_hold_branch(pkg, oracle, cables, stages, B):
    """per_robot_commands AND velocityEpsilon >= 0 in double (round 5: the PR + HOLD instantiation of the one-wave kernel, the mode
    per lane; the latch clears the rows of the Pid a robot's mode change enters): velocity Joys with cables at or below epsilon,
    position and setForce commands reaching subsets, a group going Velocity -> Position -> Velocity (both resets) while the
    others keep their Pids, fused launches and a trajectory record, a world reset - against the fp64 oracle."""
    from test_gpu_general_matrix import hold_commands

    eps = 0.004
    rng = np.random.default_rng(184 + cables)
    model = pkg.eight_cable_model() if cables == 8 else pkg.cube_model()
    cfg = pkg.Config(model=model, batch=B, stages=stages | pkg._abi.STAGE_PID_DEBUG, precision=64, perRobotCommands=True, velocityEpsilon=eps)
    eng, ora = pair64(pkg, oracle, cfg, perturbed_poses(model, B, rng, 0.02, 0.05).astype(np.float64))
    tol = dict(TOL64, eff=2e-7, twist=1e-10, qd=1e-10, pose=1e-12, q=1e-12)
    eng.update(9), ora.update(9)
    compare64(eng, ora, "position mode from Load", tol)
    grp = np.arange(B) % 4
    steps = iter([13, 1, 24, 7, 30, 12, 19, 26, 5])
    for rnd in range(2):
        v = hold_commands(rng, B, cables, eps)
        f = ((7.0 if cables == 8 else 3.97) + rng.uniform(-0.5, 0.5, (B, cables))).astype(np.float32)
        p = rng.uniform(-0.004, 0.004, (B, cables)).astype(np.float32)
        for e in (eng, ora):
            e.set_velocity_command(v, mask=(grp <= 1).astype(np.uint8))       # groups 0, 1 -> Velocity (some of their cables hold)
            e.set_force_command(f, mask=(grp == 2).astype(np.uint8))          # group 2 -> Force; group 3 stays in Position mode
        k = next(steps)
        eng.update(k, 1), ora.update(k)
        compare64(eng, ora, f"round {rnd}: velocity + force subsets", tol)
        v2 = hold_commands(rng, B, cables, eps)
        for e in (eng, ora):
            e.set_velocity_command(v2, mask=(grp == 0).astype(np.uint8))      # a new Joy in the same mode: no reset, cables change branch
            e.set_position_command(p, mask=(grp == 1).astype(np.uint8))       # group 1 back to Position: that Pid is reset
        k = next(steps)
        if rnd:
            eng.update_record(k, 5)
        else:
            eng.update(k, 5)
        ora.update(k)
        compare64(eng, ora, f"round {rnd}: position subset (fused launches)", tol)
        assert np.abs(eng.pid_debug() - ora.pid_debug()).max() < 1e-6, f"pid topic, round {rnd}"
        for e in (eng, ora):
            e.set_velocity_command(v[::-1].copy(), mask=(grp >= 1).astype(np.uint8))  # Position / Force / Position -> Velocity: the velocity Pid reset
        k = next(steps)
        eng.update(k, 1), ora.update(k)
        compare64(eng, ora, f"round {rnd}: entering Velocity mode", tol)
    eng.reset(), ora.reset()
    v = hold_commands(rng, B, cables, eps)
    eng.set_velocity_command(v), ora.set_velocity_command(v)
    eng.update(25, 3), ora.update(25)
    compare64(eng, ora, "unmasked Joy after a world reset", tol)


@pytest.mark.parametrize("per_robot", [False, True])
def test_fp64_trajectory_record_and_schedule(pkg, oracle, per_robot):
    """cdpr_update_record on precision = 64 handles (round 5): every step's observables kept, in double, equal to what step-
    by-step read-outs give (same kernels, same bits) - the role-split kernel's one-step launches (uniform FK + TD handle) and
    the one-wave kernel's fused launches (per-robot handle) - and a jointVelocities schedule queued with one call."""
    B, n, T = 70, 8, 37
    rng = np.random.default_rng(86)
    model = pkg.eight_cable_model()
    cfg = pkg.Config(model=model, batch=B, stages=3, precision=64, perRobotCommands=per_robot)
    pose = perturbed_poses(model, B, rng, 0.02, 0.05)
    a, ora = pair64(pkg, oracle, cfg, pose)
    b, _ = pair64(pkg, oracle, cfg, pose)
    v = rng.uniform(-0.03, 0.03, (B, n)).astype(np.float32)
    for e in (a, b, ora):
        e.update(6)
        e.set_velocity_command(v)
    rec = a.update_record(T, 5)
    assert rec["pose"].dtype == np.float64 and rec["effort"].shape == (T, B, n)
    for j in range(T):
        b.update(1), ora.update(1)
        q, qd, eff, p7, t6 = b.observables_f64()
        assert np.array_equal(rec["effort"][j], eff) and np.array_equal(rec["pose"][j], p7) and np.array_equal(rec["velocity"][j], qd), j
        if j % 9 == 0:
            assert np.abs(rec["effort"][j] - ora.joint_states()[2]).max() <= TOL64["eff"]
    for x, y in zip(a.observables_f64() + a.raw_state_f64(), b.observables_f64() + b.raw_state_f64()):
        assert np.array_equal(x, y)
    compare64(a, ora, "after the record")
    # a schedule of 4 batches, 10 steps each, with its record
    sched = rng.uniform(-0.03, 0.03, (4, B, n)).astype(np.float32)
    image = a.observable_image_bytes()
    d_sched, d_rec = a.device_upload(sched), a.device_alloc(image * 40)
    a.update_scheduled(40, 10, d_sched, d_rec, image * 40)
    for j in range(4):
        b.set_velocity_command(sched[j]), ora.set_velocity_command(sched[j])
        b.update(10), ora.update(10)
    for x, y in zip(a.observables_f64() + a.raw_state_f64(), b.observables_f64() + b.raw_state_f64()):
        assert np.array_equal(x, y)
    last = a.device_download(d_rec, (40, image), dtype=np.uint8)[39]
    import ctypes as C

    from cdpr_simulation_amd._native import lib

    eff = np.empty((B, n))
    assert lib().cdpr_decode_observables_f64(a._h, last.ctypes.data_as(C.c_void_p), None, None, eff.ctypes.data_as(C.POINTER(C.c_double)), None, None) == 0
    assert np.array_equal(eff, a.observables_f64()[2])
    compare64(a, ora, "after the schedule")


def test_fp64_facade_runs_n_steps_as_one_launch_chain(pkg, oracle):
    """The facade's update(n) on a precision = 64 handle: one launch chain into a trajectory record of doubles, n messages
    per topic (round 4: n device round trips), float64 arrays equal to the step-by-step facade's."""
    cfg = pkg.Config(batch=3, precision=64)
    plugs = [pkg.CdprGazeboPlugin(), pkg.CdprGazeboPlugin()]
    got = [[], []]
    for pl, g in zip(plugs, got):
        pl.Load(cfg)
        pl.bus.subscribe("jointStates", g.append)
    gen = pkg.stimulus.sine_velocity(4)
    for k in range(6):
        cmd = next(gen)
        for pl in plugs:
            pl.bus.publish("jointVelocities", pkg.Joy(axes=cmd))
        plugs[0].update(25)
        for _ in range(25):
            plugs[1].update(1)
    assert len(got[0]) == len(got[1]) == 149
    for m0, m1 in zip(got[0], got[1]):
        assert m0.header.stamp == m1.header.stamp and m0.effort.dtype == np.float64
        assert np.array_equal(m0.effort, m1.effort) and np.array_equal(m0.position, m1.position)


@pytest.mark.parametrize("cables,stages,B,split", [(8, 3, 130, None), (4, 0, 70, None), (8, 3, 5000, "0"), (8, 3, 17000, None), (6, 3, 100, "2"), (6, 3, 75, None), (7, 3, 90, None),
                                                  (7, 3, 100, "2")])
def test_fp64_hold_branch(pkg, oracle, monkeypatch, cables, stages, B, split):
    """velocityEpsilon >= 0 in the reference's own precision (round 5: the HOLD instantiations of the fp64 kernel): both Pids of
    every cable alive, cables drifting into the hold branch and back (their windows sampled at non-uniform times: the
    derivative is a least-squares fit on the real stamps in double), then Position mode (the position Pid reset), Force
    mode, Velocity again (the velocity Pid reset), fused launches and the trajectory record in between, the `pid` topic -
    against the fp64 oracle.  The fit here runs on orthogonal polynomials, the oracle's on normal equations in centred
    time: two double formulations of an ill-conditioned step (a window with a gap), hence the effort tolerance.
    Kernels: FK + TD handles take the role-split kernel's HOLD instantiations (its LDS build to 16 384 robots, its lean build
    beyond; CDPR_F64_SPLIT forces one or, "0", the one-wave kernel); the others the one-wave kernel's.  Round 6: the role-split
    kernels step the cables in passes (4 in the LDS build, 2 in the lean one; steady cables through the straight-line path, every
    other call through the per-cable code): 6 and 7 cables end on a short pass that repeats its last cable."""
    from test_gpu_general_matrix import hold_commands

    if split is not None:
        monkeypatch.setenv("CDPR_F64_SPLIT", split)

    eps = 0.004
    rng = np.random.default_rng(640 + cables)
    from dataclasses import replace

    full = pkg.eight_cable_model()
    model = pkg.cube_model() if cables == 4 else replace(full, frame_anchors=full.frame_anchors[:cables], platform_anchors=full.platform_anchors[:cables])
    cfg = pkg.Config(model=model, batch=B, stages=stages | pkg._abi.STAGE_PID_DEBUG, velocityEpsilon=eps, precision=64)
    eng, ora = pair64(pkg, oracle, cfg, perturbed_poses(model, B, rng, 0.02, 0.05).astype(np.float64))
    tol = dict(TOL64, eff=2e-7, twist=1e-10, qd=1e-10, pose=1e-12, q=1e-12)
    eng.update(3), ora.update(3)
    for j in range(7):
        cmd = hold_commands(rng, B, cables, eps)
        eng.set_velocity_command(cmd), ora.set_velocity_command(cmd)
        k = [9, 14, 3, 25, 12, 6, 31][j]
        if j % 3 == 1:
            eng.update(k, 5)
        elif j % 3 == 2:
            eng.update_record(k, 4)
        else:
            for _ in range(k):
                eng.update(1)
        ora.update(k)
        compare64(eng, ora, f"hold round {j}", tol)
        assert np.abs(eng.pid_debug() - ora.pid_debug()).max() < 1e-6, f"pid topic, round {j}"
    p = rng.uniform(-0.004, 0.004, (B, cables)).astype(np.float32)
    eng.set_position_command(p), ora.set_position_command(p)
    eng.update(40), ora.update(40)
    compare64(eng, ora, "position mode", tol)
    f = (7.0 + rng.uniform(-0.5, 0.5, (B, cables))).astype(np.float32)
    eng.set_force_command(f), ora.set_force_command(f)
    eng.update(7), ora.update(7)
    compare64(eng, ora, "force mode", tol)
    cmd = hold_commands(rng, B, cables, eps)
    eng.set_velocity_command(cmd), ora.set_velocity_command(cmd)
    eng.update(33, 11), ora.update(33)
    compare64(eng, ora, "velocity mode again", tol)
    eng.reset(), ora.reset()
    eng.set_velocity_command(cmd), ora.set_velocity_command(cmd)
    eng.update(20), ora.update(20)
    compare64(eng, ora, "after a world reset", tol)


@pytest.mark.parametrize("cables,stages,B", [(8, 3, 150), (4, 0, 70)])
def test_fp64_travel_stop(pkg, oracle, cables, stages, B):
    """The inelastic joint stop (cdpr_config_t.travel_stop sweeps) in double: the TSTOP instantiations of the one-wave kernel
    (round 5).  Both sides compute in double, so - unlike the fp32 test (test_gpu_parity.py::test_travel_stop_against_the_oracle) -
    the thresholds fall on the same step and the comparison stays tight through the contacts; one-step and fused launches, the
    limit flags, the joints resting on their stops."""
    rng = np.random.default_rng(77 + cables)
    model = pkg.eight_cable_model() if cables == 8 else pkg.cube_model()
    lim = 0.004
    model.travel_lower, model.travel_upper, model.travel_stop = -lim, lim, 4
    cfg = pkg.Config(model=model, batch=B, stages=stages, precision=64)
    eng, ora = pair64(pkg, oracle, cfg, perturbed_poses(model, B, rng, 0.002, 0.01).astype(np.float64))
    jac = oracle.ik(cfg.to_struct(), model.home_pose())[3]
    tw = np.concatenate([rng.uniform(-0.05, 0.05, (B, 3)), rng.uniform(-0.2, 0.2, (B, 3))], axis=1)
    cmd = (-(jac @ tw.T).T).astype(np.float32)
    eng.update(10), ora.update(10)
    eng.set_velocity_command(cmd), ora.set_velocity_command(cmd)
    tol = dict(TOL64, pose=1e-11, q=1e-11, twist=1e-9, qd=1e-9, eff=1e-6)
    for k in range(10):
        if k % 2:
            eng.update(30, 10)
        else:
            eng.update(30)
        ora.update(30)
        compare64(eng, ora, f"{30 * (k + 1)} steps", tol)
    q = eng.observables_f64()[0]
    assert np.abs(q).max() < lim + 1.5e-4 and (np.abs(q) > lim - 1e-5).any()
    assert (eng.limit_state() != 0).mean() > 0.3
    assert np.array_equal(eng.limit_state(), ora.limit_state())


@pytest.mark.parametrize("variant", ["cascades", "cascades_hold", "noclamp", "noclamp_hold", "per_robot_all"])
def test_fp64_rest_of_pid_update(pkg, oracle, variant):
    """What else Pid::update holds, in double on the HOLD instantiations (round 5): the biquad cascades on the P and the D input
    (Pid.cpp:27-44; different depths on the two Pids), cmdLimit = 0 (Pid.cpp:175-184: without the clamp the Pid returns its stale
    mCmd member plus the anti-windup increment and restores the integral - the oracle follows the reference there), with and
    without the hold branch, and all of it together on a per-robot handle whose two Pids also fit different windows."""
    from test_gpu_general_matrix import hold_commands

    B, n = 150, 8
    rng = np.random.default_rng(300 + len(variant))
    model = pkg.eight_cable_model()
    hold = variant.endswith("_hold") or variant == "per_robot_all"
    eps = 0.004 if hold else -0.001
    pr = variant == "per_robot_all"
    cfg = pkg.Config(model=model, batch=B, stages=3 | pkg._abi.STAGE_PID_DEBUG, precision=64, velocityEpsilon=eps, perRobotCommands=pr)
    vc, pc = cfg.velocityController, cfg.positionController
    if variant.startswith("cascades") or pr:
        for f, depth in ((vc.pFilter, 1), (vc.dFilter, 2), (pc.pFilter, 3), (pc.dFilter, 1)):
            f.cascade, f.relCutoff, f.quality = depth, 0.05, 0.5
        vc.pGain, vc.iGain, vc.dGain = 4.0, 40.0, 0.01  # (a gentle loop: the filters' lag with the shipped gains rings)
    if variant.startswith("noclamp"):
        vc.cmdLimit = 0.0
    if pr:
        pc.cmdLimit = 0.0
        pc.dBufferLength, pc.dDegree = 7, 1
    eng, ora = pair64(pkg, oracle, cfg, perturbed_poses(model, B, rng, 0.02, 0.05).astype(np.float64))
    tol = dict(TOL64, eff=2e-7, twist=1e-10, qd=1e-10, pose=1e-12, q=1e-12)
    eng.update(3), ora.update(3)
    grp = np.arange(B) % 3
    for j in range(5):
        cmd = hold_commands(rng, B, n, abs(eps) if hold else 0.002)
        kw = dict(mask=(grp != j % 3).astype(np.uint8)) if pr else {}
        eng.set_velocity_command(cmd, **kw), ora.set_velocity_command(cmd, **kw)
        if pr and j == 2:
            p = rng.uniform(-0.004, 0.004, (B, n)).astype(np.float32)
            eng.set_position_command(p, mask=(grp == 0).astype(np.uint8)), ora.set_position_command(p, mask=(grp == 0).astype(np.uint8))
        k = [14, 9, 25, 3, 17][j]
        eng.update(k, 1 if j % 2 else 5), ora.update(k)
        compare64(eng, ora, f"{variant}, round {j}", tol)
        assert np.abs(eng.pid_debug() - ora.pid_debug()).max() < 1e-6, f"{variant}: pid topic, round {j}"
    p = rng.uniform(-0.004, 0.004, (B, n)).astype(np.float32)
    eng.set_position_command(p), ora.set_position_command(p)
    eng.update(30), ora.update(30)
    compare64(eng, ora, f"{variant}, position mode", tol)
    eng.reset(), ora.reset()
    cmd = hold_commands(rng, B, n, abs(eps) if hold else 0.002)
    eng.set_velocity_command(cmd), ora.set_velocity_command(cmd)
    eng.update(20, 4), ora.update(20)
    compare64(eng, ora, f"{variant}, after a world reset", tol)


def test_fp64_refuses_nothing_of_the_controller_any_more(pkg):
    """Through round 6 precision = 64 refused combinations of its own features (the optional physics, per-robot modes, long windows,
    the hold branch); the last of them - windows beyond 11 samples with the hold branch / cascades / cmd_limit 0, or on a per-robot
    handle whose Pids fit different windows - run on HOLD records of 32 samples now; the MPC rollout runs in double on every handle kind;
    nine to twelve cables are served where the fp32 kernels serve them (uniform-mode handles on the register-resident path)."""
    stop = pkg.eight_cable_model()
    stop.travel_lower, stop.travel_upper, stop.travel_stop = -0.01, 0.01, 2
    names = []
    for kw, vel_window in ((dict(velocityEpsilon=0.01), 16), (dict(perRobotCommands=True, velocityEpsilon=0.01), 16), (dict(perRobotCommands=True), 16), (dict(model=stop, velocityEpsilon=0.01), 32)):
        cfg = pkg.Config(batch=4, precision=64, **kw)
        cfg.velocityController.dBufferLength = vel_window  # (the position Pid keeps 11 samples: on a per-robot handle that alone asks for both records)
        eng = pkg.Engine(cfg, 0)
        eng.update(3)
        names.append(eng.kernel_name)
        eng.close()
    assert names == ["cdpr_step_kernel_f64<4, HOLD = 2, HW = 32>", "cdpr_step_kernel_f64<4, PR, HOLD = 2, HW = 32>", "cdpr_step_kernel_f64<4, PR, HOLD = 2, HW = 32>",
                     "cdpr_step_kernel_f64<8, HOLD = 2, TSTOP, HW = 32>"], names
    twelve = pkg.Config(model=pkg.twelve_cable_model(), batch=4, precision=64, velocityEpsilon=0.01)  # (nine to twelve cables: uniform-mode handles on the
    with pytest.raises(pkg.CdprError) as ei:                                                           #  register-resident path, in double as in float)
        pkg.Engine(twelve, 0)
    assert ei.value.code == pkg._abi.ERR_UNSUPPORTED


@pytest.mark.parametrize("variant", ["hold", "hold_cascades_noclamp", "per_robot_different_windows", "hold_physics"])
def test_fp64_hold_branch_with_long_windows(pkg, oracle, variant):
    """The hold branch, the cascades and cmd_limit = 0 with derivative windows of 12 .. 32 samples in double (later in round 6; refused
    before): the HOLD = 2 instantiations of the one-wave kernel over Pid records of 32 samples and their 32 stamps - the fixed filter
    by ring slot for a uniform window, the least-squares fit on the real stamps for a window with a gap, through the fill of the long
    windows, cables crossing epsilon in both directions, mode changes (masked, on the per-robot variant, whose two Pids fit windows of
    24 and 11 samples), fused launches and the record - against the fp64 oracle."""
    from dataclasses import replace
    from test_gpu_general_matrix import hold_commands

    B, cables = 110, 8
    pr = variant.startswith("per_robot")
    eps = -0.001 if pr else 0.004
    rng = np.random.default_rng(960 + len(variant))
    model = pkg.eight_cable_model()
    if variant == "hold_physics":
        model = replace(model, passive_damping=0.05, leg_inertia=0.02, travel_lower=-0.004, travel_upper=0.004, travel_stop=2)
    cfg = pkg.Config(model=model, batch=B, stages=3 | pkg._abi.STAGE_PID_DEBUG, precision=64, velocityEpsilon=eps, perRobotCommands=pr)
    vc, pc = cfg.velocityController, cfg.positionController
    vc.dBufferLength, vc.dDegree = (24, 3) if pr else (20, 2)
    if variant == "hold":
        pc.dBufferLength, pc.dDegree = 32, 2
    if variant == "hold_cascades_noclamp":
        vc.pFilter.cascade, vc.pFilter.relCutoff, vc.dFilter.cascade, vc.dFilter.relCutoff = 2, 0.2, 1, 0.25
        pc.cmdLimit = 0.0
    want = f"cdpr_step_kernel_f64<8, {'PR, ' if pr else ''}HOLD = 2{', TSTOP' if variant == 'hold_physics' else ''}, HW = 32>"
    assert pkg.plan_kernel(cfg, 1) == want == pkg.plan_kernel(cfg, 7), (pkg.plan_kernel(cfg, 1), want)
    eng, ora = pair64(pkg, oracle, cfg, perturbed_poses(model, B, rng, 0.003 if variant == "hold_physics" else 0.02, 0.03).astype(np.float64))
    tol = dict(TOL64, eff=1e-6, twist=1e-9, qd=1e-9, pose=1e-11, q=1e-11)
    grp = np.arange(B) % 3
    eng.update(4), ora.update(4)
    for rnd, k in enumerate([27, 6, 41, 13, 35]):
        v = hold_commands(rng, B, cables, abs(eps) if not pr else 0.002)
        if pr:
            p = rng.uniform(-0.004, 0.004, (B, cables)).astype(np.float32)
            for e in (eng, ora):
                e.set_velocity_command(v, mask=(grp != rnd % 3).astype(np.uint8))
                e.set_position_command(p, mask=(grp == rnd % 3).astype(np.uint8))
        elif rnd == 3:
            p = rng.uniform(-0.004, 0.004, (B, cables)).astype(np.float32)
            eng.set_position_command(p), ora.set_position_command(p)  # Position mode: that Pid's (long) window starts again
        else:
            eng.set_velocity_command(v), ora.set_velocity_command(v)
        if rnd == 1:
            eng.update(k, 3)
        elif rnd == 2:
            eng.update_record(k, 5)
        else:
            for _ in range(k):
                eng.update(1)
        ora.update(k)
        compare64(eng, ora, f"{variant}, round {rnd}", tol)
        assert np.abs(eng.pid_debug() - ora.pid_debug()).max() < 1e-5, f"pid topic, round {rnd}"
    assert eng.kernel_name == want


@pytest.mark.parametrize("variant", ["per_robot", "hold", "per_robot_hold", "hold_cascades"])
def test_fp64_optional_physics_with_per_robot_modes_and_the_hold_branch(pkg, oracle, variant):
    """The joint stop and the lumped legs in double TOGETHER with per-robot modes and / or the hold branch (round 6: refused before; the
    TSTOP instantiations now take PR and HOLD - the world step does not care who set the forces): velocity Joys with cables at or
    below epsilon, position and setForce commands on subsets (per-robot variants), joints running into their stops, one-step and
    fused launches, the trajectory record - against the fp64 oracle."""
    from dataclasses import replace
    from test_gpu_general_matrix import hold_commands

    B, cables = 140, 8
    pr = variant.startswith("per_robot")
    hold = "hold" in variant
    eps = 0.004 if hold else -0.001
    rng = np.random.default_rng(900 + len(variant))
    model = replace(pkg.eight_cable_model(), inertia=(0.9, 1.1, 1.0, 0.05, -0.03, 0.02), passive_damping=0.05, leg_inertia=0.02, cable_axial_mass=0.005, anchor_point_mass=0.01,
                    anchor_inertia=0.005, travel_lower=-0.004, travel_upper=0.004, travel_stop=3)
    cfg = pkg.Config(model=model, batch=B, stages=3 | pkg._abi.STAGE_PID_DEBUG, precision=64, velocityEpsilon=eps, perRobotCommands=pr, gravity=(0.3, -0.2, -9.7))
    if variant == "hold_cascades":
        cfg.velocityController.pFilter.cascade, cfg.velocityController.pFilter.relCutoff = 2, 0.2
        cfg.velocityController.dFilter.cascade, cfg.velocityController.dFilter.relCutoff = 1, 0.25
    level = 2 if variant == "hold_cascades" else 1
    want = f"cdpr_step_kernel_f64<8, {'PR, ' if pr else ''}{f'HOLD = {level}, ' if hold else ''}TSTOP>"
    assert pkg.plan_kernel(cfg, 1) == want, (pkg.plan_kernel(cfg, 1), want)
    eng, ora = pair64(pkg, oracle, cfg, perturbed_poses(model, B, rng, 0.003, 0.02).astype(np.float64))
    tol = dict(TOL64, pose=1e-11, q=1e-11, twist=1e-9, qd=1e-9, eff=1e-6)
    eng.update(8), ora.update(8)
    grp = np.arange(B) % 3
    for rnd, k in enumerate([21, 34, 9, 27]):
        v = (1.5 * hold_commands(rng, B, cables, abs(eps) if hold else 0.002)).astype(np.float32)
        if pr:
            p = rng.uniform(-0.004, 0.004, (B, cables)).astype(np.float32)
            f = (7.0 + rng.uniform(-0.5, 0.5, (B, cables))).astype(np.float32)
            for e in (eng, ora):
                e.set_velocity_command(v, mask=(grp != rnd % 3).astype(np.uint8))
                if rnd % 2:
                    e.set_force_command(f, mask=(grp == rnd % 3).astype(np.uint8))
                else:
                    e.set_position_command(p, mask=(grp == rnd % 3).astype(np.uint8))
        else:
            eng.set_velocity_command(v), ora.set_velocity_command(v)
        if rnd == 1:
            eng.update(k, 6)
        elif rnd == 2:
            eng.update_record(k, 3)
        else:
            eng.update(k)
        ora.update(k)
        compare64(eng, ora, f"{variant}, round {rnd}", tol)
        assert np.array_equal(eng.limit_state(), ora.limit_state())
    assert eng.kernel_name == want
    assert (eng.limit_state() != 0).any()


def test_zz_report_measured_agreement():
    print("fp64 kernel vs fp64 oracle, worst over this module:", {k: f"{v:.2e}" for k, v in WORST.items()})


@pytest.mark.parametrize("cables,build", [(8, "1"), (6, "1"), (8, "2"), (7, "2")])
def test_role_split_fp64_kernel_against_the_one_wave_kernel(pkg, monkeypatch, cables, build):
    """One-step launches of FK + TD fp64 handles up to one workgroup per CU run on cdpr_split_kernel_f64 (estimator wave +
    controller wave per 64 robots; CDPR_F64_SPLIT=0 keeps the one-wave kernel, 1 forces the LDS-cached build, 2 the lean
    build that serves batches beyond one workgroup per CU): same statements in the same order over the same LDS columns.  Velocity, Position and Force mode, the first world step, a ragged last block, the `pid` topic, the
    travel-limit flags; fused launches (one-wave kernel on both handles) in between.  Same bits: every multiply-add of the
    fp64 kernels is an explicit fma (the variable is read per call, so both builds run in this process)."""
    from dataclasses import replace

    B = 64 * 3 + 11
    rng = np.random.default_rng(640 + cables)
    full = pkg.eight_cable_model()
    model = replace(full, frame_anchors=full.frame_anchors[:cables], platform_anchors=full.platform_anchors[:cables], travel_lower=-0.004, travel_upper=0.004)
    cfg = pkg.Config(model=model, batch=B, stages=3 | pkg._abi.STAGE_PID_DEBUG, precision=64)
    pose = perturbed_poses(model, B, rng, 0.02, 0.05)
    v = rng.uniform(-0.03, 0.03, (B, cables)).astype(np.float32)
    p = rng.uniform(-0.003, 0.003, (B, cables)).astype(np.float32)
    f = rng.uniform(5.0, 25.0, (B, cables)).astype(np.float32)
    out = []
    for split in ("0", build):
        monkeypatch.setenv("CDPR_F64_SPLIT", split)
        eng = pkg.Engine(cfg, 0)
        eng.set_platform_state(pose7=pose)
        eng.update(1)
        snaps = []
        for kind, cmd, k in (("vel", v, 17), ("pos", p, 13), ("frc", f, 5), ("vel", -v, 12)):
            getattr(eng, {"vel": "set_velocity_command", "pos": "set_position_command", "frc": "set_force_command"}[kind])(cmd)
            for _ in range(k):
                eng.update(1)
            if kind == "pos":
                eng.update(6, 3)
            snaps.append(eng.observables_f64() + eng.raw_state_f64() + (eng.pid_debug(), eng.limit_state()))
        out.append(snaps)
        eng.close()
    worst = 0.0
    for a, b in zip(*out):
        for x, y in zip(a, b):
            if x.dtype.kind == "f":
                worst = max(worst, float(np.abs(x.astype(np.float64) - y.astype(np.float64)).max()))
            else:
                assert np.array_equal(x, y)
    print(f"fp64 role-split (build {build}) vs one-wave kernel, n = {cables}: worst difference {worst:.3e}")
    assert worst == 0.0


@pytest.mark.parametrize("cables,build", [(8, "1"), (8, "2"), (6, "1"), (7, "2")])
def test_role_split_fp64_hold_controller_against_the_one_wave_kernel(pkg, monkeypatch, cables, build):
    """The hold branch on the role-split fp64 kernels (round 6): passes of cables whose rows are requested together, steady cables
    through the straight-line path (weights by ring slot from a rotated table, the integral's clamp by selects, unconditional stores),
    every other call through the per-cable code - against the one-wave kernel, which has only the per-cable code: same bits through
    window fills, held cables, cables crossing epsilon in both directions (windows with a gap: the fit), Position and Force mode,
    saturating commands (the clamp and its back-calculation), a ragged last block."""
    from dataclasses import replace
    from test_gpu_general_matrix import hold_commands

    B, eps = 64 * 2 + 9, 0.004
    rng = np.random.default_rng(700 + cables)
    full = pkg.eight_cable_model()
    model = replace(full, frame_anchors=full.frame_anchors[:cables], platform_anchors=full.platform_anchors[:cables])
    cfg = pkg.Config(model=model, batch=B, stages=3 | pkg._abi.STAGE_PID_DEBUG, precision=64, velocityEpsilon=eps)
    pose = perturbed_poses(model, B, rng, 0.02, 0.05)
    script = []
    for j in range(6):
        script.append(("vel", hold_commands(rng, B, cables, eps), [14, 3, 25, 12, 1, 13][j]))
    script.append(("vel", (40.0 * hold_commands(rng, B, cables, eps)).astype(np.float32), 15))  # far beyond what the joint follows: the clamps
    script.append(("pos", rng.uniform(-0.004, 0.004, (B, cables)).astype(np.float32), 14))
    script.append(("frc", rng.uniform(5.0, 25.0, (B, cables)).astype(np.float32), 4))
    script.append(("vel", hold_commands(rng, B, cables, eps), 16))
    out = []
    for split in ("0", build):
        monkeypatch.setenv("CDPR_F64_SPLIT", split)
        eng = pkg.Engine(cfg, 0)
        eng.set_platform_state(pose7=pose)
        eng.update(1)
        snaps = []
        for kind, cmd, k in script:
            getattr(eng, {"vel": "set_velocity_command", "pos": "set_position_command", "frc": "set_force_command"}[kind])(cmd)
            for _ in range(k):
                eng.update(1)
            snaps.append(eng.observables_f64() + eng.raw_state_f64() + (eng.pid_debug(),))
        out.append((snaps, eng.kernel_name))
        eng.close()
    assert "split" in out[1][1] and "split" not in out[0][1], (out[0][1], out[1][1])
    worst = 0.0
    for a, b in zip(out[0][0], out[1][0]):
        for x, y in zip(a, b):
            worst = max(worst, float(np.abs(x.astype(np.float64) - y.astype(np.float64)).max()))
    print(f"fp64 hold controller, role-split (build {build}) vs one-wave kernel, n = {cables}: worst difference {worst:.3e}")
    assert worst == 0.0


@pytest.mark.parametrize("entered_from", ["velocity", "position", "world_step_0", "velocity_long_window", "velocity_lumped_legs", "velocity_hold_branch", "position_hold_branch",
                                          "velocity_hold_branch_long_window_cascades"])
def test_fp64_rollout_against_the_oracle(pkg, oracle, entered_from):
    """cdpr_rollout_velocity on a precision = 64 handle (round 6; refused before): every (robot, sampled sequence) steps a private
    copy of the robot's state through the handle's own fp64 step kernel, one launch per step of the horizon, the cost accumulated
    in double.  Against the oracle's rollout (the cost leaves as float32: agreement to its rounding), entered from Velocity mode
    (the Pid's history carries on), from Position mode (a Joy on jointVelocities resets the velocity Pid, JFC.cpp:113-115) and at
    world step 0 (no force at t = 0, JFC.cpp:61-66); the handle itself is not advanced by a rollout.  End of round 6: with the hold
    branch live as well (the trajectories carry both Pids' records; from Position mode the velocity Pid's records are the ones cleared),
    also over records of 32 samples with a cascade."""
    B, n, S, H = 20, 8, 12, 16
    rng = np.random.default_rng(660)
    cfg = pkg.Config(model=pkg.eight_cable_model(), batch=B, stages=3, precision=64)
    if entered_from == "velocity_lumped_legs":  # (the optional physics: the TSTOP instantiation steps the trajectories)
        cfg.model.passive_damping, cfg.model.leg_inertia, cfg.model.anchor_point_mass = 0.05, 0.02, 0.01
        cfg.model.travel_lower, cfg.model.travel_upper, cfg.model.travel_stop = -0.05, 0.05, 2
    if entered_from == "velocity_long_window":  # (a 20-sample window: the kernel with a ring of 31, its rows copied per trajectory)
        cfg.velocityController.dBufferLength, cfg.velocityController.dDegree = 20, 3
    if "hold_branch" in entered_from:  # (end of round 6: both Pids' records of every cable travel with a trajectory; sampled commands cross epsilon)
        cfg.velocityEpsilon = 0.012
    if entered_from.endswith("long_window_cascades"):
        cfg.velocityController.dBufferLength, cfg.velocityController.dDegree = 18, 2
        cfg.velocityController.pFilter.cascade, cfg.velocityController.pFilter.relCutoff = 1, 0.2
    pose = np.tile(cfg.model.home_pose(), (B, 1))
    pose[:, :3] += rng.uniform(-0.02, 0.02, (B, 3))
    eng, ora = pkg.Engine(cfg, 0), oracle.OracleSim(cfg.to_struct(), oracle.DERIV_EXACT)
    eng.set_platform_state_f64(pose7=pose), ora.set_platform_state(pose7=pose)
    first = rng.uniform(-0.03, 0.03, (B, n)).astype(np.float32)
    if entered_from.startswith("velocity"):
        for sim in (eng, ora):
            sim.set_velocity_command(first)
            sim.update(23)
    elif entered_from.startswith("position"):
        for sim in (eng, ora):
            sim.set_position_command((0.1 * first).astype(np.float32))
            sim.update(17)
    cmds = (rng.uniform(-0.03, 0.03, (B, H, 1, n)) + rng.normal(0.0, 0.01, (B, H, S, n))).astype(np.float32)
    ref = pose[:, :3].astype(np.float32)
    before = eng.raw_state_f64()
    cost = eng.rollout_velocity(cmds, ref)
    ocost = ora.rollout_velocity(cmds, ref.astype(np.float64))
    assert cost.shape == (B, S) and np.isfinite(cost).all()
    assert np.abs(cost - ocost).max() <= 3e-7 * np.abs(ocost).max(), float(np.abs(cost - ocost).max() / np.abs(ocost).max())
    assert (cost.argmin(axis=1) == ocost.argmin(axis=1)).all()
    after = eng.raw_state_f64()
    assert all(np.array_equal(x, y) for x, y in zip(before, after)) and eng.step_count == ora.step_count
    eng.update(5), ora.update(5)  # ... and carries on as if nothing had happened
    assert np.abs(eng.observables_f64()[3] - ora.platform_state()[0]).max() < 1e-12


@pytest.mark.parametrize("variant", ["per_robot", "per_robot_hold_branch"])
def test_fp64_rollout_on_per_robot_handles(pkg, oracle, variant):
    """cdpr_rollout_velocity in double on per-robot handles (end of round 6; the fp32 paths had it): robots in different modes and
    with Pids reset at different times - a robot that is not in Velocity mode enters it with the rollout's first Joy (its Pid reset,
    its call count 0), the others carry their windows on; every trajectory has its own mode / call-count byte.  Against the oracle
    (B independent JointForceCalculator sets); the handle's own state and modes are left alone."""
    from test_gpu_general_matrix import hold_commands

    B, n, S, H = 90, 8, 5, 14
    hold = variant.endswith("hold_branch")
    eps = 0.004 if hold else -0.001
    rng = np.random.default_rng(980 + len(variant))
    model = pkg.eight_cable_model()
    cfg = pkg.Config(model=model, batch=B, stages=3, precision=64, perRobotCommands=True, velocityEpsilon=eps)
    eng, ora = pair64(pkg, oracle, cfg, perturbed_poses(model, B, rng, 0.02, 0.03).astype(np.float64))
    grp = rng.integers(0, 3, B)
    v = hold_commands(rng, B, n, eps) if hold else rng.uniform(-0.03, 0.03, (B, n)).astype(np.float32)
    p = rng.uniform(-0.003, 0.003, (B, n)).astype(np.float32)
    f = (7.0 + rng.uniform(-0.5, 0.5, (B, n))).astype(np.float32)
    for e in (eng, ora):
        e.update(6)
        e.set_velocity_command(v, mask=(grp >= 1).astype(np.uint8))   # group 0 stays in Position mode
    eng.update(17), ora.update(17)
    for e in (eng, ora):
        e.set_force_command(f, mask=(grp == 2).astype(np.uint8))      # group 2: Force mode; group 1 keeps its velocity Pid's window
    eng.update(9), ora.update(9)
    cmds = (rng.uniform(-0.03, 0.03, (B, H, 1, n)) + rng.normal(0.0, 0.01, (B, H, S, n))).astype(np.float32)
    ref = (eng.observables_f64()[3][:, :3] + np.array([0.0, 0.0, 0.01])).astype(np.float32)
    before = eng.raw_state_f64()
    cost, ocost = eng.rollout_velocity(cmds, ref), ora.rollout_velocity(cmds, ref.astype(np.float64))
    assert cost.shape == (B, S) and np.isfinite(cost).all()
    assert np.abs(cost - ocost).max() <= 3e-7 * np.abs(ocost).max(), float(np.abs(cost - ocost).max() / np.abs(ocost).max())
    after = eng.raw_state_f64()
    assert all(np.array_equal(x, y) for x, y in zip(before, after)) and eng.step_count == ora.step_count
    eng.update(8), ora.update(8)  # modes and Pids of the handle as they were: Position / Velocity / Force groups carry on
    compare64(eng, ora, f"{variant}: after the rollout", dict(TOL64, eff=2e-7, twist=1e-10, qd=1e-10, pose=1e-12, q=1e-12))


@pytest.mark.parametrize("variant", ["per_robot", "physics", "per_robot_physics"])
def test_fp64_long_windows_with_per_robot_modes_and_the_optional_physics(pkg, oracle, variant):
    """Derivative windows of 12 .. 32 samples in double TOGETHER with per-robot modes (both Pids on the long window, as every
    register-resident per-robot handle has them on one) and / or the joint stop and the lumped legs (later in round 6; refused
    before): the W = 31 instantiations take PR and TSTOP.  Through the window fill per robot (a robot's Pid is reset when its mode
    changes: masked commands at different times), fused launches and the record - against the fp64 oracle."""
    from dataclasses import replace

    B, cables, nbuf, degree = 100, 8, 20, 3
    pr, phys = "per_robot" in variant, "physics" in variant
    rng = np.random.default_rng(940 + len(variant))
    model = pkg.eight_cable_model()
    if phys:
        model = replace(model, passive_damping=0.05, leg_inertia=0.02, anchor_point_mass=0.01, travel_lower=-0.004, travel_upper=0.004, travel_stop=2)
    cfg = pkg.Config(model=model, batch=B, stages=3, precision=64, perRobotCommands=pr)
    for c in (cfg.velocityController, cfg.positionController):
        c.dBufferLength, c.dDegree = nbuf, degree
    want = f"cdpr_step_kernel_f64<8, {'PR, ' if pr else ''}{'TSTOP, ' if phys else ''}W = 31>"
    assert pkg.plan_kernel(cfg, 1) == want == pkg.plan_kernel(cfg, 10), (pkg.plan_kernel(cfg, 1), want)
    eng, ora = pair64(pkg, oracle, cfg, perturbed_poses(model, B, rng, 0.003 if phys else 0.02, 0.02).astype(np.float64))
    tol = dict(TOL64, pose=1e-11, q=1e-11, twist=1e-9, qd=1e-9, eff=1e-6)
    grp = np.arange(B) % 3
    eng.update(5), ora.update(5)
    for rnd, k in enumerate([nbuf - 3, 9, nbuf + 7, 14]):
        v = rng.uniform(-0.04, 0.04, (B, cables)).astype(np.float32)
        if pr:
            p = rng.uniform(-0.004, 0.004, (B, cables)).astype(np.float32)
            for e in (eng, ora):
                e.set_velocity_command(v, mask=(grp != rnd % 3).astype(np.uint8))   # (a robot entering Velocity mode starts its long window again)
                e.set_position_command(p, mask=(grp == rnd % 3).astype(np.uint8))
        else:
            eng.set_velocity_command(v), ora.set_velocity_command(v)
        if rnd == 1:
            eng.update(k, 3)
        elif rnd == 3:
            eng.update_record(k, 7)
        else:
            for _ in range(k):
                eng.update(1)
        ora.update(k)
        compare64(eng, ora, f"{variant}, round {rnd}", tol)
    assert eng.kernel_name == want
    if phys:
        assert np.array_equal(eng.limit_state(), ora.limit_state())


@pytest.mark.parametrize("nbuf,degree,cables,stages", [(16, 3, 8, 3), (32, 2, 8, 3), (12, 4, 4, 0), (24, 1, 6, 1)])
def test_fp64_long_derivative_windows(pkg, oracle, nbuf, degree, cables, stages):
    """Derivative windows of 12 .. 32 samples with precision = 64 (round 6; Pid.h:135 allows any mDbufferLength, refused in double
    before): the plain one-wave fp64 kernel with a ring of 31 errors per cable.  Against the oracle through the window fill (the D
    term is 0 until nbuf samples are in, Pid.cpp:200-203), a mode change (the entered Pid is reset), several steps per launch and
    the trajectory record; the other Pid keeps the shipped 11-sample window."""
    from dataclasses import replace

    B = 70
    rng = np.random.default_rng(670 + nbuf)
    full = pkg.eight_cable_model()
    model = pkg.cube_model() if cables == 4 else replace(full, frame_anchors=full.frame_anchors[:cables], platform_anchors=full.platform_anchors[:cables])
    cfg = pkg.Config(model=model, batch=B, stages=stages, precision=64)
    cfg.velocityController.dBufferLength, cfg.velocityController.dDegree = nbuf, degree
    assert pkg.plan_kernel(cfg, 1) == f"cdpr_step_kernel_f64<{cables}, W = 31>" == pkg.plan_kernel(cfg, 10)
    pose = np.tile(model.home_pose(), (B, 1))
    pose[:, :3] += rng.uniform(-0.02, 0.02, (B, 3))
    eng, ora = pkg.Engine(cfg, 0), oracle.OracleSim(cfg.to_struct(), oracle.DERIV_EXACT)
    eng.set_platform_state_f64(pose7=pose), ora.set_platform_state(pose7=pose)

    def same(where, tol_p=1e-12, tol_e=1e-8):
        g = eng.observables_f64()
        dp, de = np.abs(g[3] - ora.platform_state()[0]).max(), np.abs(g[2] - ora.joint_states()[2]).max()
        assert dp < tol_p and de < tol_e, (where, dp, de)

    cmd = rng.uniform(-0.03, 0.03, (B, cables)).astype(np.float32)
    eng.update(9), ora.update(9)
    for sim in (eng, ora):
        sim.set_velocity_command(cmd)
    for k in range(nbuf + 6):  # step by step through the fill of the long window
        eng.update(1), ora.update(1)
        if k in (0, nbuf - 2, nbuf - 1, nbuf, nbuf + 5):
            same(f"fill step {k}")
    eng.update(45, 9), ora.update(45)  # several steps per launch: the ring turns more than once
    same("fused")
    for sim in (eng, ora):
        sim.set_position_command((0.1 * cmd).astype(np.float32))  # the position Pid (11 samples) is entered: reset
    eng.update(30), ora.update(30)
    same("position mode")
    for sim in (eng, ora):
        sim.set_velocity_command((-cmd).astype(np.float32))  # back: the velocity Pid starts its long window again
    rec = eng.update_record(nbuf + 8, 4)
    ora.update(nbuf + 8)
    same("record")
    assert np.array_equal(rec["effort"][-1], eng.observables_f64()[2])
    assert eng.kernel_name == f"cdpr_step_kernel_f64<{cables}, W = 31>"


@pytest.mark.parametrize("scale", [1.0, 30.0])
@pytest.mark.parametrize("cables,stages,stop", [(8, 3, False), (4, 0, False), (7, 3, True)])
def test_fp64_lumped_legs(pkg, oracle, cables, stages, stop, scale):
    """The lumped legs in double (round 6; refused with precision = 64 before): passive joint damping (cube.sdf:396) and the leg
    links' masses / inertias (cube.sdf:359-382) as a 6 x 6 mass matrix in the world step of the TSTOP instantiations - the fp32
    kernels' integrate_lumped_velocity restated in double - at the shipped link values and 30 x exaggerated, with a full inertia
    tensor and tilted gravity, alone and together with the joint stop; one-step and fused launches, the trajectory record."""
    from dataclasses import replace

    B = 90
    rng = np.random.default_rng(680 + cables)
    lumped = dict(passive_damping=0.01 * scale, leg_inertia=0.004 * scale, cable_axial_mass=0.001 * scale, anchor_point_mass=0.002 * scale,
                  anchor_inertia=0.001 * scale)
    eight = pkg.eight_cable_model()
    base = pkg.cube_model() if cables == 4 else replace(eight, frame_anchors=eight.frame_anchors[:cables], platform_anchors=eight.platform_anchors[:cables])
    model = replace(base, inertia=(0.9, 1.1, 1.0, 0.05, -0.03, 0.02), **lumped)
    if stop:
        model = replace(model, travel_lower=-0.004, travel_upper=0.004, travel_stop=3)
    cfg = pkg.Config(model=model, batch=B, stages=stages, precision=64, gravity=(0.3, -0.2, -9.7))
    assert pkg.plan_kernel(cfg, 1) == f"cdpr_step_kernel_f64<{cables}, TSTOP>"
    eng, ora = pair64(pkg, oracle, cfg, perturbed_poses(model, B, rng, 0.003 if stop else 0.03, 0.02).astype(np.float64))
    cmd = rng.uniform(-0.03, 0.03, (B, cables)).astype(np.float32)
    tol = dict(TOL64, pose=1e-11, q=1e-11, twist=1e-9, qd=1e-9, eff=1e-6)
    eng.update(20), ora.update(20)
    eng.set_velocity_command(cmd), ora.set_velocity_command(cmd)
    eng.update(45), ora.update(45)
    compare64(eng, ora, "one-step launches", tol)
    eng.update(40, 10), ora.update(40)
    compare64(eng, ora, "fused launches", tol)
    rec = eng.update_record(12, 4)
    ora.update(12)
    compare64(eng, ora, "record", tol)
    assert np.array_equal(rec["pose"][-1], eng.observables_f64()[3])
    if stop:
        assert np.array_equal(eng.limit_state(), ora.limit_state())
