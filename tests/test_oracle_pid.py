"""Known-answer and property tests of the oracle's controller stack (Pid.cpp, JointForceCalculator.cpp,
the update() ordering of CdprGazeboPlugin.cpp).  The reference ships no tests for these; the numeric KATs are
the survey-time spot values (tests/golden/pid_kat.json) — see oracle/cdpr_oracle.h for the parity status."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")
KAT = json.load(open(os.path.join(GOLD, "pid_kat.json")))


def sine_cmd(k):
    """sinevelocitytest.cpp: time accumulates 1/100 per publish; float32 on the wire."""
    t = 0.0
    for _ in range(k):
        t += 1.0 / 100.0
    return float(np.float32(0.05 * np.sin(t * 0.1 * 2 * np.pi)))


def toy_plant_trace(pid, nsteps, cmd_of_step):
    dt, q, qd, out = 1e-3, 0.0, 0.0, []
    for k in range(nsteps):
        f = pid.update(cmd_of_step(k), qd, k * dt)
        out.append(f)
        qd += dt * (f - qd)
        q += dt * qd
    return out


@pytest.mark.parametrize("mode", [0, 1])
def test_velocity_pid_kat(pkg, oracle, mode):
    s = pkg.Config().to_struct()
    pid = oracle.OraclePid(s.velocity_pid, mode)
    assert abs(sine_cmd(1) - KAT["velocity_pid_toy_plant"]["first_command_sample_k1"]) < 1e-12
    out = toy_plant_trace(pid, 14, lambda k: sine_cmd(k // 10))
    assert all(v == 0.0 for v in out[: KAT["velocity_pid_toy_plant"]["force_zero_through_k"] + 1])
    for k, v in KAT["velocity_pid_toy_plant"]["force"].items():
        assert abs(out[int(k)] - v) < 1e-13 * max(1.0, abs(v)) + 2e-14, (k, out[int(k)], v)


def test_mode_switch_kat(pkg, oracle):
    """Velocity Pid freshly reset at k = 5 (what setVelocityTarget does on a mode change, JFC.cpp:113-115)."""
    s = pkg.Config().to_struct()
    pid = oracle.OraclePid(s.velocity_pid, 1)
    dt, qd, out = 1e-3, 0.0, {}
    target = float(np.float32(0.01))
    for k in range(5, 8):
        f = pid.update(target, qd, k * dt)
        out[str(k)] = f
        qd += dt * (f - qd)
    for k, v in KAT["mode_switch"]["force"].items():
        assert abs(out[k] - v) < 1e-10, (k, out[k], v)


def test_derivative_weights_closed_form(pkg):
    w = pkg.derivative_weights(11, 2)
    assert np.allclose(w, KAT["derivative_weights_n11_d2"], atol=5e-13)
    assert abs(w.sum()) < 1e-15  # a constant has zero derivative
    assert abs((w * np.arange(11)).sum() - 1.0) < 1e-14  # a unit ramp has derivative 1 per sample


@pytest.mark.parametrize("n,d", [(11, 2), (5, 1), (7, 3), (16, 4), (2, 1)])
def test_derive_equals_fir_on_uniform_grid(pkg, oracle, n, d):
    """Pid::derive + fitPolynomial on a uniform grid == the FIR the GPU kernel uses."""
    s = pkg.Config().to_struct()
    p = s.velocity_pid
    p.d_buffer_length, p.d_degree = n, d
    p.p_gain, p.i_gain, p.d_gain, p.cmd_limit, p.i_limit = 0.0, 0.0, 1.0, 1e9, 1e9
    pid = oracle.OraclePid(p, 1)
    rng = np.random.default_rng(n * 10 + d)
    e = rng.standard_normal(3 * n + 5)
    dt, w = 1e-3, pkg.derivative_weights(n, d)
    pid.update(0.0, 0.0, 0.0)  # first call after reset: 0, no sample taken
    for k in range(len(e)):
        out = pid.update(e[k], 0.0, (k + 1) * dt)  # error = desired - actual = e[k]; output = D term
        if k + 1 < n:
            assert out == 0.0
        else:
            ref = (w * e[k + 1 - n : k + 1]).sum() / dt
            assert abs(out - ref) < 1e-7 * max(1.0, abs(ref)), (k, out, ref)


def test_faithful_fit_agrees_with_exact_for_early_times_and_drifts_later(pkg, oracle):
    """The reference fits in ABSOLUTE sim time (Pid.cpp:224-244): accurate near t = 0, noise by t ~ 30 s."""
    s = pkg.Config().to_struct()
    p = s.velocity_pid
    p.p_gain, p.i_gain, p.d_gain, p.cmd_limit, p.i_limit = 0.0, 0.0, 1.0, 1e9, 1e9
    dt = 1e-3

    def max_rel_dev(t0):
        a, b = oracle.OraclePid(p, oracle.DERIV_FAITHFUL), oracle.OraclePid(p, oracle.DERIV_EXACT)
        dev = 0.0
        for k in range(40):
            t = t0 + k * dt
            e = np.sin(5.0 * t)
            fa, fb = a.update(e, 0.0, t), b.update(e, 0.0, t)
            if k > 12:
                dev = max(dev, abs(fa - fb) / 5.0)
        return dev

    assert max_rel_dev(0.0) < 1e-9
    assert max_rel_dev(1.0) < 1e-6
    assert max_rel_dev(100.0) > 1e-4  # the reference's own output is numerical noise out here


def test_first_call_after_reset_returns_zero_and_takes_no_sample(pkg, oracle):
    s = pkg.Config().to_struct()
    pid = oracle.OraclePid(s.position_pid, 1)
    assert pid.update(0.1, 0.0, 0.0) == 0.0  # Pid.cpp:123-126
    f = pid.update(0.1, 0.0, 1e-3)
    assert abs(f - (200.0 * 0.1 + 70.0 * 1e-3 * 0.1)) < 1e-12  # P + I, D window not yet full
    pid.reset()
    assert pid.update(0.1, 0.0, 2e-3) == 0.0


def test_command_clamp_and_anti_windup_overshoot(pkg, oracle):
    """Pid.cpp:175-186: clamp to +-cmdLimit, then the anti-windup fix-up adds dt*e*Ki on top of the clamp
    and rolls the integrator back."""
    s = pkg.Config().to_struct()
    pid = oracle.OraclePid(s.velocity_pid, 1)
    pid.update(10.0, 0.0, 0.0)
    f = pid.update(10.0, 0.0, 1e-3)  # P alone = 2000 -> clamped to 100
    assert abs(f - (100.0 + 1e-3 * 10.0 * 20.0)) < 1e-12
    f2 = pid.update(10.0, 0.0, 2e-3)
    assert abs(f2 - f) < 1e-12  # integrator was rolled back, nothing accumulates while saturated


def test_integral_clamp_back_calculation(pkg, oracle):
    s = pkg.Config().to_struct()
    p = s.velocity_pid
    p.p_gain, p.d_gain, p.i_gain, p.i_limit, p.cmd_limit = 0.0, 0.0, 20.0, 1.0, 1e9
    pid = oracle.OraclePid(p, 1)
    pid.update(100.0, 0.0, 0.0)
    for k in range(1, 6):
        f = pid.update(100.0, 0.0, k * 1e-3)
    assert abs(f - 1.0) < 1e-15  # Pid.cpp:143-146: I clamped at iLimit, Ierr back-computed
    f = pid.update(-100.0, 0.0, 6e-3)
    assert abs(f - (1.0 - 20.0 * 1e-3 * 100.0)) < 1e-12  # leaves the clamp at once (Ierr was I/Ki, not the raw sum)


def test_qr_solver_matches_numpy(oracle):
    rng = np.random.default_rng(0)
    for n in (2, 3, 5):
        a = rng.standard_normal((n, n))
        b = rng.standard_normal(n)
        assert np.allclose(oracle.qr_solve(a, b), np.linalg.solve(a, b), rtol=1e-10, atol=1e-12)


def test_qr_solver_truncates_by_eigens_nonzero_pivots_rule(oracle):
    """Eigen 3.3 ColPivHouseholderQR (Pid.cpp:246): a pivot is dropped when the largest remaining SQUARED column norm is below
    (eps * max column norm)^2 * (rows - k) / rows - not by |R_kk| <= eps * n * max|R_kk| (rank()'s default threshold, what the
    oracle used through round 5).  diag(1, 2e-16) separates the two rules: 4e-32 >= 2.46e-32 keeps the pivot."""
    x = oracle.qr_solve(np.diag([1.0, 2e-16]), np.array([3.0, 4e-16]))
    assert np.allclose(x, [3.0, 2.0], rtol=1e-15)
    x = oracle.qr_solve(np.diag([1.0, 1e-16]), np.array([3.0, 4e-16]))  # 1e-32 < 2.46e-32: dropped, that unknown is 0
    assert x[0] == 3.0 and x[1] == 0.0
    # an exactly dependent column: two pivots, the third unknown (the permuted-last column) set to 0, and A x = b still holds
    a = np.array([[2.0, 1.0, 3.0], [0.0, 1.0, 1.0], [1.0, 0.0, 1.0]])  # col 2 = col 0 + col 1
    b = a @ np.array([1.0, -2.0, 0.5])
    x = oracle.qr_solve(a, b)
    assert np.count_nonzero(x == 0.0) == 1 and np.allclose(a @ x, b, atol=1e-14)
    # (an all-zero matrix: threshold_helper = 0 and `0 < 0` is false, so the published rule keeps every pivot and divides by 0;
    #  Pid::fitPolynomial never builds one - its (0, 0) entry is the sample count)


def test_qr_solver_on_the_reference_fit_matrices(oracle):
    """The matrices Pid::fitPolynomial really hands over (Pid.cpp:224-244: power sums of absolute time): while they are
    well conditioned the solve agrees with an SVD least-squares solve; far out (t = 100 s, cond ~ 1e21) it must still return
    finite numbers with every pivot kept or a trailing unknown zeroed, never NaN."""
    for t0, tol in ((0.0, 1e-9), (0.5, 1e-7)):
        t = t0 + 1e-3 * np.arange(11)
        y = np.sin(5.0 * t)
        a = np.array([[np.sum(t ** (i + j)) for j in range(3)] for i in range(3)])
        b = np.array([np.sum(t ** i * y) for i in range(3)])
        x = oracle.qr_solve(a, b)
        ref = np.polyfit(t - t.mean(), y, 2)[::-1]  # centred fit, then shifted back to absolute-time coefficients
        m = t.mean()
        ref_abs = np.array([ref[0] - ref[1] * m + ref[2] * m * m, ref[1] - 2 * ref[2] * m, ref[2]])
        assert np.allclose(x[1] + 2 * x[2] * t[-1], ref_abs[1] + 2 * ref_abs[2] * t[-1], rtol=tol, atol=tol)
    t = 100.0 + 1e-3 * np.arange(11)
    a = np.array([[np.sum(t ** (i + j)) for j in range(3)] for i in range(3)])
    assert np.isfinite(oracle.qr_solve(a, np.array([np.sum(t ** i) for i in range(3)]))).all()


def test_fir_weights_and_the_three_derivative_modes(pkg, oracle):
    """ORC_DERIV_FIR (BASELINE.md section 3's `fir` baseline mode): the oracle's own end-point weights equal SURVEY.md 8(a)
    row 5's published values and the product's host-side table (cdpr_derivative_weights); FAITHFUL, EXACT and FIR agree
    while the reference's fit is accurate (t <= 2 s); a window with a gap in it makes FIR fall back to the EXACT fit."""
    import ctypes as C

    from cdpr_simulation_amd._native import lib as hip_lib

    survey = [0.129370629371, 0.0335664335664, -0.0389277389277, -0.0881118881119, -0.113986013986, -0.11655011655,
              -0.0958041958042, -0.0517482517483, 0.0156177156177, 0.106293706294, 0.22027972028]
    w = np.zeros(32)
    oracle.lib().orc_fir_weights.argtypes = [C.c_uint, C.c_uint, C.POINTER(C.c_double)]
    oracle.lib().orc_fir_weights(11, 2, w.ctypes.data_as(C.POINTER(C.c_double)))
    assert np.allclose(w[:11], survey, rtol=0, atol=5e-13) and abs(w[:11].sum()) < 1e-15
    for nbuf, deg in ((11, 2), (5, 1), (32, 3), (7, 2)):
        w = np.zeros(32)
        oracle.lib().orc_fir_weights(nbuf, deg, w.ctypes.data_as(C.POINTER(C.c_double)))
        wp = np.zeros(32)
        assert hip_lib().cdpr_derivative_weights(nbuf, deg, wp.ctypes.data_as(C.POINTER(C.c_double))) == 0
        assert np.allclose(w[:nbuf], wp[:nbuf], rtol=0, atol=1e-13), (nbuf, deg)
        x = np.arange(nbuf, dtype=np.float64)
        assert np.allclose(w[:nbuf], np.linalg.pinv(np.vander(x - x.mean(), deg + 1, increasing=True))[1] + 0, atol=1e-12) or deg >= 2
    s = pkg.Config().to_struct()
    p = s.velocity_pid
    p.p_gain, p.i_gain, p.d_gain, p.cmd_limit, p.i_limit = 0.0, 0.0, 1.0, 1e9, 1e9
    dt = 1e-3
    pids = [oracle.OraclePid(p, m) for m in (oracle.DERIV_FAITHFUL, oracle.DERIV_EXACT, oracle.DERIV_FIR)]
    for k in range(2000):
        t = k * dt
        f = [q.update(np.sin(5.0 * t), 0.0, t) for q in pids]
        assert abs(f[2] - f[1]) < 1e-9 and abs(f[0] - f[1]) < 2e-5 * (1 + t * t * 30), (k, f)
    # a gap: skip 7 steps, then the window is not uniform for 10 calls -> FIR mode must give EXACT's answer bit for bit
    for k in range(2007, 2030):
        t = k * dt
        f = [q.update(np.sin(5.0 * t), 0.0, t) for q in pids[1:]]
        if k < 2017:
            assert f[0] == f[1], k
        else:
            assert abs(f[0] - f[1]) < 1e-9


def test_plugin_ordering_first_steps(pkg, oracle):
    """update() at t = 0 sees stepTime = 0 -> force 0 and no Pid call (JFC.cpp:61-66); the next call is the Pid's
    'first' (-> 0); the position Pid (target 0 after Load, PLG.cpp:153-157) acts from the third step."""
    cfg = pkg.Config(batch=1)
    sim = oracle.OracleSim(cfg.to_struct())
    effs = []
    for k in range(4):
        sim.update(1)
        effs.append(sim.joint_states()[2][0].copy())
    assert np.all(effs[1] == 0.0)  # step 1 published (step 0 is not: now - prev = 0 is not > publishPeriod)
    assert np.all(effs[2] != 0.0) and np.all(effs[3] != 0.0)
    # free fall during the two force-free steps: q = L0 - L < 0 grows, position Pid pulls back (positive force)
    assert np.all(effs[2] > 0.0)


def test_wrong_length_commands_are_ignored(pkg, oracle):
    cfg = pkg.Config(batch=2)
    sim, ref = oracle.OracleSim(cfg.to_struct()), oracle.OracleSim(cfg.to_struct())
    assert sim.set_velocity_command(np.zeros(3, dtype=np.float32)) == pkg._abi.IGNORED  # PLG.cpp:68-73
    assert sim.set_position_command(np.zeros(9, dtype=np.float32)) == pkg._abi.IGNORED
    sim.update(20), ref.update(20)
    assert np.array_equal(sim.raw_state()[0], ref.raw_state()[0])


def test_publish_period_decimates_observables(pkg, oracle):
    cfg = pkg.Config(batch=1, publishPeriod=0.0045)
    sim = oracle.OracleSim(cfg.to_struct())
    seen = []
    for k in range(12):
        sim.update(1)
        seen.append(sim.platform_state()[0][0, 2])
    changes = [k for k in range(1, 12) if seen[k] != seen[k - 1]]
    assert changes == [5, 10]  # published when now - prev > 0.0045: t = 0.005, 0.010


def test_masked_commands_equal_independent_plugin_instances(pkg, oracle):
    """Per-robot arrival: a batch in which only some robots receive a Joy before an update() must equal B separate
    one-robot simulators (= B instances of CdprGazeboPlugin, PLG.cpp:206-219 runs per model), each of which got only
    its own messages: unaddressed robots keep target, mode and Pid state."""
    B, n = 5, 4
    cfg = pkg.Config(batch=B)
    rng = np.random.default_rng(77)
    pose = np.tile(cfg.model.home_pose(), (B, 1))
    pose[:, :3] += rng.uniform(-0.02, 0.02, (B, 3))
    whole = oracle.OracleSim(cfg.to_struct())
    solo = [oracle.OracleSim(pkg.Config(batch=1).to_struct()) for _ in range(B)]
    whole.set_platform_state(pose7=pose)
    for b, s_ in enumerate(solo):
        s_.set_platform_state(pose7=pose[b:b + 1])
    script = [("run", 12), ("vel", [1, 0, 1, 0, 0]), ("run", 20), ("pos", [0, 1, 1, 0, 0]), ("run", 15), ("vel", [1, 1, 0, 0, 1]),
              ("pos", [0, 0, 0, 0, 1]), ("run", 25)]  # robot 3 never hears anything; robot 4 gets both kinds in one update
    for kind, arg in script:
        if kind == "run":
            whole.update(arg)
            for s_ in solo:
                s_.update(arg)
            continue
        axes = rng.uniform(-0.03, 0.03, (B, n)).astype(np.float32) * (1.0 if kind == "vel" else 0.1)
        mask = np.array(arg, dtype=np.uint8)
        assert (whole.set_velocity_command if kind == "vel" else whole.set_position_command)(axes, mask) == 0
        for b, s_ in enumerate(solo):
            if mask[b]:
                (s_.set_velocity_command if kind == "vel" else s_.set_position_command)(axes[b])
    wp, wt = whole.raw_state()
    wq, wqd, we = whole.joint_states()
    for b, s_ in enumerate(solo):
        sp, st_ = s_.raw_state()
        sq, sqd, se = s_.joint_states()
        assert np.array_equal(wp[b], sp[0]) and np.array_equal(wt[b], st_[0]) and np.array_equal(we[b], se[0])
    assert whole.set_velocity_command(np.zeros((B, 3), np.float32), np.ones(B, np.uint8)) == 1  # wrong length: still dropped
