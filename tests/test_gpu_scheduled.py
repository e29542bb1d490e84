"""cdpr_update_scheduled_kind: a command schedule resident in HBM, any command kind, any handle type, against the call
sequence it replaces (`for j: <callback of the kind>(batch j [, mask j]); refresh x update()`) bit for bit, and against the
fp64 oracle.  The reference's own publishers give the schedules: squarepositiontest (squarepositiontest.cpp:21-35),
squarevelocitytest with velocityEpsilon > 0 (squarevelocitytest.cpp:20-34 + the hold branch JFC.cpp:78-82), a setForce
schedule (JFC.h:92-95) with the tension distribution on.  Tolerances: tests/test_gpu_parity.py."""
import itertools

import numpy as np
import pytest

from test_gpu_parity import TOL, compare, perturbed_poses

pytestmark = pytest.mark.gpu

SETTER = {"velocity": "set_velocity_command", "position": "set_position_command", "force": "set_force_command"}


def state_of(e, f64=False):
    if f64:
        return e.observables_f64() + e.raw_state_f64()
    return e.platform_state() + e.joint_states() + e.raw_state()


def assert_same(a, b, what, f64=False):
    for x, y in zip(state_of(a, f64), state_of(b, f64)):
        assert np.array_equal(x, y), what


def run_both(pkg, oracle, cfg, pose, kind, sched, refresh, T, masks=None, record=True, f64=False, before=None, tol=TOL):
    """Engine `a` takes the schedule in one call, `b` the call sequence, the oracle runs beside them.  Returns (a, b, ora)."""
    B, n = cfg.batch, cfg.n_cables
    a, b = pkg.Engine(cfg, 0), pkg.Engine(cfg, 0)
    ora = oracle.OracleSim(cfg.to_struct(), oracle.DERIV_EXACT)
    for e in (a, b):
        e.set_platform_state_f64(pose7=pose) if f64 else e.set_platform_state(pose7=pose.astype(np.float32))
    ora.set_platform_state(pose7=pose.astype(np.float64) if f64 else pose.astype(np.float32).astype(np.float64))
    if before:
        for e in (a, b, ora):
            before(e)
    nb = (T + refresh - 1) // refresh
    d_sched = a.device_upload(sched)
    d_masks = a.device_upload(masks) if masks is not None else 0
    image = a.observable_image_bytes() if record else 0
    d_rec = a.device_alloc(image * T) if record else 0
    a.update_scheduled(T, refresh, d_sched, d_rec, image * T, kind=kind, d_robot_masks=d_masks)
    rec_b = []
    for j in range(nb):
        m = None if masks is None else masks[j]
        for e in (b, ora):
            getattr(e, SETTER[kind])(sched[j], m) if m is not None else getattr(e, SETTER[kind])(sched[j])
        k = min(refresh, T - j * refresh)
        if record:
            rec_b.append(b.update_record(k, min(k, 10))["effort"])
        else:
            b.update(k)
        ora.update(k)
    if record:
        raw = a.device_download(d_rec, (T, image), dtype=np.uint8)
        eff_a = np.array([a.decode_observables(raw[j])[2] for j in range(T)])
        eff_b = np.concatenate(rec_b)
        first = 1 if before is None else 0  # world step 0 is never published: its image is left as it was
        assert np.array_equal(eff_a[first:], eff_b[first:]), "every step's effort"
        a.device_free(d_rec)
    assert_same(a, b, "state after the schedule", f64)
    if f64:
        ga, oa = a.observables_f64(), ora.platform_state()
        assert np.abs(ga[3] - oa[0]).max() < 1e-9
    else:
        compare(a, ora, tol=tol, where=f"{kind} schedule")
    # the last batch stays latched: plain updates carry on with it
    a.update(7), b.update(7)
    assert_same(a, b, "updates after the schedule", f64)
    return a, b, ora


def square_schedule(pkg, kind, B, n, nb, rng, eps=0.0):
    """The reference's square publishers, one amplitude per robot (the publisher sends one value on every axis)."""
    gen = pkg.stimulus.square_position(n, amp=1.0, freq=0.7) if kind == "position" else pkg.stimulus.square_velocity(n, amp=1.0, freq=0.9)
    base = np.array(list(itertools.islice(gen, nb)), dtype=np.float64)  # [nb, n], values in {-1, 0, 1}
    amp = rng.uniform(0.002, 0.004, (1, B, 1)) if kind == "position" else rng.uniform(0.01, 0.04, (1, B, 1))
    return (base[:, None, :] * amp).astype(np.float32)


def test_squarepositiontest_at_config2_size(pkg, oracle):
    """4 096 x 4 cables, jointPositions every 100 steps for 700 steps in ONE launch (uniform handle, register-resident
    path), entered from Load (Position mode) and again after a velocity phase (the first batch then resets the position Pid)."""
    B, n, refresh, T = 4096, 4, 100, 700
    rng = np.random.default_rng(5)
    cfg = pkg.Config(batch=B)
    pose = perturbed_poses(cfg.model, B, rng, 0.01, 0.03)
    sched = square_schedule(pkg, "position", B, n, (T + refresh - 1) // refresh, rng)
    # (1 400 steps of +-4 mm position steps on a loop with gains 200 / 70 / 80: fp32 against fp64 over that length is bench.py's
    #  PARITY_TOL class, not the short-run TOL; a and b are compared bit for bit)
    long_tol = {"pose": 1e-4, "twist": 1e-3, "q": 1e-4, "qd": 1e-3, "eff": 5e-2}
    a, b, ora = run_both(pkg, oracle, cfg, pose, "position", sched, refresh, T, record=False, tol=long_tol)
    v = rng.uniform(-0.02, 0.02, (B, n)).astype(np.float32)
    for e in (a, b, ora):
        e.set_velocity_command(v)
        e.update(30)
    d_sched = a.device_upload(sched)
    a.update_scheduled(T, refresh, d_sched, kind="position")
    for j in range(sched.shape[0]):
        b.set_position_command(sched[j]), ora.set_position_command(sched[j])
        b.update(min(refresh, T - j * refresh)), ora.update(min(refresh, T - j * refresh))
    assert_same(a, b, "position schedule entered from Velocity mode")
    compare(a, ora, tol=long_tol, where="position schedule entered from Velocity mode")


def test_squarevelocitytest_with_the_hold_branch_at_config2_size(pkg, oracle):
    """4 096 x 4 cables, velocityEpsilon > 0: the square wave's zero phases put every cable into the position-hold branch
    (JFC.cpp:78-82: general controller path, both Pids alive), jointVelocities every 100 steps."""
    B, n, refresh, T = 4096, 4, 100, 600
    rng = np.random.default_rng(6)
    cfg = pkg.Config(batch=B, velocityEpsilon=0.001)
    pose = perturbed_poses(cfg.model, B, rng, 0.01, 0.03)
    sched = square_schedule(pkg, "velocity", B, n, (T + refresh - 1) // refresh, rng)
    assert (sched == 0).any() and (np.abs(sched) > 0.001).any()
    run_both(pkg, oracle, cfg, pose, "velocity", sched, refresh, T, record=False, tol={"pose": 1e-4, "twist": 1e-3, "q": 1e-4, "qd": 1e-3, "eff": 5e-2})


@pytest.mark.parametrize("mapping_env", ["1", "2", "3"])
def test_force_schedule_with_the_tension_distribution(pkg, oracle, monkeypatch, mapping_env):
    """setForce batches every 10 steps, 8 cables, FK + TD on: the distribution re-shapes every batch; one launch on the
    lane-per-robot and lane-pair mappings, a chain of launches with one lane per cable; every step's effort recorded."""
    monkeypatch.setenv("CDPR_MAPPING", mapping_env)
    B, n, refresh, T = 300, 8, 10, 137
    rng = np.random.default_rng(7)
    cfg = pkg.Config(model=pkg.eight_cable_model(), batch=B, stages=3)
    pose = perturbed_poses(cfg.model, B, rng, 0.02, 0.05)
    sched = (7.0 + rng.uniform(-1.5, 1.5, ((T + refresh - 1) // refresh, B, n))).astype(np.float32)
    run_both(pkg, oracle, cfg, pose, "force", sched, refresh, T)


@pytest.mark.parametrize("kind", ["velocity", "position", "force"])
@pytest.mark.parametrize("general", [False, True])
def test_per_robot_handles_take_a_mask_per_batch(pkg, oracle, kind, general):
    """per_robot_commands: batch j reaches the robots of mask j only (the others keep target, mode and Pid state); on the
    register-resident path and on the general controller path (hold branch live)."""
    B, n, refresh, T = 200, 8, 10, 95
    rng = np.random.default_rng(8 + general)
    cfg = pkg.Config(model=pkg.eight_cable_model(), batch=B, stages=3, perRobotCommands=True, velocityEpsilon=0.004 if general else -0.001)
    pose = perturbed_poses(cfg.model, B, rng, 0.02, 0.05)
    nb = (T + refresh - 1) // refresh
    if kind == "velocity":
        sched = rng.uniform(-0.03, 0.03, (nb, B, n)).astype(np.float32)
        if general:
            low = rng.random((nb, B, n)) < 0.3
            sched[low] = (rng.uniform(-1, 1, int(low.sum())) * 0.004).astype(np.float32)
    elif kind == "position":
        sched = rng.uniform(-0.004, 0.004, (nb, B, n)).astype(np.float32)
    else:
        sched = (7.0 + rng.uniform(-1.5, 1.5, (nb, B, n))).astype(np.float32)
    masks = (rng.random((nb, B)) < 0.6).astype(np.uint8)

    def before(e):  # a mixed starting point: some robots in Velocity mode, some in Force mode, the rest still in Position mode
        e.update(12)
        e.set_velocity_command(np.full((B, n), 0.02, np.float32), mask=(np.arange(B) % 3 == 0).astype(np.uint8))
        e.set_force_command(np.full((B, n), 7.0, np.float32), mask=(np.arange(B) % 3 == 1).astype(np.uint8))
        e.update(15)

    run_both(pkg, oracle, cfg, pose, kind, sched, refresh, T, masks=masks, before=before)


@pytest.mark.parametrize("variant", ["plain", "hold", "stop"])
def test_fp64_handles_take_schedules(pkg, oracle, variant):
    """... the HOLD (velocityEpsilon >= 0: some cables of every batch at or below epsilon) and TSTOP (joint stop) instantiations too."""
    B, n, refresh, T = 70, 8, 10, 64
    rng = np.random.default_rng(10)
    model = pkg.eight_cable_model()
    if variant == "stop":
        model.travel_lower, model.travel_upper, model.travel_stop = -0.004, 0.004, 3
    cfg = pkg.Config(model=model, batch=B, stages=3, precision=64, velocityEpsilon=0.004 if variant == "hold" else -0.001)
    pose = perturbed_poses(cfg.model, B, rng, 0.02, 0.05)
    sched = rng.uniform(-0.03, 0.03, ((T + refresh - 1) // refresh, B, n)).astype(np.float32)
    if variant == "hold":
        low = rng.random(sched.shape) < 0.3
        sched[low] = (rng.uniform(-1.0, 1.0, int(low.sum())) * 0.004).astype(np.float32)
    run_both(pkg, oracle, cfg, pose, "velocity", sched, refresh, T, record=False, f64=True)


@pytest.mark.parametrize("other", ["position", "force"])
def test_a_pending_command_of_another_kind_is_latched_as_in_the_call_sequence(pkg, oracle, other):
    """with a jointPositions / setForce command pending at the call, update() latches the velocity batch
    first and the other command after it - the handle runs batch 0's hold in THAT mode and enters Velocity mode (Pid reset)
    with batch 1.  The schedule's rows must never be read as position targets or forces."""
    B, n, refresh, T = 130, 8, 10, 45
    rng = np.random.default_rng(11)
    cfg = pkg.Config(model=pkg.eight_cable_model(), batch=B, stages=3)
    pose = perturbed_poses(cfg.model, B, rng, 0.02, 0.05)
    sched = rng.uniform(-0.03, 0.03, ((T + refresh - 1) // refresh, B, n)).astype(np.float32)
    cmd = rng.uniform(-0.004, 0.004, (B, n)).astype(np.float32) if other == "position" else (7.0 + rng.uniform(-1, 1, (B, n))).astype(np.float32)

    def before(e):
        e.update(20)
        getattr(e, SETTER[other])(cmd)  # pending when the schedule arrives

    a, b, ora = run_both(pkg, oracle, cfg, pose, "velocity", sched, refresh, T, before=before)


@pytest.mark.parametrize("mapping_env", ["1", "2"])
def test_a_schedule_of_one_step_is_published(pkg, oracle, monkeypatch, mapping_env):
    """nsteps = 1 used to pick a one-step kernel that ignores kFlagPublishAll: the step went unpublished
    and the record image stayed unwritten."""
    monkeypatch.setenv("CDPR_MAPPING", mapping_env)
    B, n = 130, 8
    rng = np.random.default_rng(12)
    cfg = pkg.Config(model=pkg.eight_cable_model(), batch=B, stages=3)
    pose = perturbed_poses(cfg.model, B, rng, 0.02, 0.05)
    sched = rng.uniform(-0.03, 0.03, (2, B, n)).astype(np.float32)

    def before(e):
        e.update(25)

    for T in (1, 11):  # 11: one step remains behind the refresh boundary
        run_both(pkg, oracle, cfg, pose, "velocity", sched[: (T + 9) // 10], 10, T, before=before)


@pytest.mark.parametrize("general", [False, True])
def test_mailbox_on_both_forms_and_a_mailbox_that_never_delivers(pkg, general):
    """d_ready: batch j is taken only once ready[j] != 0, inside the launch (uniform handle) and in front of every batch of
    the chain (general path); a word that never comes raises the handle's status word instead of hanging the GPU:
    cdpr_synchronize and the getters return CDPR_ERR_DEVICE until cdpr_reset."""
    B, T, refresh = 64, 40, 10
    rng = np.random.default_rng(13)
    cfg = pkg.Config(batch=B, velocityEpsilon=0.001 if general else -0.001)
    sched = rng.uniform(-0.03, 0.03, (T // refresh, B, 4)).astype(np.float32)
    out = []
    from conftest import mapped_words

    for use_mailbox in (False, True):
        eng = pkg.Engine(cfg, 0)
        eng.update(20)
        d_sched = eng.device_upload(sched)
        # the producer is the host: the words live in pinned host memory mapped to the device and are posted by plain stores (a copy
        # enqueued on another stream can share the waiting launch's hardware queue and would never complete: seen in round 6)
        ready, d_ready, free_ready = mapped_words(T // refresh)
        ready[0] = 1
        eng.update_scheduled(T, refresh, d_sched, d_ready=d_ready if use_mailbox else 0)
        if use_mailbox:
            for j in range(1, T // refresh):
                ready[j] = 1
        eng.synchronize()
        out.append(eng.platform_state() + eng.joint_states())
        eng.close()
        free_ready()
    for x, y in zip(*out):
        assert np.array_equal(x, y)
    eng = pkg.Engine(cfg, 0)
    eng.update(5)
    d_sched = eng.device_upload(sched)
    d_ready = eng.device_upload(np.array([1, 0, 1, 1], np.uint32))  # batch 1 never arrives
    eng.update_scheduled(T, refresh, d_sched, d_ready=d_ready)
    with pytest.raises(pkg.CdprError, match="mailbox timed out"):
        eng.synchronize()
    with pytest.raises(pkg.CdprError, match="mailbox timed out"):
        eng.joint_states()
    eng.reset()
    eng.update(3)
    eng.synchronize()
    assert np.isfinite(eng.platform_state()[0]).all()
    eng.close()


def test_bad_arguments(pkg):
    eng = pkg.Engine(pkg.Config(batch=8), 0)
    d = eng.device_upload(np.zeros((2, 8, 4), np.float32))
    m = eng.device_upload(np.ones((2, 8), np.uint8))
    with pytest.raises(pkg.CdprError, match="per_robot_commands"):
        eng.update_scheduled(20, 10, d, d_robot_masks=m)
    with pytest.raises(KeyError):
        eng.update_scheduled(20, 10, d, kind="torque")
    from cdpr_simulation_amd._native import lib
    import ctypes as C

    assert lib().cdpr_update_scheduled_kind(eng._h, 3, 20, 10, C.c_void_p(d), None, None, None, 0) == pkg._abi.ERR_INVALID
    assert lib().cdpr_update_scheduled_kind(eng._h, 0, 20, 0, C.c_void_p(d), None, None, None, 0) == pkg._abi.ERR_INVALID
    eng.close()


@pytest.mark.parametrize("n_cables,kind", [(4, "velocity"), (4, "position"), (8, "velocity"), (8, "position")])
def test_steady_state_stream_kernel_is_bit_identical(pkg, oracle, monkeypatch, n_cables, kind):
    """cdpr_pair_stream_kernel (round 6: the steady-state several-steps launch of lane-pair handles: ten step copies with the
    ring position as a compile-time constant, one observable descriptor per step, the next Joy batch fetched a period
    ahead) against (a) the general several-steps kernel it replaces (CDPR_PAIR_STREAM=0), (b) one launch per world step,
    bit for bit - every recorded step's observables, the state, the derivative windows as later steps see them - and
    against the fp64 oracle.  The launch starts at an arbitrary ring position (step 37), runs a schedule whose length is
    not a multiple of the refresh period or of the ring's ten, and is followed by plain updates."""
    monkeypatch.setenv("CDPR_MAPPING", "2")  # the lane-pair mapping whatever AUTO would take for this batch
    B, refresh, T = 301, 7, 83
    rng = np.random.default_rng(60 + n_cables)
    model = pkg.eight_cable_model() if n_cables == 8 else pkg.cube_model()
    cfg = pkg.Config(model=model, batch=B, stages=0)
    pose = perturbed_poses(cfg.model, B, rng).astype(np.float32)
    nb = (T + refresh - 1) // refresh
    amp = 0.03 if kind == "velocity" else 0.003
    sched = rng.uniform(-amp, amp, (nb, B, n_cables)).astype(np.float32)
    first = rng.uniform(-amp, amp, (B, n_cables)).astype(np.float32)

    def run(stream, per_step):
        monkeypatch.setenv("CDPR_PAIR_STREAM", "1" if stream else "0")
        e = pkg.Engine(cfg, 0)
        assert e.mapping == "lane-pair"
        e.set_platform_state(pose7=pose)
        getattr(e, SETTER[kind])(first)
        e.update(37)  # past the window fill, at ring position (37 + 8) % 10
        image = e.observable_image_bytes()
        d_rec = e.device_alloc(image * T)
        if per_step:
            recs = []
            for j in range(nb):
                getattr(e, SETTER[kind])(sched[j])
                k = min(refresh, T - j * refresh)
                for _ in range(k):
                    e.update(1)
                    recs.append(np.concatenate([x.ravel() for x in e.observables()]))
            rec = np.array(recs)
        else:
            d_s = e.device_upload(sched)
            e.update_scheduled(T, refresh, d_s, d_rec, image * T, kind=kind)
            raw = e.device_download(d_rec, (T, image), dtype=np.uint8)
            rec = np.array([np.concatenate([x.ravel() for x in e.decode_observables(raw[j])]) for j in range(T)])
        e.device_free(d_rec)
        e.update(13, 13)  # a fused launch from wherever the ring stands now (stream kernel again where it is on)
        e.update(3)
        e.synchronize()
        if not per_step:
            e.device_free(d_s)  # (only now: the schedule's last batch stays latched, the updates above still read it)
        return e, rec

    a, rec_a = run(True, False)
    b, rec_b = run(False, False)
    c, rec_c = run(False, True)
    assert np.array_equal(rec_a, rec_b) and np.array_equal(rec_a, rec_c), "every step's observables"
    assert_same(a, b, "stream kernel against the general several-steps kernel")
    assert_same(a, c, "stream kernel against one launch per world step")
    ora = oracle.OracleSim(cfg.to_struct(), oracle.DERIV_EXACT)
    ora.set_platform_state(pose7=pose.astype(np.float64))
    getattr(ora, SETTER[kind])(first)
    ora.update(37)
    for j in range(nb):
        getattr(ora, SETTER[kind])(sched[j])
        ora.update(min(refresh, T - j * refresh))
    ora.update(16)
    compare(a, ora, tol=TOL, where=f"stream kernel, {kind} schedule")
