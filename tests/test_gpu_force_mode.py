"""UpdateMode::Force through the boundary (JointForceCalculator::setForce, JFC.h:42,92-95; JFC.cpp:67-70): the HIP engine
against the fp64 oracle.  Force mode is the mode a JointForceCalculator is constructed in; `force = mForce`, no Pid runs,
setForce resets nothing, leaving Force through a Joy resets the Pid of the mode entered (JFC.cpp:99-119).
Tolerances: tests/test_gpu_parity.py."""
from dataclasses import replace

import numpy as np
import pytest

from test_gpu_parity import TOL, compare, pair, perturbed_poses

pytestmark = pytest.mark.gpu


def static_forces(cfg, B, rng, spread=0.5):
    """Forces around what holds the platform: 3.9657 N per cable on the shipped cube, mid tension on 8 cables."""
    base = 3.965671444 if cfg.n_cables == 4 else 7.0
    return (base + rng.uniform(-spread, spread, (B, cfg.n_cables))).astype(np.float32)


@pytest.mark.parametrize("mapping_env", ["1", "2", "3"])
@pytest.mark.parametrize("cables,stages", [(4, 0), (8, 0), (8, 3), (6, 1), (7, 2)])
def test_force_mode_and_switches_uniform_handle(pkg, oracle, monkeypatch, mapping_env, cables, stages):
    """Force from Load, Force -> Velocity -> Force -> Position -> Force on a uniform handle, one-step and fused launches, every
    kernel family (the three mappings x stage combinations select them), against the oracle after every segment."""
    monkeypatch.setenv("CDPR_MAPPING", mapping_env)
    B = 130
    rng = np.random.default_rng(40 + cables + stages)
    full = pkg.eight_cable_model()
    model = pkg.cube_model() if cables == 4 else (full if cables == 8 else pkg.Model(full.frame_anchors[:cables], full.platform_anchors[:cables]))
    cfg = pkg.Config(model=model, batch=B, stages=stages)
    eng, ora = pair(pkg, oracle, cfg, perturbed_poses(cfg.model, B, rng, 0.02, 0.05))
    f0 = static_forces(cfg, B, rng)
    assert eng.set_force_command(f0) == ora.set_force_command(f0) == 0
    eng.update(1), ora.update(1)  # world step 0: stepTime = 0, force 0 whatever the mode (JFC.cpp:61-66)
    assert np.all(eng.joint_states()[2] == 0.0)
    eng.update(40), ora.update(40)
    compare(eng, ora, where="force from load")
    if stages == 0:  # without tension distribution the applied force IS the command (inside the effort limit)
        assert np.array_equal(eng.joint_states()[2], f0)
    v = rng.uniform(-0.03, 0.03, (B, cables)).astype(np.float32)
    eng.set_velocity_command(v), ora.set_velocity_command(v)
    eng.update(1), ora.update(1)
    compare(eng, ora, where="first velocity step after force: the Pid was reset, returns 0")
    eng.update(36, 6), ora.update(36)
    compare(eng, ora, where="velocity after force (fused)")
    f1 = static_forces(cfg, B, rng)
    eng.set_force_command(f1), ora.set_force_command(f1)
    eng.update(30, 10), ora.update(30)
    compare(eng, ora, where="force again (fused)")
    p = rng.uniform(-0.004, 0.004, (B, cables)).astype(np.float32)
    eng.set_position_command(p), ora.set_position_command(p)
    eng.update(25), ora.update(25)
    compare(eng, ora, where="position after force")
    # a velocity Joy and a force command before the same update: the force command is latched last and wins; the velocity
    # Pid was reset on the way (setVelocityTarget from Position mode)
    eng.set_velocity_command(v), ora.set_velocity_command(v)
    eng.set_force_command(f0), ora.set_force_command(f0)
    eng.update(15), ora.update(15)
    compare(eng, ora, where="velocity + force in one step")
    eng.set_velocity_command(v), ora.set_velocity_command(v)
    eng.update(20), ora.update(20)
    compare(eng, ora, where="velocity at the end")


def test_force_command_count_rule_and_broadcast(pkg, oracle):
    """Same length rule as the Joy callbacks (PLG.cpp:68-73): n*B or n, anything else is dropped and changes nothing."""
    B = 5
    eng, ora = pair(pkg, oracle, pkg.Config(batch=B))
    eng.update(20), ora.update(20)
    assert eng.set_force_command(np.zeros(3, np.float32)) == 1  # CDPR_IGNORED
    assert eng.set_force_command(np.zeros(B * 4 + 1, np.float32)) == 1
    eng.update(5), ora.update(5)
    compare(eng, ora, where="dropped force commands change nothing")
    one = np.full(4, 4.0, np.float32)
    assert eng.set_force_command(one) == ora.set_force_command(one) == 0  # one row broadcast to every robot
    eng.update(30), ora.update(30)
    compare(eng, ora, where="broadcast force")
    assert np.array_equal(eng.joint_states()[2], np.tile(one, (B, 1)))


def test_force_command_from_device_buffers(pkg, oracle):
    """cdpr_set_force_command_device (copied) and cdpr_bind_force_command_device (read in place), with graph replays."""
    B = 64
    rng = np.random.default_rng(5)
    cfg = pkg.Config(batch=B)
    eng, ora = pair(pkg, oracle, cfg)
    fa, fb = static_forces(cfg, B, rng), static_forces(cfg, B, rng)
    da, db = eng.device_upload(fa), eng.device_upload(fb)
    eng.update(3), ora.update(3)
    eng.set_force_command_device(da, B * 4), ora.set_force_command(fa)
    eng.update(120), ora.update(120)  # long enough for captured chains of ten
    compare(eng, ora, where="copied device force command")
    eng.bind_force_command_device(db, B * 4), ora.set_force_command(fb)
    eng.update(57), ora.update(57)
    compare(eng, ora, where="bound device force command")
    eng.device_free(da), eng.device_free(db)


@pytest.mark.parametrize("stages", [0, 3])
def test_force_mode_per_robot_masks(pkg, oracle, stages):
    """Per-robot handles: a third of the robots driven open loop, a third in Velocity, a third in Position mode, changing
    membership over time (masked commands): mode and Pid state per robot, as independent plugin instances."""
    B = 200
    rng = np.random.default_rng(77)
    cfg = pkg.Config(model=pkg.eight_cable_model(), batch=B, stages=stages, perRobotCommands=True)
    eng, ora = pair(pkg, oracle, cfg, perturbed_poses(cfg.model, B, rng, 0.02, 0.05))
    eng.update(12), ora.update(12)
    grp = np.arange(B) % 3
    for rnd in range(6):
        f = static_forces(cfg, B, rng)
        v = rng.uniform(-0.03, 0.03, (B, 8)).astype(np.float32)
        p = rng.uniform(-0.003, 0.003, (B, 8)).astype(np.float32)
        g = (grp + rnd) % 3
        for sim in (eng, ora):
            sim.set_force_command(f, mask=g == 0)
            sim.set_velocity_command(v, mask=g == 1)
            if rnd % 2 == 0:
                sim.set_position_command(p, mask=g == 2)
        k = [1, 13, 10, 24, 7, 16][rnd]
        if rnd % 2:
            eng.update(k, 4)
        else:
            eng.update(k)
        ora.update(k)
        compare(eng, ora, where=f"round {rnd}")


def test_force_mode_on_the_general_controller_path(pkg, oracle):
    """Hold branch live (velocityEpsilon > 0): Force mode keeps mLastPosition at the joint position (JFC.cpp:68), so a
    velocity Joy with |target| <= epsilon after Force mode holds the position reached under the commanded forces."""
    B = 70
    rng = np.random.default_rng(9)
    cfg = pkg.Config(model=pkg.eight_cable_model(), batch=B, stages=3, velocityEpsilon=0.002)
    eng, ora = pair(pkg, oracle, cfg, perturbed_poses(cfg.model, B, rng, 0.02, 0.05))
    f = static_forces(cfg, B, rng, 1.0)
    eng.set_force_command(f), ora.set_force_command(f)
    eng.update(60), ora.update(60)
    compare(eng, ora, where="force on the general path")
    v = rng.uniform(-0.03, 0.03, (B, 8)).astype(np.float32)
    v[:, ::2] = 0.001  # below epsilon: these joints hold mLastPosition, which Force mode kept current
    eng.set_velocity_command(v), ora.set_velocity_command(v)
    eng.update(50), ora.update(50)
    compare(eng, ora, where="hold after force")
    eng.set_force_command(f), ora.set_force_command(f)
    eng.update(20), ora.update(20)
    compare(eng, ora, where="force again")


def test_force_mode_fp64_handle(pkg, oracle):
    """precision = 64: the same mode in the reference's own precision."""
    B = 9
    rng = np.random.default_rng(3)
    cfg = pkg.Config(model=pkg.eight_cable_model(), batch=B, stages=3, precision=64)
    eng, ora = pkg.Engine(cfg, 0), oracle.OracleSim(cfg.to_struct(), oracle.DERIV_EXACT)
    f = static_forces(cfg, B, rng)
    v = rng.uniform(-0.03, 0.03, (B, 8)).astype(np.float32)
    for sim in (eng, ora):
        sim.set_force_command(f)
        sim.update(50)
        sim.set_velocity_command(v)
        sim.update(40)
        sim.set_force_command(f)
        sim.update(30)
    q, qd, eff, pose, twist = eng.observables_f64()
    op, ot = ora.platform_state()
    oq, oqd, oe = ora.joint_states()
    assert np.abs(pose - op).max() < 1e-12 and np.abs(eff - oe).max() < 1e-9


def test_pid_debug_topic_keeps_stale_entries_in_force_mode(pkg, oracle):
    """No Pid runs in Force mode, so axes[0..3] of the `pid` topic keep what the last Pid call wrote; axes[4] follows the
    applied force (PLG.cpp:223-227)."""
    cfg = pkg.Config(batch=3, stages=pkg._abi.STAGE_PID_DEBUG)
    eng, ora = pair(pkg, oracle, cfg)
    v = np.full((3, 4), 0.02, np.float32)
    eng.set_velocity_command(v), ora.set_velocity_command(v)
    eng.update(30), ora.update(30)
    before = eng.pid_debug().copy()
    f = np.full((3, 4), 4.25, np.float32)
    eng.set_force_command(f), ora.set_force_command(f)
    eng.update(10), ora.update(10)
    after = eng.pid_debug()
    assert np.array_equal(after[:, :4], before[:, :4]) and np.all(after[:, 4] == 4.25)
    assert np.abs(after - ora.pid_debug()).max() < 2e-2


def test_facade_set_force(pkg, oracle):
    """The facade's setForce (a Joy of forces) against the oracle."""
    from cdpr_simulation_amd.messages import Joy

    plugin = pkg.CdprGazeboPlugin()
    plugin.Load(pkg.Config(batch=1))
    ora = oracle.OracleSim(pkg.Config(batch=1).to_struct(), oracle.DERIV_EXACT)
    f = np.full(4, 3.9, np.float32)
    plugin.setForce(Joy(axes=f))
    ora.set_force_command(f)
    plugin.update(80), ora.update(80)
    compare(plugin.engine, ora, where="facade setForce")


@pytest.mark.parametrize("per_robot", [False, True])
def test_chunked_launches_are_bit_identical_to_one_launch(pkg, monkeypatch, per_robot):
    """A step issued as back-to-back launches over blocks of robots (cdpr_create does that between one and ~6 robots per
    hardware lane; CDPR_CHUNK forces it here at a small size) gives the same bits as the single launch: one-step, fused
    and recorded updates, ragged last block."""
    B = 9000 + 37
    rng = np.random.default_rng(21)
    cfg = pkg.Config(model=pkg.eight_cable_model(), batch=B, stages=3, perRobotCommands=per_robot)
    pose = perturbed_poses(cfg.model, B, rng).astype(np.float32)
    v = rng.uniform(-0.03, 0.03, (B, 8)).astype(np.float32)
    p = rng.uniform(-0.003, 0.003, (B, 8)).astype(np.float32)
    monkeypatch.setenv("CDPR_MAPPING", "1")
    out = []
    for chunk in ("0", "2048"):
        monkeypatch.setenv("CDPR_CHUNK", chunk)
        eng = pkg.Engine(cfg, 0)
        eng.set_platform_state(pose7=pose)
        eng.update(7)
        if per_robot:
            eng.set_velocity_command(v, mask=np.arange(B) % 2 == 0)
        else:
            eng.set_velocity_command(v)
        eng.update(23)
        eng.update(20, 5)
        rec = eng.update_record(12, 4)
        eng.set_position_command(p)
        eng.update(9)
        out.append((eng.platform_state(), eng.joint_states(), rec, eng.fk_state(), eng.td_state()))
        eng.close()
    a, b = out
    for x, y in zip(a[0] + a[1] + a[3] + a[4], b[0] + b[1] + b[3] + b[4]):
        assert np.array_equal(x, y)
    for key in a[2]:
        assert np.array_equal(a[2][key], b[2][key]), key


@pytest.mark.parametrize("cables,stages", [(8, 3), (8, 1), (4, 0), (7, 3)])
def test_persistent_one_wave_kernel_is_bit_identical(pkg, monkeypatch, cables, stages):
    """CDPR_PERSIST=1: one-step launches on the persistent one-wave kernel (a fixed grid of waves walks over the blocks of 64
    robots, the next block's rows requested while the current block computes - platform rows by untracked loads into
    accumulation registers behind an explicit vmcnt, controller rows by LDS-DMA into the staging buffer the PID stage has
    just read).  Forced onto 3 waves here so that every wave takes several blocks, ragged last block, Velocity, Position
    and Force mode, the first world step: same bits as the default kernels."""
    B = 64 * 13 + 5
    rng = np.random.default_rng(77)
    full = pkg.eight_cable_model()
    model = pkg.cube_model() if cables == 4 else replace(full, frame_anchors=full.frame_anchors[:cables], platform_anchors=full.platform_anchors[:cables])
    cfg = pkg.Config(model=model, batch=B, stages=stages)
    pose = perturbed_poses(model, B, rng).astype(np.float32)
    v = rng.uniform(-0.03, 0.03, (B, cables)).astype(np.float32)
    p = rng.uniform(-0.003, 0.003, (B, cables)).astype(np.float32)
    f = rng.uniform(5.0, 25.0, (B, cables)).astype(np.float32)
    monkeypatch.setenv("CDPR_MAPPING", "1")
    monkeypatch.setenv("CDPR_NO_GRAPH", "1")
    out = []
    for persist in ("0", "1"):
        monkeypatch.setenv("CDPR_PERSIST", persist)
        monkeypatch.setenv("CDPR_PERSIST_GRID", "3")
        eng = pkg.Engine(cfg, 0)
        eng.set_platform_state(pose7=pose)
        eng.update(1)  # the first world step: no Pid call
        eng.set_velocity_command(v)
        for _ in range(17):
            eng.update(1)
        eng.set_position_command(p)
        for _ in range(13):
            eng.update(1)
        eng.set_force_command(f)
        for _ in range(5):
            eng.update(1)
        eng.set_velocity_command(-v)
        for _ in range(12):
            eng.update(1)
        out.append(eng.platform_state() + eng.joint_states() + (eng.fk_state() if stages & 1 else ()) + (eng.td_state() if stages & 2 else ()))
        eng.close()
    a, b = out
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("cables,stages,mapping_env", [(4, 0, "2"), (4, 0, "1"), (8, 3, "1"), (8, 0, "2")])
def test_scheduled_update_equals_the_call_sequence(pkg, oracle, monkeypatch, cables, stages, mapping_env):
    """cdpr_update_scheduled (a whole jointVelocities schedule in ONE launch, Joy batch j latched at step 10 j, every step's
    observables recorded) against what it replaces - set_velocity_command_device + update_record per batch - bit for bit,
    from Load (world step 0, Position mode: the first Joy resets the velocity Pid) and again in steady state; 137 steps
    (ragged last hold); the last batch stays latched; and against the oracle."""
    monkeypatch.setenv("CDPR_MAPPING", mapping_env)
    B, T, refresh = 300, 137, 10
    rng = np.random.default_rng(90 + cables)
    model = pkg.cube_model() if cables == 4 else pkg.eight_cable_model()
    cfg = pkg.Config(model=model, batch=B, stages=stages)
    pose = perturbed_poses(model, B, rng, 0.02, 0.05).astype(np.float32)
    nb = (T + refresh - 1) // refresh
    a, b = pkg.Engine(cfg, 0), pkg.Engine(cfg, 0)
    ora = oracle.OracleSim(cfg.to_struct(), oracle.DERIV_EXACT)
    ora.set_platform_state(pose7=pose.astype(np.float64))
    for e in (a, b):
        e.set_platform_state(pose7=pose)
    image = a.observable_image_bytes()
    for rnd in range(2):
        sched = rng.uniform(-0.03, 0.03, (nb, B, cables)).astype(np.float32)
        d_sched = a.device_upload(sched)
        d_rec = a.device_alloc(image * T)
        a.update_scheduled(T, refresh, d_sched, d_rec, image * T)
        raw = a.device_download(d_rec, (T, image), dtype=np.uint8)
        eff_a = np.array([a.decode_observables(raw[j])[2] for j in range(T)])
        eff_b = []
        for j in range(nb):
            b.set_velocity_command(sched[j]), ora.set_velocity_command(sched[j])
            k = min(refresh, T - j * refresh)
            rec = b.update_record(k, k)
            eff_b.append(rec["effort"])
            ora.update(k)
        eff_b = np.concatenate(eff_b)
        first = 1 if rnd == 0 else 0  # world step 0 is never published: its image is left as it was
        assert np.array_equal(eff_a[first:], eff_b[first:]), f"round {rnd}"
        for x, y in zip(a.platform_state() + a.joint_states(), b.platform_state() + b.joint_states()):
            assert np.array_equal(x, y)
        compare(a, ora, where=f"scheduled update, round {rnd}")
        # the last batch stays latched: plain updates carry on with it
        a.update(7), b.update(7), ora.update(7)
        for x, y in zip(a.platform_state() + a.joint_states(), b.platform_state() + b.joint_states()):
            assert np.array_equal(x, y)
        a.device_free(d_rec)  # (the schedule stays: its last batch is the latched command)


def test_scheduled_update_waits_for_its_mailbox(pkg):
    """With a mailbox, batch j is only taken once ready[j] != 0: a second handle (its own stream) releases the batches one
    by one while the launch spins; the result equals the launch without a mailbox."""
    B, T, refresh = 64, 60, 10
    rng = np.random.default_rng(4)
    cfg = pkg.Config(batch=B)
    sched = rng.uniform(-0.03, 0.03, (T // refresh, B, 4)).astype(np.float32)
    out = []
    from conftest import mapped_words

    for use_mailbox in (False, True):
        eng = pkg.Engine(cfg, 0)
        eng.update(20)
        d_sched = eng.device_upload(sched)
        ready, d_ready, free_ready = mapped_words(T // refresh)  # pinned host memory mapped to the device: the host posts by plain stores
        ready[0] = 1
        eng.update_scheduled(T, refresh, d_sched, d_ready=d_ready if use_mailbox else 0)
        if use_mailbox:
            for j in range(1, T // refresh):  # release batch by batch while the launch polls
                ready[j] = 1
        eng.synchronize()
        out.append(eng.platform_state() + eng.joint_states())
        eng.close()
        free_ready()
    for x, y in zip(*out):
        assert np.array_equal(x, y)
