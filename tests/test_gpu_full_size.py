"""Every BASELINE config at its REAL shape on the GPU (VERDICT r01 "configs_untested"), through the C-ABI:

  config 2   4 096 x 4-cable, IK + PID + dynamics, both wavefront mappings
  config 3   65 536 x 8-cable: tests/test_gpu_parity.py::test_full_size_properties_config3
  config 4   524 288 x 8-cable over 8 GPUs: the sharded engine against the oracle on however many devices the box has
             (one device twice on a 1-GPU lease), the auto-selected low-register kernel at 131 072 robots, and
             `bench.py --gpus 2` as its own launcher
  config 5   512 x 128 x 64 MPC rollout (one GPU's share)
  configs 4 and 5 AS SPECIFIED (524 288 robots in eight shards of 65 536; 4 096 x 128 x 64 in eight shards of 512) through
             ShardedEngine, on however many devices the box has (eight shards on the one device of a 1-GPU lease)

At these sizes the oracle only checks a slice (robots are independent, so a slice of the batch is its own problem);
the rest of the batch is covered by size-independent properties: duplicates stay bit-identical, a permutation of the
batch permutes the result, unit quaternions, identical samples give identical costs.
Tolerances as in tests/test_gpu_parity.py (fp32 kernel vs fp64 oracle, absolute).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from test_gpu_parity import TOL, check_slice, perturbed_poses

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("mapping_id", [1, 2])
def test_config2_full_size(pkg, oracle, monkeypatch, mapping_id):
    """BASELINE config 2 as SURVEY 8(d) states it: 4 096 x 4-cable, IK + PID + dynamics, seeds rng(1234), per-robot sine
    commands refreshed every 10 steps."""
    import bench

    monkeypatch.setenv("CDPR_MAPPING", str(mapping_id))
    B, steps = 4096, 300
    model, pose, command, n_cmd = bench.make_workload(pkg, B, 4, 1234, steps)
    pose[B // 2:] = pose[: B // 2]  # second half duplicates the first
    dup = lambda c: np.concatenate([c[: B // 2], c[: B // 2]])  # noqa: E731
    cfg = pkg.Config(model=model, batch=B)
    eng = pkg.Engine(cfg, 0)
    assert eng.mapping == ("lane-per-robot" if mapping_id == 1 else "lane-pair")
    eng.set_platform_state(pose7=pose)
    script = []
    for j in range(n_cmd):
        c = dup(command(j))
        eng.set_velocity_command(c)
        eng.update(10)
        script.append((10, c))
    p, t = eng.platform_state()
    q, qd, eff = eng.joint_states()
    assert np.isfinite(p).all() and np.isfinite(eff).all()
    assert np.abs(np.linalg.norm(p[:, 3:], axis=1) - 1.0).max() < 1e-6
    assert np.array_equal(p[: B // 2], p[B // 2:]) and np.array_equal(eff[: B // 2], eff[B // 2:])
    check_slice(pkg, oracle, dict(model=model), slice(1000, 1256), pose, script, (p, t, q, qd, eff))
    # permutation equivariance: the same robots in another order, in a smaller batch
    perm = np.random.default_rng(0).permutation(1024)
    e2 = pkg.Engine(pkg.Config(model=model, batch=1024), 0)
    e2.set_platform_state(pose7=pose[:1024][perm])
    for nsteps, c in script:
        e2.set_velocity_command(c[:1024][perm])
        e2.update(nsteps)
    assert np.array_equal(e2.platform_state()[0], p[:1024][perm])
    assert np.array_equal(e2.joint_states()[2], eff[:1024][perm])


def test_config5_full_size_rollout(pkg, oracle, monkeypatch):
    """One GPU's share of BASELINE config 5: 512 robots x 128 sampled sequences x 64-step horizon, every stage on.
    Samples 64..127 repeat samples 0..63 => identical costs; robots 100..107 against the oracle."""
    import bench

    monkeypatch.setenv("CDPR_MAPPING", "1")
    B, S, H = bench.ROLLOUT_SHAPE
    rng = np.random.default_rng(1236)
    cfg_kwargs = dict(model=pkg.eight_cable_model(), stages=3)
    pose = perturbed_poses(cfg_kwargs["model"], B, rng).astype(np.float32)
    eng = pkg.Engine(pkg.Config(batch=B, **cfg_kwargs), 0)
    eng.set_platform_state(pose7=pose)
    eng.update(20)
    cmds = bench.make_rollout_commands(B, H, S, 8)
    cmds[:, :, S // 2:, :] = cmds[:, :, : S // 2, :]
    ref = pose[:, :3].astype(np.float32) + np.float32([0.0, 0.0, 0.01])
    cost = eng.rollout_velocity(cmds, ref)
    assert cost.shape == (B, S) and np.isfinite(cost).all() and (cost > 0).all()
    assert np.array_equal(cost[:, : S // 2], cost[:, S // 2:])
    # launch / fetch and the device-resident form give the same bits as the synchronous call
    dptr = eng.device_upload(cmds)
    eng.rollout_launch((dptr, S, H), ref)
    assert np.array_equal(eng.rollout_fetch(), cost)
    d_ref, d_cost = eng.device_upload(ref), eng.device_alloc(B * S * 4)
    eng.rollout_velocity_device(dptr, S, H, d_ref, d_cost)
    assert np.array_equal(eng.device_download(d_cost, (B, S)), cost)
    for p_ in (dptr, d_ref, d_cost):
        eng.device_free(p_)
    sl = slice(100, 108)
    ora = oracle.OracleSim(pkg.Config(batch=8, **cfg_kwargs).to_struct(), oracle.DERIV_EXACT)
    ora.set_platform_state(pose7=pose[sl].astype(np.float64))
    ora.update(20)
    oc = ora.rollout_velocity(cmds[sl], ref[sl].astype(np.float64))
    assert np.abs(cost[sl] - oc).max() < 1e-6 + 2e-4 * np.abs(oc).max()
    assert (cost[sl].argmin(axis=1) == oc.argmin(axis=1)).mean() >= 0.75  # the MPC would pick the same sample


def test_auto_selected_low_register_kernel_at_131072_robots(pkg, oracle, monkeypatch):
    """Above ~82 000 robots cdpr_create picks the one-step kernel built for two waves per SIMD by itself; at 131 072
    robots two waves really are co-resident.  Same arithmetic => bit-identical to the default kernel (CDPR_LOWREG=0)
    on the same inputs; a slice against the oracle."""
    monkeypatch.setenv("CDPR_MAPPING", "1")
    B = 131072
    rng = np.random.default_rng(1235)
    cfg_kwargs = dict(model=pkg.eight_cable_model(), stages=3)
    cfg = pkg.Config(batch=B, **cfg_kwargs)
    pose = perturbed_poses(cfg.model, B, rng).astype(np.float32)
    cmd = rng.uniform(-0.04, 0.04, (B, 8)).astype(np.float32)
    out = []
    for flag in (None, "0"):
        if flag is None:
            monkeypatch.delenv("CDPR_LOWREG", raising=False)
        else:
            monkeypatch.setenv("CDPR_LOWREG", flag)
        e = pkg.Engine(cfg, 0)
        e.set_platform_state(pose7=pose)
        e.update(15)
        e.set_velocity_command(cmd)
        e.update(45)
        out.append(e.raw_state() + e.joint_states() + e.fk_state())
        if flag is None:
            got = e.platform_state() + e.joint_states()
        e.close()
    for x, y in zip(*out):
        assert np.array_equal(x, y)
    check_slice(pkg, oracle, cfg_kwargs, slice(70000, 70128), pose, [(15, None), (45, cmd)], got)


def test_row_stride_is_padded_at_131072_robots(pkg, monkeypatch):
    """cdpr_create pads the row stride by 128 columns where it would be a multiple of 2 MiB of row distance (HBM channel
    aliasing, profiles/r04_stride_padding.txt): the observable image grows accordingly, records and read-outs follow the
    stride, and nothing a caller sees changes - same bits as a handle forced to the unpadded stride."""
    monkeypatch.setenv("CDPR_MAPPING", "1")
    B, n = 131072, 4
    rng = np.random.default_rng(9)
    cfg = pkg.Config(batch=B, stages=0)
    pose = perturbed_poses(cfg.model, B, rng).astype(np.float32)
    cmd = rng.uniform(-0.04, 0.04, (B, n)).astype(np.float32)
    out, image = [], []
    for pad in (None, "0"):
        if pad is None:
            monkeypatch.delenv("CDPR_STRIDE_PAD", raising=False)
        else:
            monkeypatch.setenv("CDPR_STRIDE_PAD", pad)
        e = pkg.Engine(cfg, 0)
        image.append(e.observable_image_bytes())
        e.set_platform_state(pose7=pose)
        e.update(3)
        e.set_velocity_command(cmd)
        e.update(12)
        rec = e.update_record(4, 2)
        assert np.array_equal(rec["pose"][-1], e.platform_state()[0]) and np.array_equal(rec["effort"][-1], e.joint_states()[2])
        out.append(e.raw_state() + e.joint_states() + (rec["pose"], rec["velocity"]))
        e.close()
    n_obs = image[1] // (B * 16)
    assert image[1] == n_obs * B * 16 and image[0] == n_obs * (B + 128) * 16
    for x, y in zip(*out):
        assert np.array_equal(x, y)


def test_sharded_engine_against_the_oracle(pkg, oracle, monkeypatch):
    """Config-4 placement against the ORACLE: contiguous robot blocks on every device the box has (on a 1-GPU lease:
    two handles on device 0), nothing exchanged; stepping, both command kinds, the concurrent rollout."""
    from cdpr_simulation_amd._native import lib

    monkeypatch.setenv("CDPR_MAPPING", "1")
    ndev = lib().cdpr_device_count()
    devices = list(range(ndev)) if ndev >= 2 else [0, 0]
    B = 500 * len(devices) + 3
    rng = np.random.default_rng(44)
    cfg = pkg.Config(model=pkg.eight_cable_model(), batch=B, stages=3)
    pose = perturbed_poses(cfg.model, B, rng).astype(np.float32)
    cmd = rng.uniform(-0.04, 0.04, (B, 8)).astype(np.float32)
    many = pkg.ShardedEngine(cfg, devices=devices)
    ora = oracle.OracleSim(cfg.to_struct(), oracle.DERIV_EXACT)
    many.set_platform_state(pose7=pose), ora.set_platform_state(pose7=pose.astype(np.float64))
    for e in (many, ora):
        e.update(20)
        assert e.set_velocity_command(cmd) == 0
        e.update(60)
    for name, g, o in zip(("pose", "twist"), many.platform_state(), ora.platform_state()):
        assert np.abs(g - o).max() <= TOL[name], name
    for name, g, o in zip(("q", "qd", "eff"), many.joint_states(), ora.joint_states()):
        assert np.abs(g - o).max() <= TOL[name], name
    cmds = rng.uniform(-0.03, 0.03, (B, 12, 6, 8)).astype(np.float32)
    ref = pose[:, :3] + np.float32([0.0, 0.0, 0.01])
    gc, oc = many.rollout_velocity(cmds, ref), ora.rollout_velocity(cmds, ref.astype(np.float64))
    assert np.abs(gc - oc).max() < 1e-6 + 2e-4 * np.abs(oc).max()
    many.close()


@pytest.mark.parametrize("B", [65536, 131072])
def test_per_robot_handles_at_full_size(pkg, oracle, monkeypatch, B):
    """Per-robot command arrival at the headline size and beyond: 65 536 robots run the PR role-split kernel with every
    SIMD hosting two waves, 131 072 the PR low-register kernel (auto-selected).  A third of the robots stays in Position
    mode, a third gets velocity Joys, a third is switched back and forth (Pids reset at different world steps); three
    slices against the oracle, which is B independent JointForceCalculator sets by construction."""
    monkeypatch.setenv("CDPR_MAPPING", "1")
    rng = np.random.default_rng(1241)
    model = pkg.eight_cable_model()
    kw = dict(model=model, stages=3, perRobotCommands=True)
    pose = perturbed_poses(model, B, rng, 0.03, 0.05).astype(np.float32)
    grp = (np.arange(B) // 7) % 3
    v1 = rng.uniform(-0.03, 0.03, (B, 8)).astype(np.float32)
    v2 = rng.uniform(-0.03, 0.03, (B, 8)).astype(np.float32)
    p1 = rng.uniform(-0.003, 0.003, (B, 8)).astype(np.float32)
    eng = pkg.Engine(pkg.Config(batch=B, **kw), 0)
    eng.set_platform_state(pose7=pose)
    slices = (slice(0, 128), slice(B // 2 - 64, B // 2 + 64), slice(B - 128, B))
    oras = []
    for sl in slices:
        o = oracle.OracleSim(pkg.Config(batch=sl.stop - sl.start, **kw).to_struct(), oracle.DERIV_EXACT)
        o.set_platform_state(pose7=pose[sl].astype(np.float64))
        oras.append(o)

    def both(fn):
        fn(eng, slice(0, B))
        for o, sl in zip(oras, slices):
            fn(o, sl)

    both(lambda e, sl: e.update(11))
    both(lambda e, sl: e.set_velocity_command(v1[sl], mask=(grp >= 1)[sl]))
    both(lambda e, sl: e.update(37))
    both(lambda e, sl: e.set_position_command(p1[sl], mask=(grp == 2)[sl]))
    both(lambda e, sl: e.update(23))
    both(lambda e, sl: (e.set_velocity_command(v2[sl], mask=(grp == 2)[sl]), e.set_velocity_command(v2[sl], mask=(grp == 1)[sl])))
    both(lambda e, sl: e.update(40))
    got = eng.platform_state() + eng.joint_states()
    assert all(np.isfinite(x).all() for x in got)
    for o, sl in zip(oras, slices):
        ref = o.platform_state() + o.joint_states()
        for name, g, r in zip(("pose", "twist", "q", "qd", "eff"), got, ref):
            err = float(np.abs(g[sl] - r).max())
            assert err <= TOL[name], f"B={B}, robots {sl}: {name} differs from the oracle by {err:.3e}"
    eng.close()


def test_config4_as_specified_eight_shards(pkg, oracle, monkeypatch):
    """BASELINE config 4 at its stated size and placement - 524 288 x 8-cable robots in eight contiguous shards of 65 536,
    one handle each, nothing exchanged - on however many devices the box has (all eight shards on device 0 of a 1-GPU
    lease: what eight GPUs run side by side runs back to back here).  Each shard's 65 536 x 8 launch is the role-split
    headline kernel; slices out of three different shards against the oracle, duplicates bit-identical across shards."""
    from cdpr_simulation_amd._native import lib

    monkeypatch.setenv("CDPR_MAPPING", "1")
    ndev = max(lib().cdpr_device_count(), 1)
    devices = [d % ndev for d in range(8)]
    B, per = 524288, 65536
    rng = np.random.default_rng(1240)
    model = pkg.eight_cable_model()
    pose = perturbed_poses(model, B, rng).astype(np.float32)
    cmd = rng.uniform(-0.04, 0.04, (B, 8)).astype(np.float32)
    pose[7 * per:] = pose[:per]  # shard 7 repeats shard 0
    cmd[7 * per:] = cmd[:per]
    many = pkg.ShardedEngine(pkg.Config(model=model, batch=B, stages=3), devices=devices)
    assert [hi - lo for lo, hi in many.spans] == [per] * 8
    many.set_platform_state(pose7=pose)
    many.update(15)
    assert many.set_velocity_command(cmd) == 0
    many.update(45)
    got = many.platform_state() + many.joint_states()
    assert all(np.isfinite(x).all() for x in got)
    for x in got:
        assert np.array_equal(x[:per], x[7 * per:])
    for sl in (slice(64, 192), slice(3 * per + 576, 3 * per + 704), slice(6 * per + per - 128, 7 * per)):
        check_slice(pkg, oracle, dict(model=model, stages=3), sl, pose, [(15, None), (45, cmd)], got)
    many.close()


def test_config5_as_specified_eight_shards(pkg, oracle, monkeypatch):
    """BASELINE config 5 at its stated size: 4 096 robots x 128 sampled sequences x 64-step horizon (33.5 M state-steps per
    rollout), fanned out over eight shards of 512 robots (ShardedEngine.rollout_velocity: launch on every shard, then
    fetch), on however many devices the box has.  Nominal + N(0, 0.01^2) commands per robot and step as SURVEY 8(d)
    states; the sample axis is tiled from 16 drawn samples so that the 1 GiB command tensor is cheap to make - identical
    samples must give identical costs; eight robots out of four shards against the oracle."""
    from cdpr_simulation_amd._native import lib

    monkeypatch.setenv("CDPR_MAPPING", "1")
    ndev = max(lib().cdpr_device_count(), 1)
    devices = [d % ndev for d in range(8)]
    B, S, H, n = 4096, 128, 64, 8
    rng = np.random.default_rng(1236)
    model = pkg.eight_cable_model()
    cfg_kwargs = dict(model=model, stages=3)
    pose = perturbed_poses(model, B, rng).astype(np.float32)
    many = pkg.ShardedEngine(pkg.Config(batch=B, **cfg_kwargs), devices=devices)
    assert [hi - lo for lo, hi in many.spans] == [512] * 8
    many.set_platform_state(pose7=pose)
    many.update(20)
    base = rng.uniform(-0.03, 0.03, (B, H, 1, n)) + rng.normal(0.0, 0.01, (B, H, 16, n))
    cmds = np.tile(base.astype(np.float32), (1, 1, S // 16, 1))
    assert cmds.shape == (B, H, S, n)
    ref = pose[:, :3] + np.float32([0.0, 0.0, 0.01])
    cost = many.rollout_velocity(cmds, ref)
    assert cost.shape == (B, S) and np.isfinite(cost).all() and (cost > 0).all()
    for k in range(1, S // 16):
        assert np.array_equal(cost[:, :16], cost[:, 16 * k:16 * (k + 1)])
    pick = np.array([3, 511, 512, 1500, 2047, 2048, 3333, 4095])
    ora = oracle.OracleSim(pkg.Config(batch=len(pick), **cfg_kwargs).to_struct(), oracle.DERIV_EXACT)
    ora.set_platform_state(pose7=pose[pick].astype(np.float64))
    ora.update(20)
    oc = ora.rollout_velocity(cmds[pick][:, :, :16], ref[pick].astype(np.float64))
    assert np.abs(cost[pick][:, :16] - oc).max() < 1e-6 + 2e-4 * np.abs(oc).max()
    many.close()


@pytest.mark.skipif(os.environ.get("CDPR_SKIP_MULTI_DEVICE") == "1", reason="disabled by CDPR_SKIP_MULTI_DEVICE")
def test_two_or_more_real_devices(pkg, oracle):
    """Only on a multi-GPU box: one handle per REAL device."""
    from cdpr_simulation_amd._native import lib

    ndev = lib().cdpr_device_count()
    if ndev < 2:
        pytest.skip("one GPU visible")
    B = 1024 * ndev
    rng = np.random.default_rng(45)
    cfg = pkg.Config(model=pkg.eight_cable_model(), batch=B, stages=3)
    pose = perturbed_poses(cfg.model, B, rng).astype(np.float32)
    many, one = pkg.ShardedEngine(cfg, devices=range(ndev)), pkg.Engine(cfg, 0)
    for e in (many, one):
        e.set_platform_state(pose7=pose)
        e.update(50)
    assert np.array_equal(many.raw_state()[0], one.raw_state()[0])
    many.close()


def test_bench_gpus_2_is_its_own_launcher_on_the_gpu():
    """`python bench.py --gpus 2` with no torchrun around it: two ranks (sharing the GPU on a 1-GPU lease), the real
    step kernels, n_gpus = 2 in the one JSON line, rollout leg measured under world > 1."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "CDPR_MAPPING", "CDPR_LOWREG"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "100", "--warmup", "20", "--batch", "8192",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    # (a loose floor, ADVICE r04: two ranks of 8 192 robots sharing one GPU reach ~1e9; an order-of-magnitude regression of
    #  the multi-rank path - host placement, blocking waits starving the launch loops - must not pass)
    assert out["n_gpus"] == 2 and out["value"] > 2e7 and out["config"]["state_finite"] is True
    assert all(p["value"] > 1e7 and p["pci"] for p in out["per_rank"]) and out["config"]["rendezvous"] == "socket"  # the default: no torch, no RCCL
    assert out["rollout"]["cost_finite"] is True and "config5: 2 x 512" in out["rollout"]["workload"]
    assert out["parity_check"]["ok_all_ranks"] and out["rollout"]["parity_check"]["ok_all_ranks"] and len(out["per_rank"]) == 2


def test_bench_gpus_8_on_the_one_gpu():
    """The N = 8 launch the driver makes at round end, rehearsed on a 1-GPU lease: eight ranks sharing device 0, each
    pinned to its own cores, the real step kernels and the rollout leg, per-rank figures and the parity checks in the
    one JSON line.  (What it cannot show is scaling; what it does show is that eight ranks start, rendezvous, time,
    check themselves against the oracle and finish.)"""
    import time

    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "CDPR_MAPPING", "CDPR_LOWREG"):
        env.pop(k, None)
    # the first import of scipy on a fresh box pages the image in: do that once here, outside the clock (no torch: the
    # default rendezvous does not import it)
    subprocess.run([sys.executable, "-c", "import scipy.spatial.transform"], env=env, capture_output=True, timeout=900)
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "50", "--warmup", "10", "--batch", "4096",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    took = time.monotonic() - t0
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout
    out = json.loads(lines[0])
    # (loose floors, ADVICE r04: eight ranks of 4 096 robots on ONE GPU and a cgroup of a few CPUs still reach > 1e8 together)
    floor = 2e7 if out["placement"]["cpus_effective"] >= 8 else 2e6
    assert out["n_gpus"] == 8 and out["value"] > floor and out["config"]["state_finite"] is True
    assert [p["rank"] for p in out["per_rank"]] == list(range(8)) and all(p["value"] > floor / 8 and p["pci"] for p in out["per_rank"])
    assert out["parity_check"]["ok"] and out["parity_check"]["ok_all_ranks"] and out["config"]["rendezvous"] == "socket"
    assert out["rollout"]["parity_check"]["ok_all_ranks"] and out["rollout"]["value_device_resident"] > 0
    # (the wall time depends on the box's load and its cgroup CPU quota: bounded generously, reported)
    assert took < 600, f"eight ranks on one GPU took {took:.0f} s"
    print(f"eight ranks on one GPU took {took:.0f} s")


def test_bench_rccl_rendezvous_on_one_rank():
    """The process-group calls bench.py makes under `torch.distributed.run` with one GPU per rank (backend "nccl" = RCCL:
    init with a device id, barrier, max over ranks on a device tensor, the shared-memory barrier's set-up) — forced on for
    a world of ONE rank, the only way to run them on a 1-GPU lease."""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", LOCAL_WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               CDPR_FORCE_RENDEZVOUS="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("CDPR_MAPPING", "CDPR_LOWREG"):
        env.pop(k, None)
    env["CDPR_BENCH_BACKEND"] = "nccl"  # the opt-in (the default rendezvous is the torch-free socket)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "50", "--warmup", "10", "--batch", "8192", "--no-cpu-baseline",
                        "--no-secondary"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["config"]["state_finite"] is True
    assert out["config"].get("rendezvous") == "nccl"


def test_pid_call_counter_never_saturates(pkg, oracle, monkeypatch):
    """the host-side Pid call counter used to clamp at 2^20, which froze the derivative ring (the
    D term went silently wrong after ~17.5 min of sim time).  It is folded with its phase kept now: run past 2^20
    steps on a tiny batch (hipGraph replays) and stay on the oracle."""
    monkeypatch.setenv("CDPR_MAPPING", "1")
    B = 2
    # the 8-cable robot is fully constrained (rank-6 structure matrix); the 4-cable robot's free DoF would wander
    cfg = pkg.Config(model=pkg.eight_cable_model(), batch=B)
    rng = np.random.default_rng(3)
    pose = perturbed_poses(cfg.model, B, rng, 0.02, 0.03).astype(np.float32)
    eng, ora = pkg.Engine(cfg, 0), oracle.OracleSim(cfg.to_struct(), oracle.DERIV_EXACT)
    eng.set_platform_state(pose7=pose), ora.set_platform_state(pose7=pose.astype(np.float64))
    # Velocity mode with a zero command brings the platform to rest (the velocity Pids regulate every cable rate to 0 and
    # their integrators end up holding gravity): a fixed point both precisions converge to, so 10^6 steps do not let
    # fp32 and fp64 drift apart; then a velocity step right after the old saturation point: the D term must see it.
    step_cmd = rng.uniform(-0.01, 0.01, (B, 8)).astype(np.float32)
    for e in (eng, ora):
        e.update(7)
        e.set_velocity_command(np.zeros((B, 8), dtype=np.float32))
        e.update((1 << 20) + 12345)
        e.set_velocity_command(step_cmd)  # same mode: no Pid reset, the call counter keeps running
        e.update(25)
    gq, gqd, ge = eng.joint_states()
    oq, oqd, oe = ora.joint_states()
    assert np.abs(ge - oe).max() < TOL["eff"] and np.abs(gqd - oqd).max() < TOL["qd"]


def test_loaded_models_run_on_the_hip_path(pkg, oracle, monkeypatch):
    """a model that comes out of the loaders (the upstream YAML layout with 8 cables; an SDF in
    gen_cdpr.py's layout) goes through the HIP engine and stays on the oracle."""
    from test_model_io import EIGHT_YAML, mini_sdf

    monkeypatch.setenv("CDPR_MAPPING", "1")
    rng = np.random.default_rng(8)
    m_yaml = pkg.load_yaml(EIGHT_YAML)
    ref = pkg.eight_cable_model()
    world_attach = np.asarray(ref.home_position) + ref.platform_anchors  # identity spawn orientation
    m_sdf = pkg.load_sdf(mini_sdf(ref.frame_anchors, world_attach, list(ref.home_position) + [0, 0, 0]))
    assert m_sdf.n_cables == 8 and m_sdf.mass == 3.0  # mini_sdf's own inertial / damping / effort values
    for model, stages in ((m_yaml, 3), (m_sdf, 3)):
        B = 200
        cfg = pkg.Config(model=model, batch=B, stages=stages)
        pose = perturbed_poses(model, B, rng, 0.03, 0.05).astype(np.float32)
        cmd = rng.uniform(-0.03, 0.03, (B, 8)).astype(np.float32)
        eng, ora = pkg.Engine(cfg, 0), oracle.OracleSim(cfg.to_struct(), oracle.DERIV_EXACT)
        eng.set_platform_state(pose7=pose), ora.set_platform_state(pose7=pose.astype(np.float64))
        for e in (eng, ora):
            e.update(50)
            e.set_velocity_command(cmd)
            e.update(150)
        for name, g, o in zip(("pose", "twist"), eng.platform_state(), ora.platform_state()):
            assert np.abs(g - o).max() <= TOL[name], name
        for name, g, o in zip(("q", "qd", "eff"), eng.joint_states(), ora.joint_states()):
            assert np.abs(g - o).max() <= TOL[name], name


def test_the_kernel_a_handle_launches_is_the_planned_one(pkg, monkeypatch):
    """cdpr_plan_kernel answers from the configuration alone (tests/test_kernel_selection.py pins its table on the CPU); the
    engine routes every launch through the same function.  Here the two are compared on the GPU: after each launch form,
    cdpr_kernel_name (what the handle really ran) equals the plan for that form - fast path (both mappings), per-robot,
    general path (role-split, lean, one-wave at world step 0), precision = 64, scheduled and fused launches."""
    for k in ("CDPR_MAPPING", "CDPR_LOWREG", "CDPR_GEN_SPLIT", "CDPR_GEN_LEAN", "CDPR_PAIR_STREAM", "CDPR_SPLIT", "CDPR_ONESTEP"):
        monkeypatch.delenv(k, raising=False)
    A = pkg._abi
    m8 = pkg.eight_cable_model()
    cases = [
        dict(model=m8, batch=65536, stages=3),
        dict(model=m8, batch=100000, stages=3),
        dict(batch=4096),
        dict(model=m8, batch=2048, stages=3, perRobotCommands=True),
        dict(model=m8, batch=4096, stages=3, velocityEpsilon=0.001),
        dict(model=m8, batch=40000, stages=3, velocityEpsilon=0.001),
        dict(model=m8, batch=300, stages=3, precision=64),
        dict(model=m8, batch=300, stages=0, precision=64),
    ]
    for kw in cases:
        cfg = pkg.Config(**kw)
        e = pkg.Engine(cfg, 0)
        e.update(1)
        assert e.kernel_name == pkg.plan_kernel(cfg, 1, A.PLAN_FIRST_WORLD_STEP), kw
        e.update(14)
        assert e.kernel_name == pkg.plan_kernel(cfg, 1), kw
        e.update(20, 10)
        assert e.kernel_name == pkg.plan_kernel(cfg, 10), kw
        if not cfg.perRobotCommands:
            d = e.device_upload(np.zeros((3, cfg.batch, cfg.n_cables), dtype=np.float32))
            e.update_scheduled(30, 10, d)
            in_launch = int(cfg.precision) != 64 and cfg.velocityEpsilon < 0  # (the other handles serve a schedule as a chain of update calls)
            assert e.kernel_name == pkg.plan_kernel(cfg, 30 if in_launch else 10, A.PLAN_SCHEDULED if in_launch else 0), kw
            e.synchronize()
            e.device_free(d)
        e.close()
