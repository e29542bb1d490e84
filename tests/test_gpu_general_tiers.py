"""The general controller's tiers (cdpr_general_step.hpp, gen_controller) against the oracle on workloads built to land in
each of them: windows that fill, fits after ONE switch between the two Pids of a hold-branch cable (two-run windows: no stamp
in memory), switches that come faster than the window empties (a second gap: the general loop writes the window out), more
queued cables than one pass of the fit queue takes."""
import numpy as np
import pytest

from test_gpu_parity import compare, pair, perturbed_poses

pytestmark = pytest.mark.gpu


def _cfg(pkg, B, eps=0.004, **kw):
    return pkg.Config(model=pkg.eight_cable_model(), batch=B, stages=3, velocityEpsilon=eps, **kw)


@pytest.mark.parametrize("period,kernel", [(25, "split"), (25, "lean"), (7, "split"), (7, "lean"), (25, "one-wave")])
def test_one_robot_per_wave_switches_pids(pkg, oracle, monkeypatch, period, kernel):
    """Robot 0 of every 64 falls into the hold branch and comes back every `period` steps, the others keep their velocity Pid:
    period 25 > window (11): every switch leaves a window with ONE gap (tier 1: ring turned in LDS, the fit on implied stamps of
    two runs); period 7 < window: the window holds an older gap when the next one comes (the general loop, which writes the
    two-run window out first).  Compared with the oracle after every stretch, on the role-split kernel, the lean role-split
    kernel (its cold tail) and the one-wave kernel."""
    monkeypatch.setenv("CDPR_GEN_SPLIT", "1" if kernel == "split" else "0")
    monkeypatch.setenv("CDPR_GEN_LEAN", "1" if kernel == "lean" else "0")
    B = 64 * 3 + 11
    rng = np.random.default_rng(77 + period)
    cfg = _cfg(pkg, B)
    eng, ora = pair(pkg, oracle, cfg, perturbed_poses(cfg.model, B, rng, 0.02, 0.05))
    hi = rng.uniform(0.01, 0.03, (B, 8)).astype(np.float32)
    lo = hi.copy()
    lo[::64] = 0.0
    for j in range(8):
        cmd = lo if j % 2 else hi
        eng.set_velocity_command(cmd), ora.set_velocity_command(cmd)
        for _ in range(period):
            eng.update(1)
        ora.update(period)
        compare(eng, ora, where=f"stretch {j} of {period} steps on the {kernel} kernel")


@pytest.mark.parametrize("kernel", ["split", "lean"])
def test_more_queued_cables_than_one_pass_of_the_queue(pkg, oracle, monkeypatch, kernel):
    """Every robot of a wave switches at once: 512 cables wait for the fit, tier 1 takes 64 - the wave goes to the general loop -
    and a step later the same; then only a few robots switch and tier 1 serves them on the records the general loop left."""
    monkeypatch.setenv("CDPR_GEN_SPLIT", "1" if kernel == "split" else "0")
    monkeypatch.setenv("CDPR_GEN_LEAN", "1" if kernel == "lean" else "0")
    B = 64 * 2
    rng = np.random.default_rng(5)
    cfg = _cfg(pkg, B)
    eng, ora = pair(pkg, oracle, cfg, perturbed_poses(cfg.model, B, rng, 0.02, 0.05))
    hi = rng.uniform(0.01, 0.03, (B, 8)).astype(np.float32)
    all_lo = np.zeros_like(hi)
    few_lo = hi.copy()
    few_lo[3::17] = 0.0
    for j, (cmd, k) in enumerate(((hi, 30), (all_lo, 14), (hi, 14), (few_lo, 20), (hi, 20), (all_lo, 4), (hi, 16))):
        eng.set_velocity_command(cmd), ora.set_velocity_command(cmd)
        for _ in range(k):
            eng.update(1)
        ora.update(k)
        compare(eng, ora, where=f"stretch {j}")


@pytest.mark.parametrize("per_robot", [False, True])
def test_hot_rows_survive_everything_that_touches_the_records(pkg, oracle, monkeypatch, per_robot):
    """On the lean kernel's handles (forced here at a small batch) a robot that has sat in the deep steady state for 63 steps keeps
    mLastTime / mIerr in its hot rows and its H slots go stale in memory.  Everything that reads or resets the records from
    then on has to see through that: cables falling into the hold branch and coming back, MPC rollouts (private copies of
    the records, the one-wave kernel), position and force Joys (a Pid reset: the rows are written back first), a masked
    Joy on a per-robot handle, the trajectory record, a state write, a world reset - one history on the oracle."""
    monkeypatch.setenv("CDPR_GEN_SPLIT", "0")
    monkeypatch.setenv("CDPR_GEN_LEAN", "1")
    B, n = 64 * 2 + 9, 8
    rng = np.random.default_rng(31 + per_robot)
    cfg = _cfg(pkg, B, perRobotCommands=per_robot)
    eng, ora = pair(pkg, oracle, cfg, perturbed_poses(cfg.model, B, rng, 0.02, 0.05))
    hi = rng.uniform(0.01, 0.03, (B, n)).astype(np.float32)
    lo = hi.copy()
    lo[2::7, 1::3] = 0.001  # some cables of some robots below epsilon

    def run(k, where, how="one"):
        if how == "one":
            for _ in range(k):
                eng.update(1)
        elif how == "fused":
            eng.update(k, 5)
        else:
            eng.update_record(k, 4)
        ora.update(k)
        compare(eng, ora, where=where)

    def rollout(where):
        cmds = rng.uniform(-0.03, 0.03, (B, 6, 3, n)).astype(np.float32)
        ref = eng.raw_state()[0][:, :3].astype(np.float64) + np.array([0.0, 0.0, 0.01])
        gc, oc = eng.rollout_velocity(cmds, ref), ora.rollout_velocity(cmds, ref)
        assert np.abs(gc - oc).max() <= 1e-9 + 2e-4 * np.abs(oc).max(), where

    eng.set_velocity_command(hi), ora.set_velocity_command(hi)
    run(90, "into the hot rows")
    rollout("a rollout from records whose H slots are stale")
    run(5, "after the rollout")
    eng.set_velocity_command(lo), ora.set_velocity_command(lo)
    run(30, "cables in the hold branch (their robots left the hot rows)")
    eng.set_velocity_command(hi), ora.set_velocity_command(hi)
    run(100, "back, and into the hot rows again", how="fused")
    pos = rng.uniform(-0.004, 0.004, (B, n)).astype(np.float32)
    kw = {"mask": np.arange(B) % 3 == 0} if per_robot else {}
    eng.set_position_command(pos, **kw), ora.set_position_command(pos, **kw)
    run(80, "position Joy (velocity Pid's rows written back, position Pid reset)")
    rollout("a rollout in position mode")
    frc = (7.0 + rng.uniform(-0.5, 0.5, (B, n))).astype(np.float32)
    kw = {"mask": np.arange(B) % 3 == 1} if per_robot else {}
    eng.set_force_command(frc, **kw), ora.set_force_command(frc, **kw)
    run(12, "force Joy")
    kw = {"mask": np.arange(B) % 2 == 0} if per_robot else {}
    eng.set_velocity_command(hi, **kw), ora.set_velocity_command(hi, **kw)
    run(75, "velocity again", how="record")
    p = perturbed_poses(cfg.model, B, rng, 0.02, 0.05).astype(np.float32)
    eng.set_platform_state(pose7=p), ora.set_platform_state(pose7=p.astype(np.float64))
    run(70, "after a state write")
    eng.reset(), ora.reset()
    eng.set_velocity_command(lo), ora.set_velocity_command(lo)
    run(20, "after a world reset")
