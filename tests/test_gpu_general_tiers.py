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
