"""bench.py's CPU-side pieces: workload generator, effective core count, the bounded oracle baseline."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_workload_is_seeded_and_shaped(pkg):
    import bench

    m1, pose1, cmd1, n1 = bench.make_workload(pkg, 200, 8, 1235, 55)
    m2, pose2, cmd2, n2 = bench.make_workload(pkg, 200, 8, 1235, 55)
    assert n1 == 6 and pose1.shape == (200, 7) and pose1.dtype == np.float32 and np.array_equal(pose1, pose2)
    c = cmd1(3)
    assert c.shape == (200, 8) and c.dtype == np.float32 and np.array_equal(c, cmd2(3)) and np.all(np.abs(c) <= 0.05)
    assert np.all(c == c[:, :1])  # the same sine on every cable of a robot, as sinevelocitytest publishes
    assert np.abs(np.linalg.norm(pose1[:, 3:], axis=1) - 1).max() < 1e-6
    assert np.abs(pose1[:, :3] - m1.home_pose()[:3]).max() <= 0.05 + 1e-6


def test_effective_cpu_count_is_sane():
    import bench

    n = bench.effective_cpu_count()
    assert 1 <= n <= (os.cpu_count() or 1)


def test_cpu_baseline_runs_on_a_bounded_sample(pkg):
    import bench

    model, pose, command, n_cmd = bench.make_workload(pkg, 300, 8, 1235, 40)
    out = bench.cpu_baseline(pkg, dict(model=model, stages=3), pose, command, 10, target_seconds=0.3)
    assert out["kind"] == "port" and out["unit"] == "state-steps/s" and out["value"] > 1e3 and out["value_1core"] > 1e3
    assert 1 <= out["cores"] <= (os.cpu_count() or 1) and "robots x" in out["sample"]
