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


def test_cpu_baseline_reports_the_faithful_mode_too(pkg):
    import bench

    model, pose, command, n_cmd = bench.make_workload(pkg, 64, 4, 1234, 40)
    out = bench.cpu_baseline(pkg, dict(model=model, stages=0), pose, command, 10, target_seconds=0.2)
    # `faithful` = per-step polynomial fit as Pid.cpp:219-247, next to the FIR-equivalent mode
    assert out["value_faithful"] > 1e3 and out["value_faithful_1core"] > 1e3 and "value_faithful" in out["sample"]
    # BASELINE.md section 3's `fir` mode is the headline figure; the fit modes ride along
    assert out["value"] == out["value_fir"] and out["value_1core"] == out["value_fir_1core"] and out["value_exact_fit"] > 1e3
    assert out["value_fir_1core"] > out["value_faithful_1core"]  # 11 multiply-adds against pow() + a 3 x 3 QR per cable-step


def test_config_switch_selects_the_baseline_shapes():
    import bench

    a = bench.parse_args(["--config", "2"])
    assert (a.batch, a.cables) == (4096, 4)
    a = bench.parse_args([])
    assert (a.batch, a.cables, a.gpus) == (65536, 8, 1)
    a = bench.parse_args(["--config", "2", "--batch", "128"])
    assert (a.batch, a.cables) == (128, 4)


def _block_torch(env, tmp_path):
    """PYTHONPATH entry whose `torch` package raises on import: a process that touches torch dies."""
    d = tmp_path / "no_torch" / "torch"
    d.mkdir(parents=True, exist_ok=True)
    (d / "__init__.py").write_text("raise ImportError('torch is blocked in this test: the default rendezvous must not need it')\n")
    env["PYTHONPATH"] = str(tmp_path / "no_torch") + os.pathsep + env.get("PYTHONPATH", "")
    return env


def _run_bench(extra, timeout=300, backend=None, no_torch_dir=None):
    import json
    import subprocess

    env = dict(os.environ, OMP_NUM_THREADS="1")
    env.pop("CDPR_BENCH_BACKEND", None)
    if backend is not None:
        env["CDPR_BENCH_BACKEND"] = backend
    if no_torch_dir is not None:
        _block_torch(env, no_torch_dir)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout  # stdout is the ONE JSON line of rank 0 and nothing else
    return json.loads(lines[0])                                     # (gloo / RCCL notices go to stderr)


def test_bench_gpus_2_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it must itself become two ranks (--dry-run skips the GPU work,
    everything else is the real path) and report n_gpus = 2 - over the default rendezvous, a plain socket, with `import
    torch` made to fail in every rank (north_star: "no PyTorch needed ... no RCCL required")."""
    out = _run_bench(["--gpus", "2", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--dry-run"], no_torch_dir=tmp_path)
    assert out["n_gpus"] == 2 and out["dry_run"] is True and out["steps"] == 20 and out["scaling"] == "weak"
    assert out["metric"].startswith("CDPR sim-steps/sec")
    assert out["rendezvous"] == "socket" and out["rendezvous_fallback"] is None


def test_bench_gpus_8_dry_run_over_the_socket_rendezvous(tmp_path):
    out = _run_bench(["--gpus", "8", "--steps", "50", "--warmup", "10", "--no-cpu-baseline", "--dry-run"], timeout=600, no_torch_dir=tmp_path)
    assert out["n_gpus"] == 8 and out["rendezvous"] == "socket"
    assert [r["rank"] for r in out["per_rank"]] == list(range(8))


def test_bench_gpus_2_over_gloo_opt_in():
    out = _run_bench(["--gpus", "2", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--dry-run"], backend="gloo")
    assert out["n_gpus"] == 2 and out["rendezvous"] == "gloo"


def test_bench_gpus_8_dry_run_pins_and_reports_every_rank():
    """The driver's N = 8 launch, rank plumbing only: eight ranks, disjoint core sets (when the box has at least eight
    CPUs), one JSON line carrying every rank's own figure."""
    # (the "nccl" = RCCL opt-in: on this GPU-less box the pre-flight must end on gloo)
    out = _run_bench(["--gpus", "8", "--steps", "50", "--warmup", "10", "--no-cpu-baseline", "--dry-run"], timeout=600, backend="nccl")
    assert out["n_gpus"] == 8 and out["dry_run"] is True
    assert [r["rank"] for r in out["per_rank"]] == list(range(8))
    # the pre-flight's fields: the rendezvous backend and why it is not RCCL on this GPU-less box, and
    # every rank's device identity (no device here: None, but the keys are there)
    assert out["rendezvous"] == "gloo" and "no GPU" in out["rendezvous_fallback"]
    assert all({"device", "pci", "torch_pci"} <= set(r) for r in out["per_rank"])
    ncpu = len(os.sched_getaffinity(0))
    if ncpu >= 8:
        assert all(r["cpus"] == ncpu // 8 for r in out["per_rank"])
        assert out["placement"]["pinned"][2] == ncpu // 8


def test_rank_cpu_sets_are_disjoint_and_cover_equal_shares():
    import bench

    cpus = list(range(3, 35))  # 32 CPUs, not starting at 0
    sets = [bench.rank_cpu_set(r, 8, cpus)[0] for r in range(8)]
    assert all(len(s) == 4 for s in sets) and len(set().union(*sets)) == 32
    assert bench.rank_cpu_set(0, 1, cpus) == (None, None) and bench.rank_cpu_set(0, 64, cpus) == (None, None)
    assert [sl for sl in bench.parity_slices(65536)] == [slice(0, 64), slice(65472, 65536)]
    assert bench.parity_slices(40) == [slice(0, 40)]


def test_parity_check_flags_a_wrong_result(pkg, oracle):
    """bench.py's post-timing check against the oracle: the oracle's own result passes, a perturbed one fails."""
    import bench

    model, pose, command, n_cmd = bench.make_workload(pkg, 96, 8, 1235, 25)
    kw = dict(model=model, stages=3)
    sim = oracle.OracleSim(pkg.Config(batch=96, **kw).to_struct(), oracle.DERIV_EXACT)
    sim.set_platform_state(pose7=pose.astype(np.float64))
    for j in range(3):
        sim.set_velocity_command(command(j))
        sim.update(10 if j < 2 else 5)
    got = sim.platform_state() + sim.joint_states()
    good = bench.parity_check(pkg, kw, pose, command, 10, 25, got, bench.parity_slices(96, 32))
    assert good["ok"] and good["robots"] == [[0, 32], [64, 96]] and good["max_abs_pose"] < 1e-12 and good["steps"] == 25
    bad = [g.copy() for g in got]
    bad[4][70, 3] += 0.2
    res = bench.parity_check(pkg, kw, pose, command, 10, 25, bad, bench.parity_slices(96, 32))
    assert not res["ok"] and abs(res["max_abs_effort"] - 0.2) < 1e-9


def test_bench_single_rank_needs_no_rendezvous():
    out = _run_bench(["--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--dry-run"])
    assert out["n_gpus"] == 1


def test_bench_fails_loudly_without_a_gpu(pkg):
    """No CPU fallback: on a box without a GPU the real bench must exit non-zero, not print a number."""
    import subprocess

    from cdpr_simulation_amd._native import lib

    if lib().cdpr_device_count() > 0:
        import pytest

        pytest.skip("a GPU is visible")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "64", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_modelled_traffic_matches_the_measured_bytes():
    """bench.py's `roofline.traffic_model` (bytes per launch from the data layout) against the rocprofv3 PMC figures that
    profiles/traffic.json carries: the line must not depend on a stale table alone."""
    import json

    import bench

    measured = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert bench.modelled_traffic_bytes(8, True, 65536) == 65536 * 800
    assert abs(bench.modelled_traffic_bytes(8, True, 65536) / measured["n8_b65536_spl1"] - 1) < 0.01
    # (round 5's pass of the padded layout: 413.6 MB against 419.4 by the layout - a little of the state is still in L2 from the launch before)
    assert abs(bench.modelled_traffic_bytes(8, True, 524288) / measured["n8_b524288_spl1"] - 1) < 0.02
    assert abs(bench.modelled_traffic_bytes(4, False, 4096) / measured["n4_b4096_spl1"] - 1) < 0.06  # 64 waves: per-launch constants show


def test_traffic_table_is_not_older_than_the_newest_pmc_summary():
    """profiles/traffic.json's headline entry must come from the NEWEST profiles/*_bench_pmc_summary.json (round tags sort:
    r02final4 < r03final < r04a < r04final) and equal that summary's figure: scripts/profile_gpu.sh rewrites the table from
    the PMC pass it has just run (scripts/update_traffic.py), and nobody may forget to copy it."""
    import glob
    import json

    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import update_traffic

    table = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_bench_pmc_summary.json")), key=os.path.basename)[-1]
    src = table["_sources"]["n8_b65536_spl1"]
    assert os.path.basename(src["file"]) == os.path.basename(newest), f"traffic.json quotes {src['file']}, newest PMC summary is {newest}"
    name, total = update_traffic.dominant_step_kernel(json.load(open(newest)))
    assert abs(total - table["n8_b65536_spl1"]) < 1.0 and name == src["kernel"]
    for key, s_ in table["_sources"].items():
        assert os.path.exists(os.path.join(ROOT, s_["file"])), f"{key}: source {s_['file']} is not in profiles/"


def _fake_sysfs(root, gpus, nodes):
    """A sysfs tree as an 8-GPU two-socket node shows it: KFD topology nodes (CPU nodes first, simd_count 0), one PCI device
    directory per GPU with its numa_node, one cpulist per NUMA node.  gpus = [(bus, numa_node)], nodes = {node: cpulist}."""
    topo = os.path.join(root, "class", "kfd", "kfd", "topology", "nodes")
    for i in range(len(nodes)):
        os.makedirs(os.path.join(topo, str(i)))
        open(os.path.join(topo, str(i), "properties"), "w").write("cpu_cores_count 48\nsimd_count 0\nlocation_id 0\ndomain 0\n")
    for g, (bus, numa) in enumerate(gpus):
        d = os.path.join(topo, str(len(nodes) + g))
        os.makedirs(d)
        open(os.path.join(d, "properties"), "w").write(f"cpu_cores_count 0\nsimd_count 1024\nlocation_id {bus << 8}\ndomain 0\ndrm_render_minor {128 + g}\n")
        pci = os.path.join(root, "bus", "pci", "devices", f"0000:{bus:02x}:00.0")
        os.makedirs(pci)
        open(os.path.join(pci, "numa_node"), "w").write(f"{numa}\n")
    for node, cpulist in nodes.items():
        d = os.path.join(root, "devices", "system", "node", f"node{node}")
        os.makedirs(d)
        open(os.path.join(d, "cpulist"), "w").write(cpulist + "\n")


def test_rank_placement_follows_the_gpus_numa_nodes(tmp_path, monkeypatch):
    """Topology-aware pinning: rank r drives GPU r, so it takes cores of THAT GPU's NUMA node (an equal share among the
    ranks sharing the node), read from sysfs by PCI address; the contiguous split remains where sysfs says nothing."""
    import bench

    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    root = str(tmp_path / "sys")
    # GPUs 0-3 hang off socket 1, GPUs 4-7 off socket 0 (the inverse of the contiguous split's assumption); hyper-threads
    # of a socket are listed in two ranges, as on real two-socket boxes
    gpus = [(0x05, 1), (0x15, 1), (0x65, 1), (0x75, 1), (0x85, 0), (0x95, 0), (0xE5, 0), (0xF5, 0)]
    _fake_sysfs(root, gpus, {0: "0-47,96-143", 1: "48-95,144-191"})
    assert bench.parse_cpulist("0-3,8,10-11") == {0, 1, 2, 3, 8, 10, 11}
    assert bench.gpu_numa_nodes(root) == [1, 1, 1, 1, 0, 0, 0, 0]
    allowed = set(range(192))
    sets = [bench.rank_cpu_set(r, 8, allowed, bench.gpu_numa_nodes(root), root) for r in range(8)]
    node1, node0 = bench.numa_cpus(1, root), bench.numa_cpus(0, root)
    for r, (cpus, node) in enumerate(sets):
        assert node == (1 if r < 4 else 0) and len(cpus) == 24 and cpus <= (node1 if r < 4 else node0)
    assert len(set().union(*(c for c, _ in sets))) == 192  # disjoint, everything used
    # a cgroup / taskset that leaves only some cores: shares come out of what is allowed on the right node
    some = set(range(40, 60)) | set(range(150, 160))
    c0, n0 = bench.rank_cpu_set(0, 8, some, bench.gpu_numa_nodes(root), root)
    c7, n7 = bench.rank_cpu_set(7, 8, some, bench.gpu_numa_nodes(root), root)
    assert n0 == 1 and c0 <= node1 & some and n7 == 0 and c7 <= node0 & some and not (c0 & c7)
    # visible-device lists re-map the ordinals
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "4,5,0,1")
    assert bench.gpu_numa_nodes(root) == [0, 0, 1, 1]
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "GPU-deadbeef")
    assert bench.gpu_numa_nodes(root) == []
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    # no sysfs / numa_node -1 (single-socket box, VM): the contiguous split
    assert bench.gpu_numa_nodes(str(tmp_path / "nothing")) == []
    cpus, node = bench.rank_cpu_set(3, 8, set(range(32)), [], root)
    assert node is None and cpus == {12, 13, 14, 15}
    cpus, node = bench.rank_cpu_set(3, 8, set(range(32)), [-1] * 8, root)
    assert node is None and cpus == {12, 13, 14, 15}
    # a node none of whose CPUs are allowed: no half-NUMA placement, the contiguous split for everybody
    cpus, node = bench.rank_cpu_set(0, 8, set(range(0, 40)), bench.gpu_numa_nodes(root), root)
    assert node is None and cpus == set(range(0, 5))


def test_bench_dry_run_reports_the_numa_node_of_every_rank(tmp_path):
    """`bench.py --gpus 8 --dry-run` against a fake two-socket sysfs (CDPR_BENCH_SYSFS): placement.numa_node per rank in
    per_rank, the ranks of GPUs 0-3 on node 1, the rest on node 0 - whatever CPUs this box really has (the fake nodes are
    built from them)."""
    import json
    import subprocess

    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 8:
        import pytest

        pytest.skip("fewer than 8 CPUs")
    half = len(allowed) // 2
    fmt = lambda cs: ",".join(str(c) for c in cs)  # noqa: E731
    root = str(tmp_path / "sys")
    gpus = [(0x05, 1), (0x15, 1), (0x65, 1), (0x75, 1), (0x85, 0), (0x95, 0), (0xE5, 0), (0xF5, 0)]
    _fake_sysfs(root, gpus, {0: fmt(allowed[:half]), 1: fmt(allowed[half:])})
    env = dict(os.environ, CDPR_BENCH_BACKEND="gloo", OMP_NUM_THREADS="1", CDPR_BENCH_SYSFS=root)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--dry-run"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.splitlines()[0])
    assert [p["placement"]["numa_node"] for p in out["per_rank"]] == [1, 1, 1, 1, 0, 0, 0, 0]
    assert out["placement"]["gpu_numa_nodes"] == [1, 1, 1, 1, 0, 0, 0, 0] and out["placement"]["numa_node"] == 1
    assert all(p["cpus"] == half // 4 for p in out["per_rank"])


def test_bench_under_torch_distributed_run_uses_the_socket_rendezvous():
    """The driver's N > 1 launch: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...`.  The launcher is torch's, the ranks' rendezvous is not: the default backend is the
    socket star keyed by MASTER_PORT (the agent's own store owns that TCP port, which is why the ranks meet on a Unix socket)."""
    import json
    import subprocess

    env = dict(os.environ, OMP_NUM_THREADS="1")
    env.pop("CDPR_BENCH_BACKEND", None)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29547",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--dry-run"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rendezvous"] == "socket" and [p["rank"] for p in out["per_rank"]] == [0, 1]
