#!/usr/bin/env python3
"""bench.py — CDPR state-steps/sec on MI355X for BASELINE.json's metric.

Workload (per GPU): 65 536 independent 8-cable robots, every
stage on (IK + Newton-Raphson FK (4 iterations) + tension distribution + per-cable PID +
platform dynamics), fp32, 1 ms step, observables written every step, one kernel launch
per world step.  Initial poses = home + U(+-0.05 m) + rotation vector U(+-0.1 rad),
rng(1235); every robot follows its own sine velocity command (amp U(0.01,0.05) m/s,
freq U(0.05,0.5) Hz, phase U(0,2pi)) refreshed every 10 steps from a schedule that is
resident in HBM before the timed region starts.  `--config 2` selects config 2 instead
(4 096 x 4-cable, IK + PID + dynamics).

Multi-GPU (config 4: 524 288 robots = 8 x 65 536; weak scaling, no collective on the data
path): `python bench.py --gpus N` starts N rank processes itself, one per GPU, BEFORE it
touches the GPU (the parent never does); under `torch.distributed.run` (WORLD_SIZE set)
it is one of those ranks.  The rendezvous (barrier, max of the elapsed time over ranks) is a
plain socket + a shared-memory spin barrier (cdpr_simulation_amd/sharding.py): no torch, no RCCL;
CDPR_BENCH_BACKEND=nccl|gloo opts into a torch.distributed process group instead.

Secondary legs on the same line (N = 1; never substituted for `value`): `config2` (BASELINE configs[1]: 4 096 x 4-cable,
cdpr_update_scheduled and one launch per step), `config1` (configs[0]: one robot under the sinevelocitytest publisher, the host in
the loop on every world step, f32 and precision = 64), `fused`, `rollout` (config 5), `general_path`, `fp64`, `large_batch`
(config 4's 524 288 robots on one GPU) - each with its own roofline object (`bound` per leg) and its own replay on the oracle.

One JSON line on stdout (rank 0).  `roofline.achieved` = algorithmic bytes per launch
(4*(39+28n) = 1052 B per state-step at n = 8) / the average launch duration
measured with HIP events on the engine's stream; `roofline.achieved_wall` / `frac_wall` price the
same bytes by the wall clock, i.e. BASELINE.md section 3's `state_steps_per_s x bytes(n) / 8.0e12`
(the more conservative figure); `traffic*` by the bytes a launch really moves (rocprofv3 PMC).  `cpu_baseline` = the fp64 oracle (oracle/, "port") timed on this box's
host cores on a bounded sample of the same workload (rank 0, N = 1 only).
`parity_check` ties the timed run to the oracle: after the timed region (never inside it) the
first and the last 64 robots are replayed on the fp64 oracle for the same warmup + steps under
the same command schedule and compared with what the engine holds; likewise 8 robots of the
rollout leg.  The process exits non-zero when a check fails.  The oracle is the checker and
the CPU baseline here, never the thing measured.

Multi-rank host placement: every rank pins itself to its own share of the CPUs before it
touches the GPU (`placement`), and `per_rank` lists each rank's own rate next to the
max-over-ranks `value`, so a straggler is visible.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_GBS = 6290.0  # what a plain row-copy kernel reaches on this part (scripts/micro/rowcopy.hip, HISTORY.md section 6): the practical ceiling
FP32_VECTOR_PEAK_TFLOPS = 157.3  # same guide
# flop per state-step counted from the gfx950 ISA of the shipped kernels (HISTORY.md section 4)
FLOP_PER_STATE_STEP = {8: 6325, 4: 648}
METRIC = "CDPR sim-steps/sec (whole node), 65 536 parallel 8-cable robots, 1 ms dt"
CONFIGS = {2: dict(batch=4096, cables=4), 3: dict(batch=65536, cables=8)}
SCHED_CHUNK = 1000  # world steps per launch on the scheduled path (small batches)
FUSED_WARM, FUSED_LAUNCHES = 3, 30  # the fused leg's own schedule: untimed / minimum timed launches of 10 steps each
ROLLOUT_SHAPE = (512, 128, 64)  # robots per GPU, sampled sequences, horizon: one GPU's share of BASELINE config 5
GENERAL_SHAPE = (65536, 0.001, 120, 400)  # general-path leg (the contract's size since round 5): robots, velocityEpsilon, untimed steps (past the window fill AND the 63 consecutive calls after which a robot keeps its steady state in the hot rows), timed steps
# ... with the position-hold branch live (robots, velocityEpsilon, warm-up, timed steps, every how-many-th CABLE is held; 0: none - every cable on its
# velocity Pid, one Pid's rows per request; with held and moving cables side by side the lanes of a wave read both Pids' rows)
FP64_HOLD_SHAPES = ((65536, 0.001, 200, 300, 3), (1, 0.001, 100, 300, 3), (65536, 0.001, 200, 300, 0))
FP64_SHAPES = ((65536, 300, 400), (1, 100, 300))  # (robots, warm-up steps, timed steps): the warm-up also brings the clock back up after the host-side replay of the leg before
FP64_TOL = {"pose": 1e-10, "eff": 1e-7}         # fp64 kernels against the fp64 oracle (two double implementations; tests/test_gpu_fp64.py: 1e-13 / 1e-9 over short runs)
LARGE_BATCH_SHAPE = (524288, 100, 200)           # HBM-streaming regime on ONE GPU: robots, untimed steps, timed steps


def make_workload(pkg, batch, n_cables, seed, steps_total, refresh=10, dt=1e-3):
    """Initial poses and the per-robot sine command schedule."""
    from scipy.spatial.transform import Rotation

    model = pkg.eight_cable_model() if n_cables == 8 else pkg.cube_model()
    rng = np.random.default_rng(seed)
    pose = np.tile(model.home_pose(), (batch, 1))
    pose[:, :3] += rng.uniform(-0.05, 0.05, (batch, 3))
    rv = rng.uniform(-0.1, 0.1, (batch, 3))
    pose[:, 3:7] = Rotation.from_rotvec(rv).as_quat()  # x y z w
    amp = rng.uniform(0.01, 0.05, (batch, 1))
    freq = rng.uniform(0.05, 0.5, (batch, 1))
    phase = rng.uniform(0.0, 2 * np.pi, (batch, 1))
    n_cmd = (steps_total + refresh - 1) // refresh

    def command(j):
        t = j * refresh * dt
        return np.repeat((amp * np.sin(2 * np.pi * freq * t + phase)).astype(np.float32), n_cables, axis=1)

    return model, pose.astype(np.float32), command, n_cmd


def make_square_workload(pkg, batch, n_cables, seed, steps_total, which, dt=1e-3):
    """Config 2 under the reference's square publishers: squarepositiontest
    (squarepositiontest.cpp:6-10,21-35: bias + copysign(amp, sin), 10 Hz) or squarevelocitytest (squarevelocitytest.cpp:6-9,20-34),
    one amplitude per robot, the same value on every axis as the publisher sends it; a sample every 100 world steps."""
    import itertools

    from scipy.spatial.transform import Rotation

    model = pkg.eight_cable_model() if n_cables == 8 else pkg.cube_model()
    rng = np.random.default_rng(seed)
    pose = np.tile(model.home_pose(), (batch, 1))
    pose[:, :3] += rng.uniform(-0.01, 0.01, (batch, 3))
    pose[:, 3:7] = Rotation.from_rotvec(rng.uniform(-0.03, 0.03, (batch, 3))).as_quat()
    refresh = 100
    n_cmd = (steps_total + refresh - 1) // refresh
    if which == "squareposition":
        gen, amp = pkg.stimulus.square_position(n_cables, amp=1.0, freq=0.7), rng.uniform(0.002, 0.004, (batch, 1))
    else:
        gen, amp = pkg.stimulus.square_velocity(n_cables, amp=1.0, freq=0.9), rng.uniform(0.01, 0.04, (batch, 1))
    base = np.array(list(itertools.islice(gen, n_cmd)), dtype=np.float64)[:, 0]  # the publisher's value per tick

    def command(j):
        return np.repeat((amp * base[j]).astype(np.float32), n_cables, axis=1)

    return model, pose.astype(np.float32), command, n_cmd, refresh


def make_rollout_commands(n_robots, horizon, samples, n_cables, seed=1236):
    """nominal command + N(0, 0.01^2) per cable per step, [B][H][S][n]."""
    rng = np.random.default_rng(seed)
    nominal = rng.uniform(-0.03, 0.03, (n_robots, horizon, 1, n_cables))
    return (nominal + rng.normal(0.0, 0.01, (n_robots, horizon, samples, n_cables))).astype(np.float32)


def effective_cpu_count():
    """Host cores this process may actually use: the affinity mask capped by the cgroup CPU quota (on the GPU box
    `nproc` says 256 while cpu.max grants 16; spinning 256 OpenMP threads on 16 CPUs throttles them all)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / p + 0.5)))
        except Exception:
            pass
    return n


def cpu_baseline(pkg, cfg_kwargs, pose, command, refresh, target_seconds=12.0, kind="velocity"):
    """Time the fp64 oracle ("port" of the reference step) on the host cores, bounded sample.  Derivative modes
    (BASELINE.md section 3): FIR (`fir`: the fixed 11-tap end-point filter on uniform windows; the headline `value`),
    FAITHFUL (per-step normal equations in absolute time + pow() + column-pivoted QR, as Pid.cpp:219-247), and EXACT
    (the same least-squares problem in centred time: what the parity checks run)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle

    cores = min(oracle.lib().orc_max_threads(), effective_cpu_count())
    sample_b = min(pose.shape[0], 512 * cores)

    def run(nsteps, nb=None, threads=None, mode=oracle.DERIV_EXACT):
        nb = sample_b if nb is None else nb
        cfg = pkg.Config(batch=nb, **cfg_kwargs)
        sim = oracle.OracleSim(cfg.to_struct(), mode)
        sim.set_platform_state(pose7=pose[:nb].astype(np.float64))
        t0 = time.perf_counter()
        done = 0
        while done < nsteps:
            getattr(sim, f"set_{kind}_command")(command(done // refresh)[:nb])
            k = min(refresh, nsteps - done)
            sim.update(k, cores if threads is None else threads)
            done += k
        dt_ = time.perf_counter() - t0
        sim.close()
        return dt_

    def timed(mode, share):
        """`share` of the time budget on the bounded sample with every core, then ~1 s on one core."""
        t_probe = run(probe_steps, mode=mode)
        nsteps = int(max(refresh, min(2000, share * target_seconds * (sample_b * probe_steps / t_probe) / sample_b)))
        t = run(nsteps, mode=mode)
        t1 = run(one_steps, one_b, 1, mode=mode)
        return sample_b * nsteps / t, one_b * one_steps / t1, nsteps, t

    probe_steps = 20
    one_b, one_steps = min(256, pose.shape[0]), 100  # single-core figures on a small sample (about 1 s each)
    v_fir, v_fir1, n_fir, t_fir = timed(oracle.DERIV_FIR, 0.4)
    v_ex, v_ex1, n_ex, t_ex = timed(oracle.DERIV_EXACT, 0.3)
    v_fa, v_fa1, n_fa, t_fa = timed(oracle.DERIV_FAITHFUL, 0.3)
    return {
        # two modes, `fir` (default: the 11-tap derivative, mathematically equal) and `faithful`
        "value": v_fir,
        "value_1core": v_fir1,
        "value_fir": v_fir,
        "value_fir_1core": v_fir1,
        "value_exact_fit": v_ex,
        "value_exact_fit_1core": v_ex1,
        "value_faithful": v_fa,
        "value_faithful_1core": v_fa1,
        "unit": "state-steps/s",
        "cores": int(cores),
        "kind": "port",
        "sample": f"{sample_b} robots x {n_fir} steps of the same workload, fp64 oracle (CPU restatement of the "
                  f"cdpr_gazebo step, not Gazebo/ODE), OpenMP over robots, {t_fir:.1f} s, derivative as the fixed 11-tap filter "
                  f"(`fir`, BASELINE.md section 3; through round 5 `value` was value_exact_fit); value_exact_fit: {n_ex} steps with the "
                  f"least-squares fit in centred time per cable-step, {t_ex:.1f} s; value_faithful: {n_fa} steps with the "
                  f"reference-style per-step polynomial fit (Pid.cpp:219-247: absolute time, pow(), column-pivoted QR), {t_fa:.1f} s",
    }


# fp32 engine against the fp64 oracle after `warmup + steps` steps of the timed workload (absolute; the parity tests'
# tolerances of tests/test_gpu_parity.py widened for a run of several thousand steps under sine commands)
PARITY_TOL = {"pose": 1e-4, "twist": 1e-3, "q": 1e-4, "qd": 1e-3, "eff": 5e-2}


def parity_slices(batch, width=64):
    """Robots replayed on the oracle: the first `width` and the last `width` of the batch (the two ends of the grid: the
    role-swapped and the un-swapped workgroups of the split kernel, the ragged tail)."""
    width = min(width, batch)
    out = [slice(0, width)]
    if batch > width:
        out.append(slice(batch - width, batch))
    return out


def parity_check(pkg, cfg_kwargs, pose, command, refresh, total_steps, got, slices, threads=1, kind="velocity"):
    """The checker for the number this run reports: replay the SAME schedule (initial poses, one Joy batch per `refresh`
    steps, `total_steps` world steps) for the robots of `slices` on the fp64 oracle and compare every published observable
    of the last step with what the engine held right after the timed region.  got = (pose7, twist6, q, qd, effort)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle

    worst = {k: 0.0 for k in PARITY_TOL}
    robots = []
    for sl in slices:
        nb = sl.stop - sl.start
        sim = oracle.OracleSim(pkg.Config(batch=nb, **cfg_kwargs).to_struct(), oracle.DERIV_EXACT)
        sim.set_platform_state(pose7=pose[sl].astype(np.float64))
        done = 0
        while done < total_steps:
            getattr(sim, f"set_{kind}_command")(command(done // refresh)[sl])
            k = min(refresh, total_steps - done)
            sim.update(k, threads)
            done += k
        ref = sim.platform_state() + sim.joint_states()
        sim.close()
        for name, g, o in zip(("pose", "twist", "q", "qd", "eff"), got, ref):
            err = np.abs(np.asarray(g[sl], dtype=np.float64) - o)
            worst[name] = max(worst[name], float(err.max()) if np.isfinite(err).all() else float("inf"))
        robots.append([sl.start, sl.stop])
    return {
        "robots": robots,
        "steps": int(total_steps),
        "max_abs_pose": worst["pose"],
        "max_abs_twist": worst["twist"],
        "max_abs_joint_position": worst["q"],
        "max_abs_joint_velocity": worst["qd"],
        "max_abs_effort": worst["eff"],
        "tolerance": PARITY_TOL,
        "against": "fp64 oracle (oracle/cdpr_oracle.c, EXACT derivative mode), same schedule, observables of the last timed step",
        "ok": bool(all(worst[k] <= PARITY_TOL[k] for k in PARITY_TOL)),
    }


def parse_cpulist(text):
    """'0-3,8,10-11' -> {0, 1, 2, 3, 8, 10, 11} (the kernel's cpulist format)."""
    out = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.update(range(int(lo), int(hi or lo) + 1))
    return out


def gpu_numa_nodes(sysfs="/sys"):
    """NUMA node of every GPU in HIP ordinal order, from sysfs alone (nothing here touches the GPU runtime): the KFD
    topology lists the GPU nodes in the order HIP enumerates them (`simd_count` > 0), each with its PCI address
    (`domain`, `location_id` = bus << 8 | device << 3 | function), and /sys/bus/pci/devices/<address>/numa_node (the
    device directory /sys/class/drm/card*/device links to) says which socket's memory controller it hangs off.
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES given as plain index lists re-map the ordinals.  [] when the tree is
    not there (no amdgpu driver, a container that hides it) or a property cannot be read."""
    base = os.path.join(sysfs, "class", "kfd", "kfd", "topology", "nodes")
    try:
        ids = sorted(int(d) for d in os.listdir(base) if d.isdigit())
    except OSError:
        return []
    nodes = []
    for i in ids:
        try:
            props = dict(line.split()[:2] for line in open(os.path.join(base, str(i), "properties")) if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) == 0:
                continue  # a CPU node
            loc, dom = int(props["location_id"]), int(props.get("domain", "0"))
            addr = f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7:x}"
            numa = int(open(os.path.join(sysfs, "bus", "pci", "devices", addr, "numa_node")).read().strip())
            nodes.append(numa)
        except (OSError, KeyError, ValueError):
            return []
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES"):  # ROCR filters first, HIP indexes what is left
        sel = os.environ.get(var)
        if sel:
            try:
                nodes = [nodes[int(x)] for x in sel.split(",") if x.strip() != ""]
            except (ValueError, IndexError):
                return []  # UUIDs or something else this reader does not follow: no topology claim
    return nodes


def numa_cpus(node, sysfs="/sys"):
    try:
        return parse_cpulist(open(os.path.join(sysfs, "devices", "system", "node", f"node{node}", "cpulist")).read())
    except (OSError, ValueError):
        return set()


def rank_cpu_set(local_rank, local_world, cpus=None, gpu_nodes=None, sysfs="/sys"):
    """Cores rank `local_rank` of `local_world` pins itself to, disjoint between ranks (eight launch loops never meet on one
    core).  With the GPUs' NUMA nodes known (gpu_numa_nodes): rank r drives GPU r % len(gpu_nodes), so it takes an equal
    share of the allowed CPUs of THAT GPU's node, split among the ranks that share the node - a launch loop that issues a
    kernel every 10 us from the far socket pays the inter-socket hop on every doorbell and every completion poll.  Without
    (no sysfs, one node, a node without allowed CPUs): the r-th of `local_world` equal, contiguous shares of the allowed
    CPUs, as before.  Returns (cpu set or None, numa node or None)."""
    cpus = sorted(os.sched_getaffinity(0)) if cpus is None else sorted(cpus)
    if local_world <= 1:
        return None, None
    if gpu_nodes:
        node_of = [gpu_nodes[r % len(gpu_nodes)] for r in range(local_world)]
        mine = node_of[local_rank]
        peers = [r for r in range(local_world) if node_of[r] == mine]
        local = sorted(set(cpus) & numa_cpus(mine, sysfs)) if mine >= 0 else []
        per = len(local) // len(peers)
        if per >= 1 and all(len(set(cpus) & numa_cpus(nd, sysfs)) >= node_of.count(nd) for nd in set(node_of) if nd >= 0) and min(node_of) >= 0:
            k = peers.index(local_rank)
            return set(local[k * per:(k + 1) * per]), mine
    per = len(cpus) // local_world
    if per < 1:
        return None, None
    return set(cpus[local_rank * per:(local_rank + 1) * per]), None


def host_placement(local_rank, local_world, sysfs="/sys"):
    """Called by every rank BEFORE anything touches the GPU (the runtime's helper threads inherit the mask): pin to
    rank_cpu_set (cores of the rank's own GPU's NUMA node where sysfs tells), and when the cgroup grants fewer than 2
    CPUs' worth of time per rank, make every wait of the library a blocking one (CDPR_SYNC_SPIN_US=0): a launch loop and
    a polling wait per rank would otherwise starve each other and show up as "poor scaling".  Returns what was done, for
    the JSON line."""
    info = {"cpus_effective": effective_cpu_count(), "pinned": None, "numa_node": None, "blocking_waits": False}
    if local_world <= 1:
        return info
    if os.environ.get("CDPR_BENCH_NO_PIN") != "1" and hasattr(os, "sched_setaffinity"):
        gpu_nodes = gpu_numa_nodes(sysfs)
        info["gpu_numa_nodes"] = gpu_nodes or None
        mine, node = rank_cpu_set(local_rank, local_world, gpu_nodes=gpu_nodes, sysfs=sysfs)
        if mine:
            try:
                os.sched_setaffinity(0, mine)
                info["pinned"] = [min(mine), max(mine), len(mine)]
                info["numa_node"] = node
            except OSError:
                pass
    info["cpus_per_rank"] = info["cpus_effective"] / local_world
    if info["cpus_per_rank"] < 2.0 and "CDPR_SYNC_SPIN_US" not in os.environ:
        os.environ["CDPR_SYNC_SPIN_US"] = "0"
        info["blocking_waits"] = True
    return info


def device_identity(local_rank, rendezvous_backend):
    """Pre-flight of a rank's GPU: the HIP ordinal this rank will time (local_rank modulo the visible
    devices) with its PCI address as libcdpr_hip sees it, and - when the rendezvous runs on RCCL, i.e. torch drives a GPU
    too - the check that torch's device of the same ordinal is the same PCI device and that both runtimes see the same number
    of GPUs (rank r must not time GPU r while RCCL sits on GPU r').  Returns a dict; `problem` is None when all is well."""
    import ctypes as C

    from cdpr_simulation_amd._native import lib

    ndev = lib().cdpr_device_count()
    info = {"hip_devices": int(ndev), "device": None, "pci": None, "torch_devices": None, "torch_pci": None, "problem": None}
    if ndev <= 0:
        return info
    dev = local_rank % ndev
    buf = C.create_string_buffer(32)
    if lib().cdpr_device_pci_bus_id(dev, buf, 32) == 0:
        info["pci"] = buf.value.decode().lower()
    info["device"] = int(dev)
    if rendezvous_backend == "nccl":
        import torch

        info["torch_devices"] = int(torch.cuda.device_count())
        if info["torch_devices"] != ndev:
            info["problem"] = f"torch sees {info['torch_devices']} GPUs, libcdpr_hip {ndev}"
        else:
            try:
                p = torch.cuda.get_device_properties(dev)
                info["torch_pci"] = f"{int(p.pci_domain_id):04x}:{int(p.pci_bus_id):02x}:{int(p.pci_device_id):02x}"
                if info["pci"] and not info["pci"].startswith(info["torch_pci"]):
                    info["problem"] = f"ordinal {dev} is PCI {info['pci']} for libcdpr_hip but {info['torch_pci']} for torch"
            except AttributeError:  # an older torch without the PCI properties: nothing to compare
                pass
    return info


def leg_bound(kind):
    """`roofline.bound` per leg (not one constant for every workload).  The roof every leg is priced
    against stays HBM (the contract's roofline); `bound` names what the launch is really limited by."""
    return {
        # >= one wave on every SIMD, each carrying one robot's serial chain of dependent vector instructions: neither HBM
        # nor the vector pipes are saturated (profiles/*_bench_pmc_summary.json), the chain's issue rate sets the time
        "chain": "valu-issue",
        # measured traffic at >= 75 % of what a plain copy kernel reaches: the HBM-streaming regime
        "stream": "hbm",
        # fewer waves than SIMDs (4 096 x 4 as lane pairs: 128 waves on 1 024 SIMDs): one wave's instruction latency, most
        # of the chip idle
        "underfilled": "latency (under-filled chip)",
        # one robot with the host in the loop: launch + completion + read-back round trip per world step
        "host": "host round trip",
    }[kind]


def leg_config2(pkg, device, rank, placement, no_parity):
    """BASELINE.json configs[1]: 4 096 parallel 4-cable robots, IK + PID + dynamics, fp32, one GPU - sinevelocitytest-style
    commands (every robot its own sine, a Joy batch every 10 world steps: sinevelocitytest.cpp:34-49 publishes at 100 Hz
    against the 1 ms world step), observables every step (PLG.cpp:236-242).  Both launch forms, each with HIP-event kernel
    time and a replay of the first and last 64 robots on the fp64 oracle:
      scheduled        cdpr_update_scheduled, 1 000 world steps per launch, Joy batches read from HBM inside the kernel
      launch_per_step  one launch per world step (hipGraph replays of 10), the Joy batch copied device to device"""
    Bc, n, refresh = CONFIGS[2]["batch"], CONFIGS[2]["cables"], 10
    warm_s, steps_s = SCHED_CHUNK, 2 * SCHED_CHUNK
    warm_l, steps_l = 200, 1000
    total = max(warm_s + steps_s, warm_l + steps_l)
    model, pose, command, n_cmd = make_workload(pkg, Bc, n, 1234 + rank, total, refresh)
    cfg_kwargs = dict(model=model, stages=0)
    count = Bc * n
    bytes_step = 4 * (39 + 28 * n)
    try:
        ttab = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except (OSError, ValueError):
        ttab = {}
    threads = max(1, min(4, int(placement["cpus_effective"])))
    out = {"workload": f"config2: {Bc} x {n}-cable robots, IK + PID + dynamics, f32, observables every step, per-robot sine "
                       f"jointVelocities refreshed every {refresh} steps from HBM", "unit": "state-steps/s", "dtype": "f32"}

    def roof(us_per_step, traffic_per_step, steps_per_launch):
        ach = bytes_step * Bc / (us_per_step * 1e-6) / 1e9
        return {"bound": leg_bound("underfilled"), "roof": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "traffic": None if traffic_per_step is None else traffic_per_step * steps_per_launch,  # HBM bytes per LAUNCH (rocprofv3 PMC)
                "traffic_frac": None if traffic_per_step is None else traffic_per_step / (us_per_step * 1e-6) / 1e9 / HBM_PEAK_GBS,
                "bytes_per_state_step": bytes_step, "waves": Bc * 2 // 64, "simds": 1024}

    # ---- scheduled form
    eng = pkg.Engine(pkg.Config(batch=Bc, **cfg_kwargs), device=device)
    eng.set_platform_state(pose7=pose)
    d_all = eng.device_upload(np.stack([command(j) for j in range(n_cmd)]))
    eng.update_scheduled(warm_s, refresh, d_all)
    eng.synchronize()
    eng.profile_begin()
    t0 = time.perf_counter()
    done = 0
    while done < steps_s:
        eng.update_scheduled(SCHED_CHUNK, refresh, d_all + ((warm_s + done) // refresh) * count * 4)
        done += SCHED_CHUNK
    ms, nl = eng.profile_end()
    wall = time.perf_counter() - t0
    mapping = eng.mapping
    got = eng.platform_state() + eng.joint_states()
    par = None if no_parity else parity_check(pkg, cfg_kwargs, pose, command, refresh, warm_s + steps_s, got, parity_slices(Bc), threads=threads)
    eng.device_free(d_all)
    eng.close()
    us_step = ms * 1e3 / steps_s
    t_s = ttab.get(f"n{n}_b{Bc}_sched{SCHED_CHUNK}")
    out["scheduled"] = {"launch_form": f"cdpr_update_scheduled: one launch per {SCHED_CHUNK} world steps", "steps_timed": steps_s, "launches": int(nl),
                        "kernel_us_per_step": us_step, "value_per_gpu": Bc / (us_step * 1e-6), "value_wall": Bc * steps_s / wall, "mapping": mapping,
                        "roofline": roof(us_step, None if t_s is None else t_s / SCHED_CHUNK, SCHED_CHUNK), "parity_check": par}

    # ---- one launch per world step
    eng = pkg.Engine(pkg.Config(batch=Bc, **cfg_kwargs), device=device)
    eng.set_platform_state(pose7=pose)
    sched = [eng.device_upload(command(j)) for j in range((warm_l + steps_l) // refresh)]

    def advance(first, nsteps):
        for s_ in range(first, first + nsteps, refresh):
            eng.set_velocity_command_device(sched[s_ // refresh], count)
            eng.update(refresh)

    advance(0, warm_l)
    eng.synchronize()
    eng.profile_begin()
    t0 = time.perf_counter()
    advance(warm_l, steps_l)
    ms, nl = eng.profile_end()
    wall = time.perf_counter() - t0
    got = eng.platform_state() + eng.joint_states()
    par_l = None if no_parity else parity_check(pkg, cfg_kwargs, pose, command, refresh, warm_l + steps_l, got, parity_slices(Bc), threads=threads)
    for p_ in sched:
        eng.device_free(p_)
    eng.close()
    us_launch = ms * 1e3 / max(nl, 1)
    us_step_l = ms * 1e3 / steps_l  # (events bracket the whole region: the gaps between launches are in it)
    out["launch_per_step"] = {"launch_form": "cdpr_update: one launch per world step (hipGraph replays of 10), Joy batch copied device to device every 10 steps",
                              "steps_timed": steps_l, "launches": int(nl), "kernel_us": us_launch, "us_per_step_on_stream": us_step_l,
                              "value_per_gpu": Bc / (us_step_l * 1e-6), "value_wall": Bc * steps_l / wall,
                              "roofline": roof(us_step_l, ttab.get(f"n{n}_b{Bc}_spl1"), 1), "parity_check": par_l}
    out["value_per_gpu"] = out["scheduled"]["value_per_gpu"]
    out["kernel_us_per_step"] = us_step
    out["roofline"] = out["scheduled"]["roofline"]
    out["parity_check"] = None if no_parity else {"ok": bool(par["ok"] and par_l["ok"])}
    return out


def leg_config1(pkg, device, no_parity):
    """BASELINE.json configs[0]: ONE 4-cable robot under the sinevelocitytest publisher (sinevelocitytest.cpp:6-10,34-49: 100 Hz,
    amp 0.05 m/s, 0.1 Hz, the same value on every axis), 1 ms world step, the host in the loop as the reference's plugin
    is (PLG.cpp:202-246): every world step = latch the pending Joy (every 10th step), one launch, read the step's JointState +
    PlatformState back through cdpr_get_observables.  Wall-clock microseconds per world step in f32 and in the reference's
    own arithmetic (precision = 64); the whole trajectory is compared with the fp64 oracle stepping beside it."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    warm, steps, refresh = 200, 2000, 10
    out = {"workload": "config1: one 4-cable robot, sinevelocitytest publisher (100 Hz), 1 ms step, host in the loop: per world step "
                       "one launch + cdpr_get_observables", "unit": "us per world step", "steps_timed": steps}
    ok = True
    for prec in (32, 64):
        cfg = pkg.Config(model=pkg.cube_model(), batch=1, precision=prec)
        eng = pkg.Engine(cfg, device=device)
        gen = pkg.stimulus.sine_velocity(4)
        joys = [next(gen) for _ in range((warm + steps) // refresh)]
        read = eng.observables_f64 if prec == 64 else eng.observables
        traj = np.empty((warm + steps, 7 + 4))
        t0 = 0.0
        for k in range(warm + steps):
            if k == warm:
                eng.synchronize()
                t0 = time.perf_counter()
            if k % refresh == 0:
                eng.set_velocity_command(joys[k // refresh])
            eng.update(1)
            q, qd, e, p, t = read()
            traj[k, :7], traj[k, 7:] = p[0], e[0]
        wall = time.perf_counter() - t0
        mapping = eng.mapping
        eng.close()
        leg = {"us_per_step": wall / steps * 1e6, "steps_per_s": steps / wall, "real_time_factor": 1e-3 / (wall / steps), "mapping": mapping}
        if not no_parity:
            import oracle

            osim = oracle.OracleSim(cfg.to_struct(), oracle.DERIV_EXACT)
            dp = de = 0.0
            for k in range(warm + steps):
                if k % refresh == 0:
                    osim.set_velocity_command(joys[k // refresh])
                osim.update(1)
                dp = max(dp, float(np.abs(osim.platform_state()[0][0] - traj[k, :7]).max()))
                de = max(de, float(np.abs(osim.joint_states()[2][0] - traj[k, 7:]).max()))
            osim.close()
            tol = FP64_TOL if prec == 64 else {"pose": PARITY_TOL["pose"], "eff": PARITY_TOL["eff"]}
            leg["parity_check"] = {"steps": warm + steps, "compared": "pose and effort of every published step", "max_abs_pose": dp, "max_abs_effort": de,
                                   "tolerance": tol, "ok": bool(np.isfinite(traj).all() and dp <= tol["pose"] and de <= tol["eff"])}
            ok = ok and leg["parity_check"]["ok"]
        out["f32" if prec == 32 else "f64"] = leg
    out["us_per_step"] = out["f32"]["us_per_step"]
    out["roofline"] = {"bound": leg_bound("host"), "roof": "hbm", "achieved": 4 * (39 + 28 * 4) / (out["f32"]["us_per_step"] * 1e-6) / 1e9, "peak": HBM_PEAK_GBS,
                       "unit": "GB/s", "frac": 4 * (39 + 28 * 4) / (out["f32"]["us_per_step"] * 1e-6) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                       "note": "one robot is 604 B per step: the figure that matters is us_per_step against the reference's real-time budget of 1 000 us"}
    out["parity_check"] = None if no_parity else {"ok": bool(ok)}
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--config", type=int, default=3, choices=(2, 3), help="BASELINE config: 3 = 65 536 x 8-cable (metric), 2 = 4 096 x 4-cable")
    ap.add_argument("--batch", type=int, default=None, help="robots per GPU (default: the config's)")
    ap.add_argument("--cables", type=int, default=None, choices=(4, 8))
    ap.add_argument("--steps-per-launch", type=int, default=1)
    ap.add_argument("--stimulus", default="sine", choices=("sine", "squareposition", "squarevelocity"),
                    help="command generator: the contract's per-robot sines every 10 steps (default), or the reference's square publishers "
                         "(squarepositiontest / squarevelocitytest: 10 Hz, a Joy every 100 steps)")
    ap.add_argument("--launch-per-step", action="store_true", help="small batches: one launch per world step (hipGraph replays) instead of the scheduled update")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the fused / rollout secondary figures")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-parity-check", action="store_true", help="skip the post-timing replay of two robot slices on the fp64 oracle")
    ap.add_argument("--dry-run", action="store_true", help="rank plumbing only (spawn, rendezvous, report): no GPU work, value 0")
    args = ap.parse_args(argv)
    if args.batch is None:
        args.batch = CONFIGS[args.config]["batch"]
    if args.cables is None:
        args.cables = CONFIGS[args.config]["cables"]
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    return args


def modelled_traffic_bytes(n, fk, batch, steps_per_launch=1):
    """HBM bytes one launch of the fast path moves, from the data layout: float4 rows of 16 B per
    robot — read: platform rows (4, +1 with FK), 5 ring rows per cable pair, the integral rows, the Joy; written: the
    platform rows, ONE ring row per cable pair and the integral rows per step, the observable rows (4 + 3 per four
    cables) per published step.  Within 0.5 %% of the PMC figure where both exist (n = 8: 800 B against 803 B measured
    per robot); reported next to it so that the line never depends on a stale profiles/traffic.json alone."""
    pairs, plat = (n + 1) // 2, 5 if fk else 4
    hot, groups = (pairs + 1) // 2, (n + 3) // 4
    read_rows = plat + 5 * pairs + hot
    write_rows = plat + steps_per_launch * (pairs + hot) + steps_per_launch * (4 + 3 * groups)
    cmd_bytes = 4 * n  # the latched Joy row of the robot, once per launch
    return batch * (16 * (read_rows + write_rows) + cmd_bytes)


def spawn_ranks(n_ranks):
    """`python bench.py --gpus N` with no launcher around it: start N rank processes (one per GPU) and wait.  Runs
    before this process has loaded the HIP library or torch, so the parent never touches the GPU; rank 0's stdout is
    the parent's, so the one JSON line comes out unchanged."""
    with socket.socket() as s, socket.socket() as s2:  # two free ports: MASTER_PORT, and the TCP store of the nccl / gloo opt-in
        s.bind(("127.0.0.1", 0))
        s2.bind(("127.0.0.1", 0))
        port, store_port = s.getsockname()[1], s2.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CDPR_STORE_PORT=str(store_port), CDPR_RDV_PORT=str(store_port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        pending = set(range(n_ranks))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is not None:
                    pending.discard(r)
                    if code != 0:
                        rc = rc or code
            if rc != 0:  # a rank failed: the others would wait in the rendezvous for ever
                for r in pending:
                    procs[r].kill()
                break
            time.sleep(0.02)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
            p.wait()
    return rc


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    # stdout carries the ONE JSON line and nothing else: native libraries (gloo's "[Gloo] Rank 0 is connected ...", RCCL
    # notices) write to file descriptor 1 behind Python's back, so descriptor 1 points at stderr from here on and the
    # line goes out through a private copy of the original descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        os.write(json_fd, (line + "\n").encode())

    from cdpr_simulation_amd.sharding import RankContext

    # host placement first: nothing has touched the GPU yet, so the runtime's threads inherit this rank's core set
    placement = host_placement(int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))),
                               sysfs=os.environ.get("CDPR_BENCH_SYSFS", "/sys"))

    # the rendezvous: barrier + max over ranks + a few gathered words, no data-path collective.  Default: a plain socket
    # between the ranks (no torch, no RCCL: north_star); CDPR_BENCH_BACKEND=nccl|gloo opts into a torch process group
    ctx = RankContext.from_env(backend=os.environ.get("CDPR_BENCH_BACKEND", "socket"))
    rank, local_rank, world = ctx.rank, ctx.local_rank, ctx.world
    n = args.cables
    refresh = 10

    if args.dry_run:  # exercises exactly the rank plumbing (used by the CPU test suite; nothing is measured)
        ctx.barrier()
        ctx.fast_barrier()
        t0 = time.perf_counter()
        time.sleep(0.01)
        ctx.fast_barrier()
        mine = time.perf_counter() - t0
        elapsed = ctx.max_over_ranks(mine)
        per_rank = ctx.gather_over_ranks([mine, float(len(os.sched_getaffinity(0))), float(-1 if placement.get("numa_node") is None else placement["numa_node"])])
        ident = device_identity(local_rank, ctx.backend_name())
        idents = [json.loads(x) for x in ctx.gather_strings(json.dumps(ident))]
        if rank == 0:
            emit(json.dumps({"metric": METRIC, "value": 0.0, "placement": placement, "rendezvous": ctx.backend_name(), "rendezvous_fallback": ctx.fallback,
                             "per_rank": [{"rank": i, "elapsed_s": v[0], "cpus": int(v[1]), "placement": {"numa_node": None if v[2] < 0 else int(v[2])},
                                           "device": idents[i]["device"], "pci": idents[i]["pci"], "torch_pci": idents[i]["torch_pci"]} for i, v in enumerate(per_rank)], "unit": "state-steps/s", "n_gpus": world, "steps": args.steps,
                             "warmup": args.warmup, "ms_per_step": elapsed / max(args.steps, 1) * 1e3, "higher_is_better": True,
                             "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "dry_run": True,
                             "config": {"workload": "dry run: rank plumbing only, no GPU work"}}))
        ctx.close()
        return

    import cdpr_simulation_amd as pkg
    from cdpr_simulation_amd import _abi
    from cdpr_simulation_amd._native import lib

    stages = (_abi.STAGE_FK | _abi.STAGE_TD) if n == 8 else 0
    total = args.warmup + args.steps
    seed = (1235 if n == 8 else 1234) + rank
    kind = "position" if args.stimulus == "squareposition" else "velocity"
    if args.stimulus == "sine":
        model, pose, command, n_cmd = make_workload(pkg, args.batch, n, seed, total, refresh)
    else:
        model, pose, command, n_cmd, refresh = make_square_workload(pkg, args.batch, n, seed, total, args.stimulus)
    cfg_kwargs = dict(model=model, stages=stages)
    cfg = pkg.Config(batch=args.batch, **cfg_kwargs)
    ndev = lib().cdpr_device_count()
    device = local_rank % max(ndev, 1)  # one rank per GPU; ranks only share a GPU on a box with fewer GPUs than ranks
    ident = device_identity(local_rank, ctx.backend_name())
    idents = [json.loads(x) for x in ctx.gather_strings(json.dumps(ident))]
    problems = [f"rank {i}: {d['problem']}" for i, d in enumerate(idents) if d.get("problem")]
    if problems:  # every rank sees the same list: all stop together, before any GPU work
        if rank == 0:
            print("bench.py: device pre-flight FAILED: " + "; ".join(problems), file=sys.stderr)
        ctx.close()
        sys.exit(4)
    eng = pkg.Engine(cfg, device=device)
    eng.set_platform_state(pose7=pose)
    # command schedule resident in HBM before timing starts
    sched = [eng.device_upload(command(j)) for j in range(n_cmd)]
    count = args.batch * n

    # Small batches (config 2: a 4 096 x 4 step is 1.4 us of work behind 3-4 us of launch) take the whole schedule in one
    # launch per SCHED_CHUNK steps (cdpr_update_scheduled: the Joy batch of every 10th step is read from the schedule in
    # HBM inside the kernel, every step's observables are written, bit-identical to the launch-per-step path: tested)
    scheduled = args.batch * n <= 131072 and args.steps_per_launch == 1 and not args.launch_per_step
    d_sched_all = eng.device_upload(np.stack([command(j) for j in range(n_cmd)])) if scheduled else 0

    def advance(first_step, nsteps):
        done = 0
        while done < nsteps:
            s = first_step + done
            if scheduled and s % refresh == 0:
                k = min(SCHED_CHUNK, nsteps - done)
                eng.update_scheduled(k, refresh, d_sched_all + (s // refresh) * count * 4, kind=kind)
                done += k
                continue
            if s % refresh == 0 and args.batch * n <= 131072:
                # launch per step on a small batch: the launches replay from captured hipGraphs, and a graph is tied to the buffer
                # its kernels read, so the Joy batch is copied (device to device, 64 KiB) into the handle's own latched buffer
                getattr(eng, f"set_{kind}_command_device")(sched[s // refresh], count)
            elif s % refresh == 0:  # the schedule lives in HBM: the engine reads the Joy batch in place (zero copy)
                getattr(eng, f"bind_{kind}_command_device")(sched[s // refresh], count)
            k = min(refresh - s % refresh, nsteps - done)
            eng.update(k, args.steps_per_launch)
            done += k

    def barrier():
        eng.synchronize()  # hipStreamSynchronize on the engine's stream (all of this process's GPU work)
        ctx.barrier()      # every rank here (socket rendezvous; + torch.cuda.synchronize() on the nccl opt-in) when N > 1

    def edge():
        """Edge of a timed region: this rank's GPU work done, then every rank here.  The cross-rank part is a shared-memory
        spin barrier (microseconds): an RCCL / gloo barrier costs ~100 us, a third of a 20-step timed region."""
        eng.synchronize()
        ctx.fast_barrier()

    advance(0, args.warmup)
    barrier()
    edge()
    eng.profile_begin()
    t0 = time.perf_counter()
    advance(args.warmup, args.steps)
    ev_ms, launches = eng.profile_end()  # records the closing event and synchronises the engine's stream: this rank is done
    ctx.fast_barrier()                   # ... and so is every other rank
    elapsed_mine = time.perf_counter() - t0
    elapsed = ctx.max_over_ranks(elapsed_mine)
    # every rank's own figures, so that a straggler shows next to the max-over-ranks value
    per_rank = ctx.gather_over_ranks([elapsed_mine, ev_ms * 1e3 / max(launches, 1), float(device),
                                      float(-1 if placement.get("numa_node") is None else placement["numa_node"]),
                                      float(placement["pinned"][0] if placement.get("pinned") else -1), float(placement["pinned"][2] if placement.get("pinned") else 0)])

    pose_end, twist_end = eng.platform_state()
    joint_end = eng.joint_states()
    finite = bool(np.isfinite(pose_end).all())

    # ---- parity of what was just timed (outside the timed region): two robot slices replayed on the fp64 oracle
    parity = None
    if not args.no_parity_check:
        parity = parity_check(pkg, cfg_kwargs, pose, command, refresh, total, (pose_end, twist_end) + tuple(joint_end),
                              parity_slices(args.batch), threads=max(1, min(4, int(placement["cpus_effective"] // max(world, 1)))), kind=kind)
        worst_ok = ctx.min_over_ranks(1.0 if parity["ok"] else 0.0)
        parity["ok_all_ranks"] = bool(worst_ok > 0.5)

    # ---- secondary figures (every rank runs them, outside the timed region above; never substituted for `value`)
    secondary = {}
    if args.steps_per_launch == 1 and not args.no_secondary and args.stimulus == "sine":
        # (a) same workload with the 10 steps of each command hold fused into one launch: state stays on chip
        #     between the steps, observables are still written every step
        # A schedule of its own, whatever --steps says (the driver's --steps 20 used to leave this leg two cold launches):
        # FUSED_WARM launches untimed, then FUSED_LAUNCHES (or --steps / 10, if that is more, up to 200) timed.
        spl = refresh
        launches_timed = max(FUSED_LAUNCHES, min(args.steps, 2000) // refresh)
        steps2 = launches_timed * refresh
        start = args.warmup + args.steps
        sched2 = [eng.device_upload(command((start + j * refresh) // refresh)) for j in range(FUSED_WARM + launches_timed)]
        image = eng.observable_image_bytes()
        d_rec = eng.device_upload(np.zeros(image * refresh, dtype=np.uint8))  # trajectory record of one command hold
        for j in range(FUSED_WARM):
            eng.bind_velocity_command_device(sched2[j], count)
            eng.update_record_device(refresh, spl, d_rec, image * refresh)
        barrier()
        edge()
        eng.profile_begin()
        t0 = time.perf_counter()
        for j in range(FUSED_WARM, FUSED_WARM + launches_timed):
            eng.bind_velocity_command_device(sched2[j], count)
            eng.update_record_device(refresh, spl, d_rec, image * refresh)  # every step's observables stay in HBM
        ms2, launches2 = eng.profile_end()
        ctx.fast_barrier()
        el2 = ctx.max_over_ranks(time.perf_counter() - t0)
        eng.device_free(d_rec)
        for p_ in sched2:
            eng.device_free(p_)
        # per launch: command n + state round trip 2*(13+12n) + spl * observables (13+3n), in floats
        bytes_launch = 4 * (n + 2 * (13 + 12 * n) + spl * (13 + 3 * n))
        v2 = world * args.batch * steps2 / el2
        secondary["fused"] = {
            "steps_per_launch": spl,
            "launches_timed": launches_timed,
            "launches_warmup": FUSED_WARM,
            "value": v2,
            "unit": "state-steps/s",
            "kernel_us": ms2 * 1e3 / max(launches2, 1),
            "bytes_per_state_step": bytes_launch / spl,
            "achieved_GBps": bytes_launch * args.batch / (ms2 * 1e-3 / max(launches2, 1)) / 1e9,
            "f32_tflops": v2 / world * FLOP_PER_STATE_STEP[n] / 1e12,
            "f32_vector_frac": v2 / world * FLOP_PER_STATE_STEP[n] / 1e12 / FP32_VECTOR_PEAK_TFLOPS,
            "note": "compute (f32 VALU) bound: state never leaves the registers between the fused steps; the observables "
                    "of every step are kept in a trajectory record in HBM (cdpr_update_record), nothing published is dropped",
        }
        # (b) MPC rollout, BASELINE config 5: 512 robots x 128 samples x 64 steps on every GPU (4 096 robots on 8)
        if n == 8:
            Br, S, H = ROLLOUT_SHAPE
            cfg_r = pkg.Config(batch=Br, **cfg_kwargs)
            er = pkg.Engine(cfg_r, device=device)
            er.set_platform_state(pose7=pose[:Br])
            er.update(20)
            cmds = make_rollout_commands(Br, H, S, n, seed=1236 + rank)
            dptr = er.device_upload(cmds)
            ref = pose[:Br, :3].astype(np.float32)
            cost = er.rollout_velocity((dptr, S, H), ref)  # warm-up (also sizes the persistent scratch)
            reps = 10
            er.synchronize()
            ctx.barrier()
            ctx.fast_barrier()
            t0 = time.perf_counter()
            for _ in range(reps):
                cost = er.rollout_velocity((dptr, S, H), ref)  # ref upload + kernel + cost download, synchronous
            ctx.fast_barrier()
            elr = ctx.max_over_ranks(time.perf_counter() - t0) / reps
            # kernel alone: device-resident reference and costs, HIP events on the engine's stream
            d_ref = er.device_upload(ref)
            d_cost = er.device_alloc(Br * S * 4)
            er.rollout_velocity_device(dptr, S, H, d_ref, d_cost)
            er.synchronize()
            ctx.fast_barrier()
            t0 = time.perf_counter()
            er.profile_begin()
            for _ in range(reps):
                er.rollout_velocity_device(dptr, S, H, d_ref, d_cost)
            msr, nl = er.profile_end()
            ctx.fast_barrier()
            el_dev = ctx.max_over_ranks(time.perf_counter() - t0) / reps  # whole node, sampler and costs resident in HBM
            cost_dev = er.device_download(d_cost, (Br, S))
            roll_parity = None
            if not args.no_parity_check:  # 8 robots of this rollout on the fp64 oracle
                sys.path.insert(0, os.path.join(ROOT, "oracle"))
                import oracle

                rs = slice(Br - 8, Br)
                osim = oracle.OracleSim(pkg.Config(batch=8, **cfg_kwargs).to_struct(), oracle.DERIV_EXACT)
                osim.set_platform_state(pose7=pose[rs].astype(np.float64))
                osim.update(20)
                oc = osim.rollout_velocity(cmds[rs], ref[rs].astype(np.float64))
                osim.close()
                rel = float(np.abs(cost[rs] - oc).max() / max(float(np.abs(oc).max()), 1e-30))
                roll_parity = {"robots": [Br - 8, Br], "trajectories": int(8 * S), "max_rel_cost": rel, "tolerance": 2e-4,
                               "same_best_sample": float((cost[rs].argmin(axis=1) == oc.argmin(axis=1)).mean()),
                               "ok": bool(np.isfinite(cost).all() and rel <= 2e-4)}
                roll_parity["ok_all_ranks"] = bool(ctx.min_over_ranks(1.0 if roll_parity["ok"] else 0.0) > 0.5)
            for p_ in (dptr, d_ref, d_cost):
                er.device_free(p_)
            er.close()
            kern_s = msr * 1e-3 / max(nl, 1)
            secondary["rollout"] = {
                "workload": f"config5: {world} x {Br} robots x {S} sampled sequences x {H}-step horizon, cost copied back to the host",
                "value": world * Br * S * H / elr,
                "unit": "state-steps/s",
                "ms_per_rollout": elr * 1e3,
                "kernel_us": kern_s * 1e6,
                "kernel_value_per_gpu": Br * S * H / kern_s,
                "f32_tflops": Br * S * H / kern_s * FLOP_PER_STATE_STEP[n] / 1e12,
                "f32_vector_frac": Br * S * H / kern_s * FLOP_PER_STATE_STEP[n] / 1e12 / FP32_VECTOR_PEAK_TFLOPS,
                # the same rollout with the sampler's commands, the reference and the costs resident in HBM
                # (cdpr_rollout_velocity_device): wall clock over all ranks, no host copy inside
                "value_device_resident": world * Br * S * H / el_dev,
                "cost_finite": bool(np.isfinite(cost).all()),
                "device_path_identical": bool(np.array_equal(cost, cost_dev)),
                "parity_check": roll_parity,
            }

        # (c) the general controller path (position-hold branch live: velocityEpsilon > 0 keeps both Pids of every cable
        #     alive), one launch per step, 65 536 x 8 with every stage (rounds 3-4: 16 384 x 8)
        if n == 8 and args.config == 3:
            Bg, eps_g, warm_g, steps_g = GENERAL_SHAPE
            Bg = min(Bg, args.batch)
            model_g, pose_g, command_g, _ = make_workload(pkg, Bg, n, 1235 + rank, refresh)
            cfg_g = pkg.Config(batch=Bg, **dict(cfg_kwargs, velocityEpsilon=eps_g))
            eg = pkg.Engine(cfg_g, device=device)
            eg.set_platform_state(pose7=pose_g)
            d_cmd_g = eg.device_upload(command_g(0))
            eg.bind_velocity_command_device(d_cmd_g, Bg * n)
            eg.update(warm_g)  # past the window fill: every Pid on a full uniform window
            eg.synchronize()
            eg.profile_begin()
            eg.update(steps_g)
            msg, nlg = eg.profile_end()
            gen_parity = None
            if not args.no_parity_check:  # 64 robots of this run on the fp64 oracle (hold branch, both Pids, fit on real stamps)
                sys.path.insert(0, os.path.join(ROOT, "oracle"))
                import oracle

                gs = slice(Bg - 64, Bg)
                osim = oracle.OracleSim(pkg.Config(batch=64, **dict(cfg_kwargs, velocityEpsilon=eps_g)).to_struct(), oracle.DERIV_EXACT)
                osim.set_platform_state(pose7=pose_g[gs].astype(np.float64))
                osim.set_velocity_command(command_g(0)[gs])
                osim.update(warm_g + steps_g)
                gp, ge = eg.platform_state()[0][gs], eg.joint_states()[2][gs]
                op, oe = osim.platform_state()[0], osim.joint_states()[2]
                osim.close()
                dp, de = float(np.abs(gp - op).max()), float(np.abs(ge - oe).max())
                gen_parity = {"robots": [Bg - 64, Bg], "steps": warm_g + steps_g, "max_abs_pose": dp, "max_abs_effort": de,
                              "tolerance": {"pose": PARITY_TOL["pose"], "eff": PARITY_TOL["eff"]},
                              "ok": bool(np.isfinite(gp).all() and dp <= PARITY_TOL["pose"] and de <= PARITY_TOL["eff"])}
            eg.device_free(d_cmd_g)
            eg.close()
            # ... and with the cables SWITCHING between their two Pids: velocityEpsilon = 0.004 in the middle of
            # the commands' amplitudes (0.01-0.05 m/s sines), the schedule refreshed every 10 steps - a cable crosses epsilon twice a
            # period, each crossing leaves a window with a gap (the fit on real stamps for the next ten steps)
            eps_s, warm_sw, periods_sw = 0.004, 120, 30
            model_s, pose_s, command_s, _ = make_workload(pkg, Bg, n, 1235 + rank, warm_sw + periods_sw * refresh, refresh)
            kw_s = dict(cfg_kwargs, velocityEpsilon=eps_s)
            es = pkg.Engine(pkg.Config(batch=Bg, **kw_s), device=device)
            es.set_platform_state(pose7=pose_s)
            sched_s = [es.device_upload(command_s(j)) for j in range(warm_sw // refresh + periods_sw)]
            for j in range(warm_sw // refresh):
                es.bind_velocity_command_device(sched_s[j], Bg * n)
                es.update(refresh)
            es.synchronize()
            es.profile_begin()
            for j in range(warm_sw // refresh, warm_sw // refresh + periods_sw):
                es.bind_velocity_command_device(sched_s[j], Bg * n)
                es.update(refresh)
            mss, nls = es.profile_end()
            sw_parity = None
            if not args.no_parity_check:
                gs = slice(Bg - 64, Bg)
                osim = oracle.OracleSim(pkg.Config(batch=64, **kw_s).to_struct(), oracle.DERIV_EXACT)
                osim.set_platform_state(pose7=pose_s[gs].astype(np.float64))
                for j in range(warm_sw // refresh + periods_sw):
                    osim.set_velocity_command(command_s(j)[gs])
                    osim.update(refresh)
                gp, ge = es.platform_state()[0][gs], es.joint_states()[2][gs]
                dp, de = float(np.abs(gp - osim.platform_state()[0]).max()), float(np.abs(ge - osim.joint_states()[2]).max())
                osim.close()
                held_now = float((np.abs(command_s(warm_sw // refresh + periods_sw - 1)) <= eps_s).mean())
                sw_parity = {"robots": [Bg - 64, Bg], "steps": warm_sw + periods_sw * refresh, "max_abs_pose": dp, "max_abs_effort": de, "held_cable_share_last_period": held_now,
                             "tolerance": {"pose": PARITY_TOL["pose"], "eff": PARITY_TOL["eff"]},
                             "ok": bool(np.isfinite(gp).all() and dp <= PARITY_TOL["pose"] and de <= PARITY_TOL["eff"])}
            for p_ in sched_s:
                es.device_free(p_)
            es.close()
            kus_sw = mss * 1e3 / max(nls, 1)
            kus = msg * 1e3 / max(nlg, 1)
            gen_traffic = None  # HBM bytes per launch of this leg's kernel (rocprofv3 PMC, profiles/traffic.json)
            try:
                gen_traffic = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(f"general_n{n}_b{Bg}_spl1")
            except (OSError, ValueError):
                pass
            secondary["general_path"] = {
                "workload": f"{Bg} x {n}-cable robots, every stage, velocityEpsilon = {eps_g} (position-hold branch live: general controller path), "
                            "one launch per step, one held Joy",
                "kernel_us": kus,
                "value_per_gpu": Bg / (kus * 1e-6),
                "unit": "state-steps/s",
                "steps_timed": steps_g,
                "bound": leg_bound("chain"),
                "frac": 4 * (39 + 28 * n) * Bg / (kus * 1e-6) / 1e9 / HBM_PEAK_GBS,  # algorithmic bytes per launch / kernel time / 8 TB/s
                "traffic": gen_traffic,
                "traffic_frac": (gen_traffic / (kus * 1e-6) / 1e9 / HBM_PEAK_GBS) if gen_traffic else None,
                "switching": {"workload": f"velocityEpsilon = {eps_s}, per-robot sines refreshed every {refresh} steps: cables keep switching between their two Pids",
                              "kernel_us": kus_sw, "value_per_gpu": Bg / (kus_sw * 1e-6), "steps_timed": periods_sw * refresh,
                              "frac": 4 * (39 + 28 * n) * Bg / (kus_sw * 1e-6) / 1e9 / HBM_PEAK_GBS, "parity_check": sw_parity},
                "parity_check": None if gen_parity is None else dict(gen_parity, ok=bool(gen_parity["ok"] and (sw_parity or {"ok": True})["ok"])),
            }

        # (d) the step in the reference's own precision (cdpr_config_t.precision = 64: Pid.h and Gazebo/ODE compute in double):
        #     the contract's size and one robot, kernel time by HIP events, 64 robots replayed on the fp64 oracle
        if n == 8 and args.config == 3:
            fp64_legs = []
            hold_legs = []
            for Bf, eps_f, warm_f, steps_f, held_every in tuple((b, None, w, k, 0) for b, w, k in FP64_SHAPES) + FP64_HOLD_SHAPES:
                Bf = min(Bf, args.batch)  # (--batch below the contract's size: the leg follows)
                cfg_kwargs_f = cfg_kwargs if eps_f is None else dict(cfg_kwargs, velocityEpsilon=eps_f)
                ef = pkg.Engine(pkg.Config(batch=Bf, precision=64, **cfg_kwargs_f), device=device)
                ef.set_platform_state(pose7=pose[:Bf])
                cmd_f = command(0)[:Bf].copy()
                held_share = None
                if eps_f is not None:
                    # hold legs: every third CABLE (not whole robots) commanded 0 <= velocityEpsilon, so that the hold
                    # branch runs - and the two Pids of a robot diverge per cable - at every size including the one-robot leg,
                    # and the parity slice holds held cables
                    held = ((np.arange(Bf * n).reshape(Bf, n) % held_every) == 0) if held_every else np.zeros((Bf, n), dtype=bool)
                    cmd_f[held] = 0.0
                    cmd_f[~held & (np.abs(cmd_f) <= eps_f)] = 0.02  # (the others clearly above epsilon: on the velocity Pid)
                    held_share = float(held.mean())
                d_cmd_f = ef.device_upload(cmd_f)
                ef.bind_velocity_command_device(d_cmd_f, Bf * n)
                ef.update(warm_f)
                ef.synchronize()
                ef.profile_begin()
                ef.update(steps_f)
                msf, nlf = ef.profile_end()
                fpar = None
                if not args.no_parity_check:
                    sys.path.insert(0, os.path.join(ROOT, "oracle"))
                    import oracle

                    fs = slice(max(0, Bf - 64), Bf)
                    osim = oracle.OracleSim(pkg.Config(batch=fs.stop - fs.start, **cfg_kwargs_f).to_struct(), oracle.DERIV_EXACT)
                    osim.set_platform_state(pose7=pose[fs].astype(np.float64))
                    osim.set_velocity_command(cmd_f[fs])
                    osim.update(warm_f + steps_f)
                    g64 = ef.observables_f64()
                    dp = float(np.abs(g64[3][fs] - osim.platform_state()[0]).max())
                    de = float(np.abs(g64[2][fs] - osim.joint_states()[2]).max())
                    osim.close()
                    fpar = {"robots": [fs.start, fs.stop], "steps": warm_f + steps_f, "max_abs_pose": dp, "max_abs_effort": de, "tolerance": FP64_TOL,
                            "ok": bool(np.isfinite(g64[3]).all() and dp <= FP64_TOL["pose"] and de <= FP64_TOL["eff"])}
                ef.device_free(d_cmd_f)
                ef.close()
                kus_f = msf * 1e3 / max(nlf, 1)
                (fp64_legs if eps_f is None else hold_legs).append({"robots": Bf, "kernel_us": kus_f, "value_per_gpu": Bf / (kus_f * 1e-6), "steps_timed": steps_f,
                                                                     "velocity_epsilon": eps_f, "held_cable_share": held_share, "parity_check": fpar})
            secondary["fp64"] = {
                "workload": f"precision = 64 (the reference's own arithmetic): {n}-cable robots, every stage, one launch per step, one held Joy",
                "note": "the part runs chip-wide fp64 at a shader clock of 1.65-1.85 GHz (2.4 GHz at <= 4 096 robots): s_memtime against s_memrealtime inside the kernel, profiles/r06_fp64_timeline.txt",
                "dtype": "f64",
                "unit": "state-steps/s",
                "bound": leg_bound("chain"),
                "frac": 4 * (39 + 28 * n) * fp64_legs[0]["robots"] / (fp64_legs[0]["kernel_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS,  # the contract's (f32) bytes; in doubles twice that
                "legs": fp64_legs,
                "value_per_gpu": fp64_legs[0]["value_per_gpu"],
                "kernel_us": fp64_legs[0]["kernel_us"],
                "kernel_us_one_robot": fp64_legs[-1]["kernel_us"],
                # the position-hold branch live (velocityEpsilon >= 0; a third of the cables commanded 0 and held by their position
                # Pid, the others on their velocity Pid: held_cable_share) in double
                # (kernel_us: the leg with a third of the cables held, as since round 5; kernel_us_none_held: every cable on its velocity Pid)
                "hold_branch": {"legs": hold_legs, "kernel_us": hold_legs[0]["kernel_us"], "kernel_us_one_robot": hold_legs[1]["kernel_us"],
                                "kernel_us_none_held": hold_legs[2]["kernel_us"]},
                "parity_check": {"ok": all((l["parity_check"] or {"ok": True})["ok"] for l in fp64_legs + hold_legs)} if not args.no_parity_check else None,
            }

        # (e) the HBM-streaming regime: 524 288 x 8 on ONE GPU (a 65 536-robot handle's ~40 MB of state and observables stay in
        #     L2 + Infinity Cache between launches; eight times that does not), one launch per step
        if n == 8 and args.config == 3 and world == 1:
            Bl, warm_l, steps_l = LARGE_BATCH_SHAPE
            model_l, pose_l, command_l, _ = make_workload(pkg, Bl, n, 1235, refresh)
            el = pkg.Engine(pkg.Config(batch=Bl, **cfg_kwargs), device=device)
            el.set_platform_state(pose7=pose_l)
            d_cmd_l = el.device_upload(command_l(0))
            el.bind_velocity_command_device(d_cmd_l, Bl * n)
            el.update(warm_l)
            el.synchronize()
            el.profile_begin()
            tl0 = time.perf_counter()
            el.update(steps_l)
            msl, nll = el.profile_end()
            wall_l = time.perf_counter() - tl0
            lpar = None
            if not args.no_parity_check:
                lp_, lt_ = el.platform_state()
                lpar = parity_check(pkg, cfg_kwargs, pose_l, lambda j: command_l(0), refresh, warm_l + steps_l, (lp_, lt_) + tuple(el.joint_states()),
                                    parity_slices(Bl), threads=max(1, min(4, int(placement["cpus_effective"]))))
            el.device_free(d_cmd_l)
            el.close()
            kus_l = msl * 1e3 / max(nll, 1)
            t_l = None
            try:
                t_l = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(f"n{n}_b{Bl}_spl1")
            except (OSError, ValueError):
                pass
            bytes_l = 4 * (39 + 28 * n) * Bl
            secondary["large_batch"] = {
                "workload": f"{Bl} x {n}-cable robots on one GPU (config 4's whole batch: the HBM-streaming regime), every stage, one launch per step, one held Joy",
                "kernel_us": kus_l,
                "value_per_gpu": Bl / (kus_l * 1e-6),
                "value_wall": Bl * steps_l / wall_l,
                "unit": "state-steps/s",
                "steps_timed": steps_l,
                "bound": leg_bound("stream"),
                "frac": bytes_l / (kus_l * 1e-6) / 1e9 / HBM_PEAK_GBS,  # algorithmic bytes per launch / kernel time / 8 TB/s
                "traffic": t_l,                                           # HBM bytes per launch (rocprofv3 PMC of this round's layout)
                "traffic_frac": (t_l / (kus_l * 1e-6) / 1e9 / HBM_PEAK_GBS) if t_l else None,
                "traffic_frac_of_copy_rate": (t_l / (kus_l * 1e-6) / 1e9 / HBM_COPY_GBS) if t_l else None,
                "parity_check": lpar,
            }

        # (f) / (g) the other BASELINE configs that run on one GPU: config 2 in both launch forms, config 1
        #     with the host in the loop; each with its own roofline object and parity check
        if args.config == 3 and world == 1:
            secondary["config2"] = leg_config2(pkg, device, rank, placement, args.no_parity_check)
            secondary["config1"] = leg_config1(pkg, device, args.no_parity_check)

    if rank == 0:
        bytes_step = eng.bytes_per_state_step()
        launch_s = ev_ms * 1e-3 / max(launches, 1)
        robots_per_launch_steps = args.batch * args.steps / max(launches, 1)
        achieved_kernel = bytes_step * robots_per_launch_steps / launch_s / 1e9
        value = world * args.batch * args.steps / elapsed
        achieved = value / world * bytes_step / 1e9  # BASELINE.md section 3, per GPU
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                # (key by what ONE launch really is: the scheduled path's launches are SCHED_CHUNK steps long)
                tkey = f"n{n}_b{args.batch}_sched{SCHED_CHUNK}" if scheduled else f"n{n}_b{args.batch}_spl{args.steps_per_launch}"
                traffic = json.load(open(tpath)).get(tkey)
            except Exception:
                traffic = None
        residency = None
        try:
            res_tab = json.load(open(tpath)).get("_residency", {})
            r_ = res_tab.get(f"n{n}_b{args.batch}")
            if r_:
                state_mb = (eng.observable_image_bytes() + args.batch * 16 * (30 if n == 8 else 16)) / 1e6
                residency = {"tcc_hit_rate": r_["tcc_hit_rate"], "source": r_["file"], "working_set_MB": state_mb,
                             "note": "state and observables of the handle stay in L2 + Infinity Cache (256 MiB) between launches: `traffic` counts "
                                     "L2 <-> fabric bytes, not HBM row activations; see `large_batch` for the HBM-streaming regime"}
        except Exception:
            residency = None
        out = {
            "metric": METRIC,
            "value": value,
            "unit": "state-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{'config3' if n == 8 else 'config2'}: {args.batch} x {n}-cable robots per GPU, "
                            + ("IK + NR-FK(4 it) + tension distribution + PID + dynamics" if n == 8 else "IK + PID + dynamics")
                            + f", observables every step, commands ({'per-robot sines' if args.stimulus == 'sine' else args.stimulus + 'test, one amplitude per robot'}: "
                            f"{'jointPositions' if kind == 'position' else 'jointVelocities'}) refreshed every {refresh} steps from HBM",
                "robots_per_gpu": args.batch,
                "cables": n,
                "steps_per_launch": args.steps_per_launch,
                "launch_form": (f"cdpr_update_scheduled: one launch per {SCHED_CHUNK} steps, Joy batches read from the schedule inside the kernel" if scheduled
                                else "one launch per world step" if args.steps_per_launch == 1 else f"{args.steps_per_launch} steps per launch"),
                "mapping": eng.mapping,
                "state_finite": finite,
                "rendezvous": ctx.backend_name(),  # "none" (one rank), "socket" (default), or the opt-ins "nccl" (= RCCL) / "gloo": barrier + max only, no data-path collective
                "rendezvous_fallback": ctx.fallback,  # why the rendezvous is not on RCCL, if it is not (cdpr_simulation_amd/sharding.py)
            },
            "roofline": {
                # what limits the launch, and the tighter of the two roofs it is priced against (frozen in round 4: these
                # keys keep their meaning from here on)
                "bound": leg_bound("underfilled" if args.batch * (2 if eng.mapping == "lane-pair" else 1) < 65536 else "stream" if args.batch > 131072 else "chain"),
                "roof": "hbm",
                "frac_definition": "achieved / peak with achieved = algorithmic bytes per launch (SURVEY.md 8(d): 4*(39+28n) B per "
                                   "state-step x robots x steps of one launch) / the step kernel's average launch duration by HIP events "
                                   "on the engine's stream over the timed region; frac_wall = the same bytes by the wall clock "
                                   "(BASELINE.md section 3); traffic_frac = measured HBM bytes (rocprofv3 PMC) / the same duration",
                # algorithmic bytes per launch (4 (39 + 28 n) B per state-step x the robots and steps of one
                # launch) / the kernel's average launch duration, HIP events on the engine's stream over the timed region
                "achieved": achieved_kernel,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved_kernel / HBM_PEAK_GBS,
                "basis": "algorithmic bytes (SURVEY 8(d), shifted-window accounting) per launch / average launch duration by HIP events",
                "traffic": traffic,
                "traffic_model": modelled_traffic_bytes(n, n >= 6, args.batch, max(1, int(round(args.steps / max(launches, 1))))),  # from the data layout, for the steps one launch really runs
                "bytes_per_state_step": bytes_step,
                "kernel_us": launch_s * 1e6,  # one LAUNCH (the scheduled path: SCHED_CHUNK steps)
                "steps_per_launch": args.steps / max(launches, 1),
                "kernel_us_per_step": launch_s * 1e6 / (args.steps / max(launches, 1)),
                "achieved_kernel": achieved_kernel,  # (same as `achieved`; kept for readers of earlier rounds' lines)
                "frac_kernel": achieved_kernel / HBM_PEAK_GBS,
                # BASELINE.md section 3's formula: per-GPU state-steps/s by the WALL clock x the same algorithmic bytes
                # (the more conservative figure: it also carries the edges of the timed region)
                "achieved_wall": achieved,
                "frac_wall": achieved / HBM_PEAK_GBS,
                # the same launch priced by the bytes it really moved (rocprofv3 PMC, profiles/): the ring-buffer window
                # rewrites 6 controller rows per step where the contract's accounting assumes 24
                "traffic_GBps": (traffic / launch_s / 1e9) if traffic else None,
                "traffic_frac": (traffic / launch_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
                # ... and against what a plain copy kernel reaches on this part: how much of the practical ceiling the launch uses
                "traffic_frac_of_copy_rate": (traffic / launch_s / 1e9 / HBM_COPY_GBS) if traffic else None,
                "limiter": "instruction issue of the one wave that carries a robot's serial chain (5 cycles per vector instruction, DESIGN.md section 4), not HBM bandwidth",
                # where the bytes of this launch come from: a 65 536-robot handle's state + observables (~40 MB) fit the 256 MiB
                # Infinity Cache and partly L2, so consecutive launches are NOT streaming from HBM; the HBM-streaming figure is
                # the `large_batch` leg (524 288 robots on one GPU).  tcc_hit_rate: L2 hits / requests of the step kernel (PMC)
                "residency": residency,
            },
        }
        out["parity_check"] = parity
        out["placement"] = placement
        out["per_rank"] = [{"rank": i, "value": args.batch * args.steps / v[0], "ms_per_step": v[0] / args.steps * 1e3, "kernel_us": v[1],
                            "device": int(v[2]), "pci": idents[i]["pci"], "torch_pci": idents[i]["torch_pci"], "placement": {"numa_node": None if v[3] < 0 else int(v[3]), "first_cpu": None if v[4] < 0 else int(v[4]),
                                                                "cpus": int(v[5])}} for i, v in enumerate(per_rank)]
        out.update(secondary)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pkg, cfg_kwargs, pose, command, refresh, args.cpu_seconds, kind=kind)
        emit(json.dumps(out))
    for p in sched:
        eng.device_free(p)
    eng.close()
    ctx.close()
    # a number that is not tied to a verified computation is not reported as a success
    failed = [name for name, chk in (("step", parity), ("rollout", (secondary.get("rollout") or {}).get("parity_check")),
                                     ("general_path", (secondary.get("general_path") or {}).get("parity_check")),
                                     ("fp64", (secondary.get("fp64") or {}).get("parity_check")),
                                     ("large_batch", (secondary.get("large_batch") or {}).get("parity_check")),
                                     ("config2", (secondary.get("config2") or {}).get("parity_check")),
                                     ("config1", (secondary.get("config1") or {}).get("parity_check"))) if chk and not chk.get("ok_all_ranks", chk["ok"])]
    if failed:
        print(f"bench.py: parity check against the oracle FAILED for: {', '.join(failed)}", file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
