"""One-off confidence run for cdpr_pair_stream_kernel: random batch sizes, refresh periods, schedule lengths, ring start positions,
modes and cable counts; the scheduled / fused launch (steady-state kernel) against the general several-steps kernel
(CDPR_PAIR_STREAM=0) bit for bit.  Prints one line per case; exits non-zero on the first difference."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
os.environ["CDPR_MAPPING"] = "2"
import cdpr_simulation_amd as pkg

rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "1")))
SETTER = {"velocity": "set_velocity_command", "position": "set_position_command"}
bad = 0
for case in range(int(os.environ.get("FUZZ_CASES", "40"))):
    n = int(rng.choice([4, 8])); B = int(rng.integers(1, 700)); refresh = int(rng.integers(1, 23)); T = int(rng.integers(2, 160))
    start = int(rng.integers(12, 60)); kind = str(rng.choice(["velocity", "position"])); fused = bool(rng.integers(0, 2))
    model = pkg.eight_cable_model() if n == 8 else pkg.cube_model()
    cfg = pkg.Config(model=model, batch=B, stages=0)
    pose = np.tile(model.home_pose(), (B, 1)); pose[:, :3] += rng.uniform(-0.03, 0.03, (B, 3))
    nb = (T + refresh - 1) // refresh
    amp = 0.03 if kind == "velocity" else 0.003
    sched = rng.uniform(-amp, amp, (nb, B, n)).astype(np.float32)
    first = rng.uniform(-amp, amp, (B, n)).astype(np.float32)
    outs = []
    for stream in ("1", "0"):
        os.environ["CDPR_PAIR_STREAM"] = stream
        e = pkg.Engine(cfg, 0)
        e.set_platform_state(pose7=pose.astype(np.float32)); getattr(e, SETTER[kind])(first); e.update(start)
        if fused:
            for j in range(nb):
                getattr(e, SETTER[kind])(sched[j]); k = min(refresh, T - j * refresh); e.update(k, min(max(k, 1), 64))
        else:
            d = e.device_upload(sched); e.update_scheduled(T, refresh, d, kind=kind)
        name = e.kernel_name
        e.update(3)  # (the schedule's last batch stays latched: the buffer lives until the handle is closed)
        e.synchronize()
        outs.append([x.copy() for x in e.raw_state() + e.joint_states() + e.platform_state()] + [name])
        e.close()
    same = all(np.array_equal(x, y) for x, y in zip(outs[0][:-1], outs[1][:-1]))
    print(f"case {case}: n={n} B={B} refresh={refresh} T={T} start={start} {kind} {'fused' if fused else 'scheduled'}: {outs[0][-1]} vs {outs[1][-1]} -> {'same bits' if same else 'DIFFERENT'}", flush=True)
    bad += 0 if same else 1
sys.exit(1 if bad else 0)
