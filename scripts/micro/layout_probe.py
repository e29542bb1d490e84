"""Where does a build variant's result first leave the shipped build's?  (the __builtin_expect hint.)

Drives a hold-branch handle (cables switching Pids) with several-steps-per-launch trajectory records, so that every world
step's effort / pose is kept, and writes everything to an .npz; run once per build (CDPR_LIB) and compare offline:

  CDPR_LIB=libcdpr_hip_var_expect.so python scripts/micro/layout_probe.py gpurun_out/probe_expect.npz
  python scripts/micro/layout_probe.py gpurun_out/probe_main.npz
  python scripts/micro/layout_probe.py --diff gpurun_out/probe_main.npz gpurun_out/probe_expect.npz"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def run(out):
    import cdpr_simulation_amd as pkg
    from variant_digest import hold_commands, poses

    os.environ["CDPR_GEN_SPLIT"] = "0"
    B, n, eps = int(os.environ.get("PROBE_B", "1500")), 8, 0.004
    rng = np.random.default_rng(41)
    model = pkg.eight_cable_model()
    eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=3, velocityEpsilon=eps), 0)
    eng.set_platform_state(pose7=poses(model, B, rng))
    eng.update(14)
    keep = {}
    step = 14
    for j, k in enumerate((3, 11, 17, 6, 25, 9, 12, 31, 10, 40, 22)):
        cmd = hold_commands(rng, B, n, eps)
        if j == 6:
            cmd = rng.uniform(-0.004, 0.004, (B, n)).astype(np.float32)
            eng.set_position_command(cmd)
        else:
            eng.set_velocity_command(cmd)
        spl = int(os.environ.get("PROBE_SPL", "4"))
        r = eng.update_record(k, spl)
        keep[f"cmd{j}"] = cmd
        keep[f"first{j}"] = np.int32(step)
        for key in ("effort", "pose", "velocity"):
            keep[f"{key}{j}"] = r[key]
        step += k
    np.savez_compressed(out, **keep)


def diff(a, b):
    A, Bz = np.load(a), np.load(b)
    eps = 0.004
    prev_cmd = None
    for j in range(11):
        ea, eb = A[f"effort{j}"], Bz[f"effort{j}"]
        pa, pb = A[f"pose{j}"], Bz[f"pose{j}"]
        cmd = A[f"cmd{j}"]
        bad = ea != eb  # [steps, B, n]
        print(f"round {j}: first step {int(A[f'first{j}'])}, {ea.shape[0]} steps; efforts differing per step: {bad.reshape(bad.shape[0], -1).sum(1).tolist()}; "
              f"poses differing per step: {(pa != pb).any(2).sum(1).tolist()}")
        if bad.any():
            s0 = int(np.argmax(bad.reshape(bad.shape[0], -1).any(1)))
            rob, cab = np.nonzero(bad[s0])
            print(f"   first deviating step {s0} of the round: {len(set(rob))} robots; cables histogram {np.bincount(cab, minlength=8).tolist()}; robots mod 64 {sorted(set(int(x) % 64 for x in rob))[:64]}")
            print(f"   waves touched: {sorted(set(int(x) // 64 for x in rob))}")
            hold_now = np.abs(cmd) <= eps
            print(f"   deviating (robot, cable) in the hold branch now: {int(hold_now[rob, cab].sum())} of {len(rob)}")
            if prev_cmd is not None and prev_cmd.shape == cmd.shape:
                hold_before = np.abs(prev_cmd) <= eps
                sw = hold_now != hold_before
                print(f"   ... that switched Pid with this command: {int(sw[rob, cab].sum())}; robots with any switching cable: {int(sw.any(1).sum())} of {cmd.shape[0]}; "
                      f"deviating robots with a switching cable: {int(sw.any(1)[sorted(set(rob))].sum())} of {len(set(rob))}")
            d = np.abs(ea[s0] - eb[s0])
            print(f"   max |effort diff| {d.max():.3e}; examples: {[(int(r_), int(c_), float(ea[s0, r_, c_]), float(eb[s0, r_, c_])) for r_, c_ in list(zip(rob, cab))[:6]]}")
            break
        prev_cmd = cmd


def run_stage_mix(out):
    """the n = 6 FK + TD case of variant_digest.py's stage_mix, one launch per step, everything readable kept per step"""
    import cdpr_simulation_amd as pkg
    from dataclasses import replace
    from variant_digest import poses

    os.environ["CDPR_MAPPING"] = "1"
    cables, B = int(os.environ.get("PROBE_N", "6")), 300
    full = pkg.eight_cable_model()
    model = replace(full, frame_anchors=full.frame_anchors[:cables], platform_anchors=full.platform_anchors[:cables])
    cfg = pkg.Config(model=model, batch=B, stages=3)
    rng = np.random.default_rng(20 + cables)
    eng = pkg.Engine(cfg, 0)
    eng.set_platform_state(pose7=poses(model, B, rng))
    keep = {k: [] for k in ("pose", "twist", "q", "qd", "eff", "raw_pose", "raw_twist", "fk_pose", "fk_res", "fk_it", "td_t", "td_flag")}

    def snap():
        p, t = eng.platform_state()
        q, qd, e = eng.joint_states()
        rp, rt = eng.raw_state()
        fp, fr, fi = eng.fk_state()
        tt, tf = eng.td_state()
        for k, v in zip(keep, (p, t, q, qd, e, rp, rt, fp, fr, fi, tt, tf)):
            keep[k].append(v)

    for _ in range(9):
        eng.update(1), snap()
    for j, k in enumerate((13, 31, 17, 40)):
        eng.set_velocity_command(rng.uniform(-0.03, 0.03, (B, cables)).astype(np.float32))
        for _ in range(k):
            eng.update(1), snap()
    np.savez_compressed(out, **{k: np.array(v) for k, v in keep.items()})


def diff_stage_mix(a, b):
    A, Bz = np.load(a), np.load(b)
    for k in A.files:
        bad = (A[k] != Bz[k]).reshape(A[k].shape[0], A[k].shape[1], -1).any(2)  # [steps, B]
        steps = np.nonzero(bad.any(1))[0]
        if len(steps):
            s0 = int(steps[0])
            print(f"{k}: first deviation after step {s0 + 1} (robots {np.nonzero(bad[s0])[0].tolist()[:12]}), deviating steps {len(steps)} of {bad.shape[0]}, "
                  f"max |diff| at that step {np.abs(A[k][s0].astype(np.float64) - Bz[k][s0].astype(np.float64)).max():.3e}")
        else:
            print(f"{k}: identical")


if __name__ == "__main__":
    if sys.argv[1] == "--diff":
        diff(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "--stage-mix":
        run_stage_mix(sys.argv[2])
    elif sys.argv[1] == "--diff-stage-mix":
        diff_stage_mix(sys.argv[2], sys.argv[3])
    else:
        run(sys.argv[1])
