"""One profiled workload per invocation (run under rocprofv3 by scripts/profile_kernels.sh):
  fused      65 536 x 8, all stages, 10 steps per launch into a trajectory record (cdpr_update_record)
  rollout    512 x 128 x 64 MPC rollout, device-resident reference and costs (kernel only)
  config2m1  4 096 x 4, lane-per-robot          config2m2  4 096 x 4, lane-pair
  lowreg     524 288 x 8, all stages, one launch per step (auto-selected low-register kernel)
  general / general65k        16 384 / 65 536 x 8 on the general controller path (velocityEpsilon = 0.001: hold branch live), one held Joy
  general_sw / general65k_sw  the same with epsilon = 0.004 and the bench's sines refreshed every 10 steps: cables keep switching Pids
  onestep    65 536 x 8, all stages, one launch per step (the headline kernel)
  perrobot   65 536 x 8, all stages, one launch per step on a per_robot_commands handle: a third of the robots in
             Position mode, the rest in Velocity mode with Pids reset at two different times (PR split kernel)
  perrobot_general  16 384 x 8 per-robot handle forced onto the general controller path (velocityEpsilon = 0)
  scan_<B>_<split|lowreg|auto>  <B> x 8, all stages, one launch per step, with the role-split kernel (CDPR_LOWREG=0), the
             low-register kernel (CDPR_LOWREG=1) or whatever CDPR_MAP_AUTO picks: the 65 536 < B < 200 000 range
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cdpr_simulation_amd as pkg
import bench

case = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
if case.startswith("scan_"):
    _, b_, kind = case.split("_")
    if kind != "auto":
        os.environ["CDPR_LOWREG"] = "1" if kind == "lowreg" else "0"
    B = int(b_)
    model, pose, command, _ = bench.make_workload(pkg, B, 8, 1235, 10)
    eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=3), 0)
    eng.set_platform_state(pose7=pose); eng.set_velocity_command(command(0)); eng.update(50); eng.synchronize()
    eng.update(reps * 2); eng.synchronize()
elif case in ("fused", "onestep", "lowreg"):
    B = 524288 if case == "lowreg" else 65536
    model, pose, command, _ = bench.make_workload(pkg, B, 8, 1235, 10)
    eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=3), 0)
    eng.set_platform_state(pose7=pose); eng.set_velocity_command(command(0)); eng.update(50); eng.synchronize()
    if case == "fused":
        image = eng.observable_image_bytes()
        d_rec = eng.device_alloc(image * 10)
        for _ in range(reps):
            eng.update_record_device(10, 10, d_rec, image * 10)
    else:
        eng.update(reps if case == "lowreg" else reps * 3)
    eng.synchronize()
elif case == "rollout":
    Br, S, H = bench.ROLLOUT_SHAPE
    model, pose, command, _ = bench.make_workload(pkg, Br, 8, 1235, 10)
    er = pkg.Engine(pkg.Config(model=model, batch=Br, stages=3), 0)
    er.set_platform_state(pose7=pose); er.update(20)
    dptr = er.device_upload(bench.make_rollout_commands(Br, H, S, 8))
    d_ref, d_cost = er.device_upload(pose[:, :3].copy()), er.device_alloc(Br * S * 4)
    for _ in range(max(reps // 5, 10)):
        er.rollout_velocity_device(dptr, S, H, d_ref, d_cost)
    er.synchronize()
elif case in ("config2m1", "config2m2"):
    os.environ["CDPR_MAPPING"] = case[-1]
    os.environ["CDPR_NO_GRAPH"] = "1"  # eager launches so every dispatch is a traced kernel of its own
    model, pose, command, _ = bench.make_workload(pkg, 4096, 4, 1234, 10)
    eng = pkg.Engine(pkg.Config(model=model, batch=4096), 0)
    eng.set_platform_state(pose7=pose); eng.set_velocity_command(command(0)); eng.update(50); eng.synchronize()
    eng.update(reps * 3); eng.synchronize()
elif case in ("perrobot", "perrobot_general"):
    B = 65536 if case == "perrobot" else 16384
    model, pose, command, _ = bench.make_workload(pkg, B, 8, 1235, 10)
    extra = {} if case == "perrobot" else {"velocityEpsilon": 0.0}
    eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=3, perRobotCommands=True, **extra), 0)
    eng.set_platform_state(pose7=pose)
    grp = np.arange(B) % 3
    eng.set_velocity_command(np.where(np.abs(command(0)) < 1e-3, 1e-3, command(0)).astype(np.float32), mask=grp >= 1); eng.update(30)
    eng.set_velocity_command(np.where(np.abs(command(1)) < 1e-3, 1e-3, command(1)).astype(np.float32), mask=grp == 2)
    eng.set_position_command(np.zeros((B, 8), np.float32), mask=grp == 0); eng.update(20); eng.synchronize()
    eng.update(reps * 3); eng.synchronize()
elif case in ("general", "general65k", "general_sw", "general65k_sw"):
    # general controller path (hold branch live).  plain: one Joy, held (round 3's case: every window stays a uniform grid);
    # _sw: the bench's per-robot sines refreshed every 10 steps with epsilon = 0.004, so cables keep crossing into and out
    # of the hold branch and their windows go through the fit
    B = 65536 if "65k" in case else 16384
    sw = case.endswith("_sw")
    model, pose, command, _ = bench.make_workload(pkg, B, 8, 1235, 4000)
    eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=3, velocityEpsilon=0.004 if sw else 0.001), 0)
    # (100 steps before anything is measured: on the lean kernel's handles a robot moves its steady state into the hot rows once
    #  its windows' runs are saturated, 63 consecutive calls - the steady workload is measured in that state)
    eng.set_platform_state(pose7=pose); eng.set_velocity_command(command(0)); eng.update(100); eng.synchronize()
    if sw:
        sched = [eng.device_upload(command(j)) for j in range(1, 1 + reps // 5 + 1)]
        for d in sched:
            eng.bind_velocity_command_device(d, B * 8); eng.update(10)
    else:
        eng.update(reps * 10)  # (long enough that the 62 launches before the hot rows take over weigh little in the per-kernel averages)
    eng.synchronize()
elif case in ("fp64", "fp64hold", "fp64hold1", "fp64hold3"):
    # precision = 64 at the contract's size: the role-split kernel's lean build; fp64hold: with the hold branch live (its HOLD
    # instantiation), one held Joy with a share of the cables at or below epsilon; fp64hold1: one robot (the LDS build); fp64hold3: every third cable held (bench.py's hold leg)
    B = 1 if case.endswith("1") else 65536
    model, pose, command, _ = bench.make_workload(pkg, B, 8, 1235, 4000)
    extra = {"velocityEpsilon": 0.001} if "hold" in case else {}
    eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=3, precision=64, **extra), 0)
    cmd = command(0).copy()
    if case == "fp64hold3":  # bench.py's hold leg: every third CABLE commanded 0 and held by its position Pid, the others on their velocity Pid
        held = (np.arange(B * 8).reshape(B, 8) % 3) == 0
        cmd[held] = 0.0
        cmd[~held & (np.abs(cmd) <= 0.001)] = 0.02
    eng.set_platform_state(pose7=pose); eng.set_velocity_command(cmd); eng.update(100); eng.synchronize()
    eng.update(reps * 5); eng.synchronize()
else:
    raise SystemExit(f"unknown case {case}")
print("done", case, flush=True)
