"""Diagnostic: what the rare paths of the general controller cost per wave in the lean role-split kernel when cables keep
switching Pids (velocityEpsilon = 0.004, the bench's sines refreshed every 10 steps): per workgroup the time spent in the
per-cable loop, in the fit passes and the queue length, over several consecutive steps after a refresh.
Build: make -C cdpr-simulation_amd/csrc OUT=../libcdpr_hip_stamps.so OBJDIR=build_stamps EXTRA="-DCDPR_STAMPS -DCDPR_STAMPS_COLD" all"""
import os, sys, ctypes as C
os.environ["CDPR_LIB"] = os.environ.get("STAMP_LIB", "libcdpr_hip_stamps.so"); os.environ["CDPR_MAPPING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cdpr_simulation_amd as pkg, bench
from cdpr_simulation_amd._native import lib
L = lib(); L.cdpr_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
B, n = int(os.environ.get("STAMP_B", "65536")), 8
G = B // 64
model, pose, command, n_cmd = bench.make_workload(pkg, B, n, 1235, 200)
eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=3, velocityEpsilon=0.004), 0)
eng.set_platform_state(pose7=pose); eng.set_velocity_command(command(0)); eng.update(60); eng.synchronize()
for j in range(1, 6):
    eng.set_velocity_command(command(j)); eng.update(10)
buf = np.zeros((2 * G, 8), dtype=np.uint64)
dptr = eng.device_upload(buf)
L.cdpr_debug_set_stamps(eng._h, C.c_void_p(dptr))
print(f"B={B}: per step after a command refresh: workgroups in the cold path | queue length median / p90 / max | any rotation | "
      "us: loop median / max, fit passes median / max, finish median / max | span")
for step in range(12):
    if step % 10 == 0:
        eng.set_velocity_command(command(6 + step // 10))
    eng.device_upload_into(dptr, np.zeros_like(buf))
    eng.update(1); eng.synchronize()
    L.cdpr_device_download(eng._h, buf.ctypes.data_as(C.c_void_p), C.c_void_p(dptr), buf.nbytes)
    ph, cold = buf[:G], buf[G:]
    t = ph.astype(np.float64) * 0.01
    span = t[:, 6].max() - t[:, 0].min()
    took = cold[:, 0] > 0
    c = cold[took].astype(np.float64) * 0.01
    qlen = (cold[took, 4] & 0xFFFFFFFF).astype(np.int64)
    kind = ((cold[took, 4] >> 32) & 3).astype(np.int64)  # 0 / 1: general loop (1: some ring turned); 2 / 3: tier 1 (3: some Pid called after a gap)
    rot = (kind & 1)
    if took.sum() == 0:
        print(f"  step {step}: no workgroup in the cold path, span {span:.2f}"); continue
    loop = c[:, 1] - c[:, 0]
    hasq = cold[took, 2] > 0
    fit = np.where(hasq, c[:, 3] - c[:, 2], 0.0)
    fit_only = np.where(hasq & (cold[took, 7] > 0), c[:, 7] - c[:, 2], 0.0)  # window and stamps assembled (stamp 7 sits in front of the fit: its call is pure and sinks below the stamp's store)
    ctl = (t[took, 4] - t[took, 7])
    pre = c[:, 0] - t[took, 7]   # steady check + call entry
    e1 = c[:, 5] - t[took, 7]    # ... of which: up to the tail's first instruction behind its prologue
    e2 = c[:, 6] - c[:, 5]       # ... the Joy targets are back
    post = t[took, 4] - c[:, 3]  # return + registers back + forces to LDS
    ctl_s = (t[~took, 4] - t[~took, 7])
    print(f"  step {step:2d}: {took.sum():4d}/{G} (tier 1: {(kind >= 2).sum()}) | q {np.median(qlen):5.0f} {np.percentile(qlen, 90):5.0f} {qlen.max():5d} | rot {rot.sum():4d} | "
          f"loop {np.median(loop):5.2f} {loop.max():5.2f}  fit+finish {np.median(fit):5.2f} {fit.max():5.2f} (of which assembling the window and its stamps {np.median(fit_only):5.2f})  controller {np.median(ctl):5.2f} {ctl.max():5.2f} (entry {np.median(pre):5.2f} {pre.max():5.2f} [call {np.median(e1):4.2f}, targets {np.median(e2):4.2f}], exit {np.median(post):5.2f} {post.max():5.2f}; steady waves {np.median(ctl_s) if len(ctl_s) else 0:5.2f}) | span {span:.2f}")
    t0 = t[:, 0].min()
    print(f"        tier-1/2 workgroups: controller starts {np.median(t[took, 7] - t0):5.2f}, forces out {np.median(t[took, 4] - t0):5.2f} (max {(t[took, 4] - t0).max():5.2f}), "
          f"tensions back {np.median(t[took, 5] - t0):5.2f}, end {np.median(t[took, 6] - t0):5.2f} (max {(t[took, 6] - t0).max():5.2f}) | the others: "
          f"forces out {np.median(t[~took, 4] - t0):5.2f}, end {np.median(t[~took, 6] - t0):5.2f} (max {(t[~took, 6] - t0).max():5.2f})")
    if step in (1, 5):
        order = np.argsort(-ctl)[:5]
        for o in order:
            print(f"      worst: kind {kind[o]} q {qlen[o]:4d} rot {rot[o]} loop {loop[o]:.2f} fit {fit[o]:.2f} controller {ctl[o]:.2f}")
eng.close()
