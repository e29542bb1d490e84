"""Is a wave off the common path slow because its code is not in the instruction cache?  The same tier-1 work (one robot of 64
switches Pids: 8 queued cables) in EVERY wave of the launch against the bench's mix where ~3 waves of 1 024 do it.
Build: make -C cdpr-simulation_amd/csrc OUT=../libcdpr_hip_stamps.so OBJDIR=build_stamps EXTRA="-DCDPR_STAMPS -DCDPR_STAMPS_COLD" all"""
import os, sys, ctypes as C
os.environ["CDPR_LIB"] = os.environ.get("STAMP_LIB", "libcdpr_hip_stamps.so"); os.environ["CDPR_MAPPING"] = "1"; os.environ["CDPR_GEN_LEAN"] = "1"
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
eng.set_platform_state(pose7=pose)
hi = np.full((B, n), 0.01, dtype=np.float32)
lo = hi.copy(); lo[::64] = 0.0   # one robot per wave falls into the hold branch
eng.set_velocity_command(hi); eng.update(40)
for j in range(6):
    eng.set_velocity_command(lo if j % 2 == 0 else hi); eng.update(10)
eng.synchronize()
buf = np.zeros((2 * G, 8), dtype=np.uint64)
dptr = eng.device_upload(buf)
L.cdpr_debug_set_stamps(eng._h, C.c_void_p(dptr))
for step in range(12):
    if step % 10 == 0:
        eng.set_velocity_command(lo if (step // 10) % 2 == 0 else hi)
    eng.device_upload_into(dptr, np.zeros_like(buf))
    eng.update(1); eng.synchronize()
    L.cdpr_device_download(eng._h, buf.ctypes.data_as(C.c_void_p), C.c_void_p(dptr), buf.nbytes)
    ph, cold = buf[:G], buf[G:]
    t = ph.astype(np.float64) * 0.01
    took = cold[:, 0] > 0
    if took.sum() == 0:
        print(f"step {step}: no wave off the common path"); continue
    c = cold[took].astype(np.float64) * 0.01
    qlen = (cold[took, 4] & 0xFFFFFFFF).astype(np.int64)
    ctl = t[took, 4] - t[took, 7]
    entry = c[:, 0] - t[took, 7]
    build = c[:, 1] - c[:, 0]
    hasq = cold[took, 2] > 0
    fit = np.where(hasq, c[:, 3] - c[:, 2], 0.0)
    print(f"step {step:2d}: {took.sum():4d}/{G} waves in tier 1/2, q median {np.median(qlen):.0f} | controller {np.median(ctl):5.2f} (max {ctl.max():5.2f})  entry {np.median(entry):4.2f}  queue {np.median(build):4.2f}  fit + second pass {np.median(fit):4.2f} | span {t[:, 6].max() - t[:, 0].min():.2f}")
eng.close()
