"""Diagnostic: the shader clock under the headline launch (cdpr_split_kernel<8> at 65 536 x 8) and at small batches - s_memtime ticks of
the estimator wave between its entry and the end of the tension distribution over the same interval by s_memrealtime (100 MHz).
Build: make -C cdpr-simulation_amd/csrc OUT=../libcdpr_probe.so OBJDIR=build_stamps_ctl EXTRA="-DCDPR_STAMPS -DCDPR_STAMPS_CLOCK" all"""
import os, sys, ctypes as C
os.environ["CDPR_LIB"] = os.environ.get("STAMP_LIB", "libcdpr_probe.so"); os.environ["CDPR_MAPPING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cdpr_simulation_amd as pkg, bench
from cdpr_simulation_amd._native import lib
L = lib(); L.cdpr_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
for B in (64, 4096, 16384, 65536):
    os.environ["CDPR_SPLIT"] = "1"
    model, pose, command, _ = bench.make_workload(pkg, B, 8, 1235, 10)
    eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=3), 0)
    eng.set_platform_state(pose7=pose); eng.set_velocity_command(command(0)); eng.update(300); eng.synchronize()
    buf = np.zeros(((B + 63) // 64, 8), dtype=np.uint64)
    dptr = eng.device_upload(buf)
    L.cdpr_debug_set_stamps(eng._h, C.c_void_p(dptr))
    out = []
    for rep in range(4):
        eng.update(200 if rep else 3); eng.synchronize()
        L.cdpr_device_download(eng._h, buf.ctypes.data_as(C.c_void_p), C.c_void_p(dptr), buf.nbytes)
        t = buf.astype(np.float64)
        us = (t[:, 3] - t[:, 0]) * 0.01
        mhz = t[:, 7] / np.maximum(us, 1e-9)
        out.append((np.median(mhz), np.median(us), (t[:, 6].max() - t[:, 0].min()) * 0.01))
    print(f"{eng.kernel_name} B={B}: shader clock median " + ", ".join(f"{m:.0f} MHz" for m, _, _ in out) + " (after 3, 200, 400, 600 more launches); estimator wave entry -> TD done " +
          ", ".join(f"{u:.2f}" for _, u, _ in out) + " us; span " + ", ".join(f"{s:.2f}" for _, _, s in out) + " us", flush=True)
    eng.close()

# ... and the ramp: after 0.5 s of idle, the clock and the span of the last of N launches
import time
B = 65536
model, pose, command, _ = bench.make_workload(pkg, B, 8, 1235, 10)
eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=3), 0)
eng.set_platform_state(pose7=pose); eng.set_velocity_command(command(0)); eng.update(300); eng.synchronize()
buf = np.zeros((B // 64, 8), dtype=np.uint64)
dptr = eng.device_upload(buf)
L.cdpr_debug_set_stamps(eng._h, C.c_void_p(dptr))
row = []
for nl in (1, 3, 5, 10, 23, 50, 100, 200, 500, 1000):
    time.sleep(0.5)
    eng.update(nl); eng.synchronize()
    L.cdpr_device_download(eng._h, buf.ctypes.data_as(C.c_void_p), C.c_void_p(dptr), buf.nbytes)
    t = buf.astype(np.float64)
    us = (t[:, 3] - t[:, 0]) * 0.01
    row.append(f"{nl}: {np.median(t[:, 7] / np.maximum(us, 1e-9)):.0f} MHz / {(t[:, 6].max() - t[:, 0].min()) * 0.01:.2f} us")
print("after 0.5 s idle, last of N launches (clock / span): " + " | ".join(row), flush=True)
eng.close()
