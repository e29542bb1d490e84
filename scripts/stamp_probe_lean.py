"""Diagnostic: phase timeline (s_memrealtime stamps per workgroup) of the lean general kernel (cdpr_gen_lean_kernel) beside the
fast path's cdpr_split_kernel at the same batch.  Build: make -C cdpr-simulation_amd/csrc stamps"""
import os, sys, ctypes as C
os.environ["CDPR_LIB"] = os.environ.get("STAMP_LIB", "libcdpr_hip_stamps.so"); os.environ["CDPR_MAPPING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cdpr_simulation_amd as pkg, bench
from cdpr_simulation_amd._native import lib
L = lib(); L.cdpr_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
B, n = int(os.environ.get("STAMP_B", "65536")), 8
for eps, label in ((0.001, "general path, lean kernel (steady)"), (0.004, "general path, lean kernel (cables switching Pids)"), (-0.001, "fast path, cdpr_split_kernel")):
    model, pose, command, n_cmd = bench.make_workload(pkg, B, n, 1235, 100)
    eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=3, velocityEpsilon=eps), 0)
    eng.set_platform_state(pose7=pose); eng.set_velocity_command(command(0)); eng.update(60); eng.synchronize()
    if eps == 0.004:
        for j in range(1, 6):
            eng.set_velocity_command(command(j)); eng.update(10)
        eng.set_velocity_command(command(6)); eng.update(3); eng.synchronize()
    buf = np.zeros((B // 64, 8), dtype=np.uint64)
    dptr = eng.device_upload(buf)
    L.cdpr_debug_set_stamps(eng._h, C.c_void_p(dptr))
    eng.update(5); eng.synchronize()
    L.cdpr_device_download(eng._h, buf.ctypes.data_as(C.c_void_p), C.c_void_p(dptr), buf.nbytes)
    t = buf.astype(np.float64) * 0.01  # us (100 MHz)
    t0 = t[:, 0].min()
    names = ["est: entry", "est: Newton done", "est: forces received", "est: TD done", "ctl: forces out", "ctl: tensions received", "ctl: end", "ctl: DMA landed, controller starts"]
    print(f"{label}, B={B}: span {t[:, 6].max() - t0:.2f} us; median / min / max over workgroups, us since the first entry")
    for i in (0, 7, 4, 1, 2, 3, 5, 6):
        col = t[:, i] - t0
        if t[:, i].max() > 0:
            print(f"  {i} {names[i]:36s} {np.median(col):7.2f} {col.min():7.2f} {col.max():7.2f}")
    eng.close()
