"""Diagnostic: where does one launch of the one-step kernel spend its time? (s_memrealtime stamps per wave)"""
import os, sys, ctypes as C
os.environ["CDPR_LIB"] = os.environ.get("STAMP_LIB", "libcdpr_hip_stamps.so"); os.environ["CDPR_MAPPING"] = os.environ.get("MAPSEL", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cdpr_simulation_amd as pkg, bench
from cdpr_simulation_amd._native import lib
L = lib(); L.cdpr_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
for stages in (3, 0):
    B, n = int(os.environ.get("STAMP_B", "65536")), 8
    model, pose, command, n_cmd = bench.make_workload(pkg, B, n, 1235, 10)
    eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=stages), 0)
    eng.set_platform_state(pose7=pose); eng.set_velocity_command(command(0)); eng.update(50); eng.synchronize()
    nb = B // (64 if os.environ.get("MAPSEL","1")=="1" else 32)
    buf = np.zeros((nb, 8), dtype=np.uint64)
    dptr = eng.device_upload(buf)
    L.cdpr_debug_set_stamps(eng._h, C.c_void_p(dptr))
    eng.update(5); eng.synchronize()
    L.cdpr_device_download(eng._h, buf.ctypes.data_as(C.c_void_p), C.c_void_p(dptr), buf.nbytes)
    t = buf.astype(np.float64) * 0.01  # us (100 MHz)
    t0 = t[:, 0].min()
    split = os.environ.get("CDPR_ONESTEP") != "1" and os.environ.get("CDPR_SPLIT") != "0" and stages == 3
    names = ["est: entry", "est: Newton done", "est: forces received", "est: TD done", "ctl: PID done, forces out", "ctl: tensions received", "ctl: end", "-"] if split else (["entry", "data arrived / IK", "PID", "FK", "TD", "obs store", "final store", "end"] if os.environ.get("CDPR_ONESTEP") == "1" else
             ["entry", "platform rows in, DMA issued", "IK + early obs", "Newton FK done", "DMA landed", "PID done", "TD + obs done", "end"])
    last = 6 if split else 7
    print(f"stages={stages}: kernel span {t[:,last].max()-t0:.2f} us; per-phase (median / min / max over waves), us since first wave entry:")
    for i, nm in enumerate(names):
        col = t[:, i] - t0
        print(f"  {i} {nm:20s} {np.median(col):7.2f} {col.min():7.2f} {col.max():7.2f}")
    eng.close()
