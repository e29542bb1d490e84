"""Diagnostic: where one launch of the one-step GENERAL-path kernel (cdpr_gen_step_kernel) spends its time: per-wave
s_memrealtime stamps.  Build: make -C cdpr-simulation_amd/csrc stamps"""
import os, sys, ctypes as C
os.environ["CDPR_LIB"] = os.environ.get("STAMP_LIB", "libcdpr_hip_stamps.so")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cdpr_simulation_amd as pkg, bench
from cdpr_simulation_amd._native import lib
L = lib(); L.cdpr_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
for B in (64, 16384, 65536):
    model, pose, command, n_cmd = bench.make_workload(pkg, B, 8, 1235, 10)
    eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=3, velocityEpsilon=0.001), 0)
    eng.set_platform_state(pose7=pose); eng.set_velocity_command(command(0)); eng.update(50); eng.synchronize()
    buf = np.zeros((B // 64, 8), dtype=np.uint64)
    dptr = eng.device_upload(buf)
    L.cdpr_debug_set_stamps(eng._h, C.c_void_p(dptr))
    eng.update(5); eng.synchronize()
    L.cdpr_device_download(eng._h, buf.ctypes.data_as(C.c_void_p), C.c_void_p(dptr), buf.nbytes)
    t = buf.astype(np.float64) * 0.01  # us (100 MHz)
    t0 = t[:, 0].min()
    names = ["entry", "commands in, DMA issued", "IK + early observables", "Newton FK done", "DMA landed", "controller done", "TD + limits + observables", "end"]
    print(f"B={B}: span {t[:, 7].max() - t0:.2f} us; median / min / max over waves, us since first entry")
    for i, nm in enumerate(names):
        col = t[:, i] - t0
        print(f"  {i} {nm:28s} {np.median(col):7.2f} {col.min():7.2f} {col.max():7.2f}")
    eng.close()
