import os, sys, ctypes as C
os.environ["CDPR_LIB"] = "libcdpr_hip_stamps.so"; os.environ["CDPR_MAPPING"] = "1"; os.environ["CDPR_PERSIST"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cdpr_simulation_amd as pkg, bench
from cdpr_simulation_amd._native import lib
L = lib(); L.cdpr_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
for B in (65536, 131072, 262144):
    model, pose, command, n_cmd = bench.make_workload(pkg, B, 8, 1235, 10)
    eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=3), 0)
    eng.set_platform_state(pose7=pose); eng.set_velocity_command(command(0)); eng.update(50); eng.synchronize()
    buf = np.zeros((1024, 8), dtype=np.uint64)
    dptr = eng.device_upload(buf)
    L.cdpr_debug_set_stamps(eng._h, C.c_void_p(dptr))
    eng.update(5); eng.synchronize()
    L.cdpr_device_download(eng._h, buf.ctypes.data_as(C.c_void_p), C.c_void_p(dptr), buf.nbytes)
    t = buf.astype(np.float64) * 0.01
    t0 = t[:, 0].min()
    names = ["entry", "block top done (last block)", "IK + early obs", "Newton done", "DMA landed", "PID done", "TD + obs done", "end"]
    print(f"B={B} ({B // 65536} blocks per wave): span {t[:, 7].max() - t0:.2f} us")
    for i, nm in enumerate(names):
        col = t[:, i] - t0
        print(f"  {i} {nm:28s} median {np.median(col):7.2f} min {col.min():7.2f} max {col.max():7.2f}")
    eng.close()
