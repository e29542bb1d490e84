"""Diagnostic: phase timeline (s_memrealtime stamps, lane 0 of each wave, 16 per workgroup) of the role-split fp64 kernel
(cdpr_split_kernel_f64) at one robot and at full size, plain and with the hold branch live.
Build: make -C cdpr-simulation_amd/csrc OUT=../libcdpr_probe_f64.so OBJDIR=build_stamps EXTRA=-DCDPR_STAMPS all
       (and EXTRA="-DCDPR_STAMPS -DCDPR_STAMPS_WAIT" -> libcdpr_probe_f64w.so: every phase waits for the loads issued before it)"""
import os, sys, ctypes as C
os.environ["CDPR_LIB"] = os.environ.get("STAMP_LIB", "libcdpr_probe_f64.so")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cdpr_simulation_amd as pkg, bench
from cdpr_simulation_amd._native import lib
L = lib(); L.cdpr_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
NAMES = {0: "est: entry", 1: "est: rows requested (WAIT: landed)", 2: "est: measured lengths done", 3: "est: Newton + TD factor done", 4: "est: forces received (barrier 1)",
         5: "est: tensions out (barrier 2), exit", 8: "ctl: entry", 9: "ctl: rows requested (WAIT: landed)", 10: "ctl: IK + PID done, forces out", 11: "ctl: tensions received (barrier 2)",
         12: "ctl: limits, observables, world step done", 13: "ctl: stores issued", 14: "ctl: stores acknowledged"}
for B in [int(x) for x in os.environ.get("STAMP_B", "64,4096,65536").split(",")]:
    for eps, label in ((-1.0, "plain"), (0.001, "hold branch live"), (0.001, "hold branch live, a third of the cables held (bench.py's hold leg)")):
        model, pose, command, n_cmd = bench.make_workload(pkg, B, 8, 1235, 100)
        eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=3, precision=64, velocityEpsilon=eps), 0)
        cmd = command(0).copy()
        if "third" in label:
            held = (np.arange(B * 8).reshape(B, 8) % 3) == 0
            cmd[held] = 0.0
            cmd[~held & (np.abs(cmd) <= eps)] = 0.02
        eng.set_platform_state(pose7=pose); eng.set_velocity_command(cmd); eng.update(60); eng.synchronize()
        nwg = (B + 63) // 64
        buf = np.zeros((nwg, 16), dtype=np.uint64)
        dptr = eng.device_upload(buf)
        L.cdpr_debug_set_stamps(eng._h, C.c_void_p(dptr))
        rows = []
        for rep in range(5):
            eng.update(3); eng.synchronize()
            L.cdpr_device_download(eng._h, buf.ctypes.data_as(C.c_void_p), C.c_void_p(dptr), buf.nbytes)
            rows.append(buf.astype(np.float64) * 0.01)  # us (100 MHz)
        t = rows[-1]
        t0 = min(t[:, 0].min(), t[:, 8].min())
        print(f"{eng.kernel_name} {label}, B={B}: span {t[:, 14].max() - t0:.2f} us (5 launches: {', '.join('%.2f' % (x[:, 14].max() - min(x[:, 0].min(), x[:, 8].min())) for x in rows)}); "
              f"median / min / max over workgroups, us since the first entry", flush=True)
        raw = buf.astype(np.float64)
        mhz = (raw[:, 7] - raw[:, 6]) / np.maximum(t[:, 5] - t[:, 0], 1e-9)
        print(f"  shader clock over the estimator wave (s_memtime / s_memrealtime): median {np.median(mhz):.0f} MHz, min {mhz.min():.0f}, max {mhz.max():.0f}")
        for i in (0, 1, 2, 3, 4, 5, 8, 9, 10, 11, 12, 13, 14):
            col = t[:, i] - t0
            print(f"  {i:2d} {NAMES[i]:44s} {np.median(col):7.2f} {col.min():7.2f} {col.max():7.2f}")
        if nwg >= 64:  # who are the late ones?  percentiles of the workgroups' end, by XCD (workgroup id mod 8) and by launch order
            endt = t[:, 14] - t0
            print("     end of a workgroup, percentiles 10 / 50 / 90 / 99 / 100: " + " ".join("%.2f" % np.percentile(endt, q) for q in (10, 50, 90, 99, 100)))
            print("     median end by XCD: " + " ".join("%.2f" % np.median(endt[x::8]) for x in range(8)) + " | by quarter of the grid: " + " ".join("%.2f" % np.median(endt[k * nwg // 4:(k + 1) * nwg // 4]) for k in range(4)))
            fo = t[:, 10] - t0
            print("     ctl forces out, percentiles 10 / 50 / 90 / 99 / 100: " + " ".join("%.2f" % np.percentile(fo, q) for q in (10, 50, 90, 99, 100)) + " | by quarter of the grid: " + " ".join("%.2f" % np.median(fo[k * nwg // 4:(k + 1) * nwg // 4]) for k in range(4)))
        eng.close()
