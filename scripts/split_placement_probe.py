"""Where do the two waves of every cdpr_split_kernel workgroup run?  (diagnostic build: HW_ID / XCC_ID per wave)"""
import os, sys, ctypes as C
os.environ["CDPR_LIB"] = "libcdpr_hip_stamps.so"; os.environ["CDPR_MAPPING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from collections import Counter, defaultdict
import cdpr_simulation_amd as pkg, bench
from cdpr_simulation_amd._native import lib
L = lib(); L.cdpr_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
B, n = 65536, 8
model, pose, command, n_cmd = bench.make_workload(pkg, B, n, 1235, 10)
eng = pkg.Engine(pkg.Config(model=model, batch=B, stages=3), 0)
eng.set_platform_state(pose7=pose); eng.set_velocity_command(command(0)); eng.update(50); eng.synchronize()
nb = B // 64
buf = np.zeros((nb, 8), dtype=np.uint64)
dptr = eng.device_upload(buf)
L.cdpr_debug_set_stamps(eng._h, C.c_void_p(dptr))
for trial in range(3):
    eng.update(1); eng.synchronize()
    L.cdpr_device_download(eng._h, buf.ctypes.data_as(C.c_void_p), C.c_void_p(dptr), buf.nbytes)
    ids = buf[:, 7].copy().view(np.uint32).reshape(nb, 2)
    def dec(v): return dict(simd=(v >> 4) & 3, cu=(v >> 8) & 15, sh=(v >> 12) & 1, se=(v >> 13) & 7, xcc=(v >> 16) & 15)
    w0, w1 = dec(ids[:, 0]), dec(ids[:, 1])
    print("trial", trial, "SIMD pair (wave0, wave1) histogram:", Counter(zip(w0["simd"].tolist(), w1["simd"].tolist())).most_common(8))
    same_cu = (w0["cu"] == w1["cu"]) & (w0["se"] == w1["se"]) & (w0["sh"] == w1["sh"]) & (w0["xcc"] == w1["xcc"])
    print("  both waves on one CU:", int(same_cu.sum()), "of", nb)
    key = lambda w, i: (int(w["xcc"][i]), int(w["se"][i]), int(w["sh"][i]), int(w["cu"][i]))
    per_cu = defaultdict(list)
    for b in range(nb):
        per_cu[key(w0, b)].append((b, int(w0["simd"][b]), int(w1["simd"][b])))
    print("  CUs used:", len(per_cu), " workgroups per CU:", Counter(len(v) for v in per_cu.values()))
    for k in list(per_cu)[:4]:
        print("   CU", k, "-> (workgroup, simd of wave0, simd of wave1):", per_cu[k])
    # role collisions under a swap mask: count SIMDs hosting two estimators
    for mask in (0, 1, 8, 9, 0x100, 0x200, 0x300):
        simd_roles = defaultdict(list)
        for b in range(nb):
            swap = bin(b & mask).count("1") & 1
            est, ctl = (1, 0) if swap else (0, 1)   # physical wave index of the estimator / controller
            w = (w0, w1)
            simd_roles[key(w[est], b) + (int(w[est]["simd"][b]),)].append("E")
            simd_roles[key(w[ctl], b) + (int(w[ctl]["simd"][b]),)].append("C")
        c = Counter("".join(sorted(v)) for v in simd_roles.values())
        print(f"  mask {mask:#x}: SIMD occupancy by role:", dict(c))
