"""ctypes wrapper of the CPU fp64 oracle (oracle/libcdpr_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by the product package.  See oracle/cdpr_oracle.h for the
parity status of each part (several are "parity unpinned").
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcdpr_oracle.so")
REF_FILTER_PATH = os.path.join(_HERE, "_ref", "libref_filter.so")

DERIV_FAITHFUL = 0
DERIV_EXACT = 1
DERIV_FIR = 2

_lib = None


def build(force=False):
    """Compile the oracle (and oracle/_ref where /root/reference exists)."""
    srcs = [os.path.join(_HERE, f) for f in ("cdpr_oracle.c", "cdpr_oracle.h")] + [os.path.join(_HERE, "..", "include", "cdpr.h")]
    stale = force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if stale or (os.path.isdir("/root/reference") and not os.path.exists(REF_FILTER_PATH)):
        subprocess.run(["make", "-C", _HERE, "all"], check=True, capture_output=True)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        dp, fp, ip = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int32)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_void_p, C.c_int]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_reset.argtypes = [C.c_void_p]
        L.orc_set_platform_state.argtypes = [C.c_void_p, dp, dp]
        L.orc_set_velocity_command.argtypes = [C.c_void_p, fp, C.c_size_t]
        L.orc_set_position_command.argtypes = [C.c_void_p, fp, C.c_size_t]
        L.orc_set_velocity_command_masked.argtypes = [C.c_void_p, fp, C.c_size_t, C.POINTER(C.c_uint8)]
        L.orc_set_force_command.argtypes = [C.c_void_p, fp, C.c_size_t]
        L.orc_set_force_command_masked.argtypes = [C.c_void_p, fp, C.c_size_t, C.POINTER(C.c_uint8)]
        L.orc_set_position_command_masked.argtypes = [C.c_void_p, fp, C.c_size_t, C.POINTER(C.c_uint8)]
        L.orc_update.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_rollout_velocity.argtypes = [C.c_void_p, C.c_int, C.c_int, fp, dp, dp, C.c_int]
        L.orc_step_count.restype = C.c_uint64
        L.orc_step_count.argtypes = [C.c_void_p]
        L.orc_get_joint_states.argtypes = [C.c_void_p, dp, dp, dp]
        L.orc_get_platform_state.argtypes = [C.c_void_p, dp, dp]
        L.orc_get_raw_state.argtypes = [C.c_void_p, dp, dp]
        L.orc_get_pid_debug.argtypes = [C.c_void_p, dp]
        L.orc_get_fk_state.argtypes = [C.c_void_p, dp, dp, ip]
        L.orc_get_td_state.argtypes = [C.c_void_p, dp, ip]
        L.orc_get_limit_state.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        L.orc_ik.argtypes = [C.c_void_p, dp, dp, dp, dp, dp, dp]
        L.orc_fk.restype = C.c_int
        L.orc_fk.argtypes = [C.c_void_p, dp, dp, dp, dp]
        L.orc_td_forces.restype = C.c_int
        L.orc_td_forces.argtypes = [C.c_void_p, dp, dp, dp]
        L.orc_td_wrench.restype = C.c_int
        L.orc_td_wrench.argtypes = [C.c_void_p, dp, dp, dp]
        L.orc_pid_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_pid_reset.argtypes = [C.c_void_p]
        L.orc_pid_update.restype = C.c_double
        L.orc_pid_update.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
        L.orc_biquad_set_fc.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
        L.orc_biquad_set_value.argtypes = [C.c_void_p, C.c_double]
        L.orc_biquad_process.restype = C.c_double
        L.orc_biquad_process.argtypes = [C.c_void_p, C.c_double]
        L.orc_colpiv_qr_solve.argtypes = [C.c_int, dp, dp, dp]
        L.orc_max_threads.restype = C.c_int
        _lib = L
    return _lib


_threads_default = None


def _default_threads():
    """CPUs this process may use: the affinity mask capped by the cgroup CPU quota."""
    global _threads_default
    if _threads_default is None:
        n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        try:
            quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if quota != "max":
                n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
        except Exception:
            try:
                q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
                p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / p + 0.5)))
            except Exception:
                pass
        _threads_default = max(1, min(n, lib().orc_max_threads()))
    return _threads_default


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _as_f64(a, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


class OracleSim:
    """Batched fp64 simulator: the checker for the HIP engine."""

    def __init__(self, cfg_struct, deriv_mode=DERIV_EXACT):
        self._cfg = cfg_struct  # keep alive
        self.n = int(cfg_struct.n_cables)
        self.B = int(cfg_struct.batch)
        self._h = lib().orc_create(C.byref(cfg_struct), deriv_mode)
        if not self._h:
            raise ValueError("orc_create failed (bad configuration)")

    def close(self):
        if self._h:
            lib().orc_destroy(self._h)
            self._h = None

    __del__ = close

    def reset(self):
        lib().orc_reset(self._h)

    def set_platform_state(self, pose7=None, twist6=None):
        p = _as_f64(pose7, (self.B, 7))
        t = _as_f64(twist6, (self.B, 6))
        lib().orc_set_platform_state(self._h, _dp(p), _dp(t))

    def _cmd(self, fn, fn_masked, axes, mask):
        a = np.ascontiguousarray(axes, dtype=np.float32).ravel()
        if mask is None:
            return fn(self._h, a.ctypes.data_as(C.POINTER(C.c_float)), a.size)
        m = np.ascontiguousarray(mask, dtype=np.uint8).reshape(self.B)
        return fn_masked(self._h, a.ctypes.data_as(C.POINTER(C.c_float)), a.size, m.ctypes.data_as(C.POINTER(C.c_uint8)))

    def set_velocity_command(self, axes, mask=None):
        """mask[B]: only these robots receive the Joy (per-robot arrival); None = every robot."""
        return self._cmd(lib().orc_set_velocity_command, lib().orc_set_velocity_command_masked, axes, mask)

    def set_position_command(self, axes, mask=None):
        return self._cmd(lib().orc_set_position_command, lib().orc_set_position_command_masked, axes, mask)

    def set_force_command(self, axes, mask=None):
        """JointForceCalculator::setForce on every joint (JFC.h:92-95)."""
        return self._cmd(lib().orc_set_force_command, lib().orc_set_force_command_masked, axes, mask)

    def _threads(self, nthreads, units):
        """OpenMP threads for a call over `units` independent robots / trajectories when the caller names none: never more
        than the CPUs this process may really use (on the GPU box `nproc` says 256 while the cgroup grants 16: a parallel
        region of 256 threads on 16 CPUs costs ~100 ms per call) and not more than a thread per 16 units."""
        if nthreads and nthreads > 0:
            return int(nthreads)
        return max(1, min(_default_threads(), (int(units) + 15) // 16))

    def update(self, nsteps=1, nthreads=0):
        return lib().orc_update(self._h, int(nsteps), self._threads(nthreads, self.B))

    def rollout_velocity(self, commands, ref_position, nthreads=0):
        c = np.ascontiguousarray(commands, dtype=np.float32)
        H, S = int(c.shape[1]), int(c.shape[2])
        ref = np.ascontiguousarray(ref_position, dtype=np.float64).reshape(self.B, 3)
        cost = np.empty((self.B, S))
        lib().orc_rollout_velocity(self._h, S, H, c.ctypes.data_as(C.POINTER(C.c_float)), _dp(ref), _dp(cost), self._threads(nthreads, self.B * S))
        return cost

    @property
    def step_count(self):
        return int(lib().orc_step_count(self._h))

    def joint_states(self):
        q, qd, e = (np.empty((self.B, self.n)) for _ in range(3))
        lib().orc_get_joint_states(self._h, _dp(q), _dp(qd), _dp(e))
        return q, qd, e

    def platform_state(self):
        p, t = np.empty((self.B, 7)), np.empty((self.B, 6))
        lib().orc_get_platform_state(self._h, _dp(p), _dp(t))
        return p, t

    def raw_state(self):
        p, t = np.empty((self.B, 7)), np.empty((self.B, 6))
        lib().orc_get_raw_state(self._h, _dp(p), _dp(t))
        return p, t

    def pid_debug(self):
        d = np.empty((self.B, 9))
        lib().orc_get_pid_debug(self._h, _dp(d))
        return d

    def fk_state(self):
        p, r, it = np.empty((self.B, 7)), np.empty(self.B), np.empty(self.B, dtype=np.int32)
        lib().orc_get_fk_state(self._h, _dp(p), _dp(r), it.ctypes.data_as(C.POINTER(C.c_int32)))
        return p, r, it

    def td_state(self):
        t, f = np.empty((self.B, self.n)), np.empty(self.B, dtype=np.int32)
        lib().orc_get_td_state(self._h, _dp(t), f.ctypes.data_as(C.POINTER(C.c_int32)))
        return t, f

    def limit_state(self):
        m = np.empty(self.B, dtype=np.uint32)
        lib().orc_get_limit_state(self._h, m.ctypes.data_as(C.POINTER(C.c_uint32)))
        return m


def ik(cfg_struct, pose7, twist6=None):
    n = int(cfg_struct.n_cables)
    pose = _as_f64(pose7, (7,))
    tw = _as_f64(np.zeros(6) if twist6 is None else twist6, (6,))
    q, qd, ln, jac = np.empty(n), np.empty(n), np.empty(n), np.empty((n, 6))
    lib().orc_ik(C.byref(cfg_struct), _dp(pose), _dp(tw), _dp(q), _dp(qd), _dp(ln), _dp(jac))
    return q, qd, ln, jac


def fk(cfg_struct, lengths, seed7):
    out, res = np.empty(7), np.empty(1)
    it = lib().orc_fk(C.byref(cfg_struct), _dp(_as_f64(lengths)), _dp(_as_f64(seed7, (7,))), _dp(out), _dp(res))
    return out, float(res[0]), int(it)


def td_forces(cfg_struct, pose7, forces):
    t = np.empty(int(cfg_struct.n_cables))
    flag = lib().orc_td_forces(C.byref(cfg_struct), _dp(_as_f64(pose7, (7,))), _dp(_as_f64(forces)), _dp(t))
    return t, int(flag)


def td_wrench(cfg_struct, pose7, wrench6):
    t = np.empty(int(cfg_struct.n_cables))
    flag = lib().orc_td_wrench(C.byref(cfg_struct), _dp(_as_f64(pose7, (7,))), _dp(_as_f64(wrench6, (6,))), _dp(t))
    return t, int(flag)


class OraclePid:
    """One Pid instance (Pid.cpp) for unit-level known-answer tests."""

    _SIZE = 4096  # >= sizeof(orc_pid)

    def __init__(self, pid_params_struct, deriv_mode=DERIV_EXACT):
        self._buf = C.create_string_buffer(self._SIZE)
        self._prm = pid_params_struct
        lib().orc_pid_init(self._buf, C.byref(pid_params_struct), deriv_mode)

    def reset(self):
        lib().orc_pid_reset(self._buf)

    def update(self, desired, actual, now):
        return lib().orc_pid_update(self._buf, desired, actual, now)


class OracleBiquad:
    def __init__(self, fc, fs, q):
        self._buf = C.create_string_buffer(128)
        lib().orc_biquad_set_value(self._buf, 0.0)
        lib().orc_biquad_set_fc(self._buf, fc, fs, q)

    def set_value(self, v):
        lib().orc_biquad_set_value(self._buf, v)

    def process(self, x):
        return lib().orc_biquad_process(self._buf, x)


class RefBiquad:
    """The reference's own Filter.h BiQuad<double> (oracle/_ref, built where /root/reference exists)."""

    def __init__(self, fc, fs, q):
        L = C.CDLL(REF_FILTER_PATH)
        L.ref_biquad_new.restype = C.c_void_p
        L.ref_biquad_new.argtypes = [C.c_double] * 3
        L.ref_biquad_process.restype = C.c_double
        L.ref_biquad_process.argtypes = [C.c_void_p, C.c_double]
        L.ref_biquad_set_value.argtypes = [C.c_void_p, C.c_double]
        L.ref_biquad_free.argtypes = [C.c_void_p]
        self._L = L
        self._p = L.ref_biquad_new(fc, fs, q)

    def set_value(self, v):
        self._L.ref_biquad_set_value(self._p, v)

    def process(self, x):
        return self._L.ref_biquad_process(self._p, x)

    def __del__(self):
        if getattr(self, "_p", None):
            self._L.ref_biquad_free(self._p)
            self._p = None


def qr_solve(a, b):
    a = np.array(a, dtype=np.float64, order="C")
    b = np.array(b, dtype=np.float64)
    x = np.empty_like(b)
    lib().orc_colpiv_qr_solve(a.shape[0], _dp(a), _dp(b), _dp(x))
    return x
