"""Thin object wrapper over the C-ABI: one `Engine` = one `cdpr_handle_t` on one GPU."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _abi
from ._native import lib
from .config import Config


class CdprError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"cdpr error {code}: {message}")
        self.code = code


def _fp(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))


def derivative_weights(n: int, degree: int) -> np.ndarray:
    """End-point least-squares derivative weights (closed form of Pid::derive, Pid.cpp:193-247)."""
    w = np.empty(n, dtype=np.float64)
    rc = lib().cdpr_derivative_weights(n, degree, w.ctypes.data_as(C.POINTER(C.c_double)))
    if rc != _abi.OK:
        raise CdprError(rc, "bad (n, degree)")
    return w


def plan_kernel(config: Config, steps_per_launch: int = 1, flags: int = 0) -> str:
    """The kernel CDPR_MAP_AUTO would run a launch of `steps_per_launch` world steps of this configuration on (cdpr_plan_kernel:
    answered from the configuration alone, no GPU needed).  flags: _abi.PLAN_* bits.  Raises CdprError with the reason where
    cdpr_create would refuse the configuration."""
    buf = C.create_string_buffer(256)
    s = config.to_struct()
    rc = lib().cdpr_plan_kernel(C.byref(s), int(steps_per_launch), int(flags), buf, 256)
    if rc != _abi.OK:
        raise CdprError(rc, buf.value.decode())
    return buf.value.decode()


class Engine:
    """B independent robots advanced in lock step on one GPU."""

    def __init__(self, config: Config, device: int = 0):
        self.config = config
        self._cfg = config.to_struct()
        self.n = config.n_cables
        self.B = int(config.batch)
        self._h = C.c_void_p()
        self._rollout_pending = None
        rc = lib().cdpr_create(C.byref(self._cfg), int(device), C.byref(self._h))
        if rc != _abi.OK:
            msg = lib().cdpr_last_error(None).decode()
            self._h = C.c_void_p()
            if rc == _abi.ERR_INVALID:
                raise ValueError(msg)  # PLG.cpp:167-168 throws on a bad joint count
            raise CdprError(rc, msg)

    # -- lifecycle
    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().cdpr_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc: int) -> int:
        if rc < 0:
            raise CdprError(rc, lib().cdpr_last_error(self._h).decode())
        return rc

    def reset(self) -> None:
        self._check(lib().cdpr_reset(self._h))

    # -- inputs
    def set_platform_state(self, pose7=None, twist6=None) -> None:
        p = None if pose7 is None else np.ascontiguousarray(pose7, dtype=np.float32).reshape(self.B, 7)
        t = None if twist6 is None else np.ascontiguousarray(twist6, dtype=np.float32).reshape(self.B, 6)
        self._check(lib().cdpr_set_platform_state(self._h, _fp(p), _fp(t)))

    def _command(self, fn, fn_masked, axes, mask) -> int:
        a = np.ascontiguousarray(axes, dtype=np.float32).ravel()
        if mask is None:
            return self._check(fn(self._h, _fp(a), a.size))
        m = np.ascontiguousarray(mask, dtype=np.uint8).reshape(self.B)
        return self._check(fn_masked(self._h, _fp(a), a.size, m.ctypes.data_as(C.POINTER(C.c_uint8))))

    def set_velocity_command(self, axes, mask=None) -> int:
        """Joy on `jointVelocities`.  mask[B] (Config.perRobotCommands): only these robots receive it, the others keep
        their target, mode and Pid state (what independent plugin instances do, PLG.cpp:206-219)."""
        return self._command(lib().cdpr_set_velocity_command, lib().cdpr_set_velocity_command_masked, axes, mask)

    def set_position_command(self, axes, mask=None) -> int:
        return self._command(lib().cdpr_set_position_command, lib().cdpr_set_position_command_masked, axes, mask)

    def set_force_command(self, axes, mask=None) -> int:
        """JointForceCalculator::setForce on every joint (JFC.h:92-95): open-loop forces, held until another command of
        any kind arrives; no Pid runs (JFC.cpp:67-70).  Latched after a pending velocity / position command."""
        return self._command(lib().cdpr_set_force_command, lib().cdpr_set_force_command_masked, axes, mask)

    def set_force_command_device(self, dptr: int, count: int) -> int:
        return self._check(lib().cdpr_set_force_command_device(self._h, C.c_void_p(dptr), count))

    def bind_force_command_device(self, dptr: int, count: int) -> int:
        return self._check(lib().cdpr_bind_force_command_device(self._h, C.c_void_p(dptr), count))

    def bind_velocity_command_device(self, dptr: int, count: int) -> int:
        """Zero-copy: the device buffer float[B][n] at dptr is the latched Joy batch from the next update on; it must stay
        valid and unchanged until another velocity command has been latched."""
        return self._check(lib().cdpr_bind_velocity_command_device(self._h, C.c_void_p(dptr), count))

    def bind_position_command_device(self, dptr: int, count: int) -> int:
        return self._check(lib().cdpr_bind_position_command_device(self._h, C.c_void_p(dptr), count))

    def set_velocity_command_device(self, dptr: int, count: int) -> int:
        return self._check(lib().cdpr_set_velocity_command_device(self._h, C.c_void_p(dptr), count))

    def set_position_command_device(self, dptr: int, count: int) -> int:
        return self._check(lib().cdpr_set_position_command_device(self._h, C.c_void_p(dptr), count))

    # -- stepping
    def update(self, nsteps: int = 1, steps_per_launch: int = 1) -> None:
        if steps_per_launch == 1:
            self._check(lib().cdpr_update(self._h, int(nsteps)))
        else:
            self._check(lib().cdpr_update_fused(self._h, int(nsteps), int(steps_per_launch)))

    def observable_image_bytes(self) -> int:
        nbytes = C.c_size_t()
        self._check(lib().cdpr_observable_image_bytes(self._h, C.byref(nbytes)))
        return int(nbytes.value)

    def update_record_device(self, nsteps: int, steps_per_launch: int, d_record: int, record_bytes: int) -> None:
        """As update_record, into a caller-owned device buffer (nothing is copied back)."""
        self._check(lib().cdpr_update_record(self._h, int(nsteps), int(steps_per_launch), C.c_void_p(d_record), record_bytes))

    def update_scheduled(self, nsteps: int, refresh_steps: int, d_commands: int, d_record: int = 0, record_bytes: int = 0, d_ready: int = 0,
                         kind: str = "velocity", d_robot_masks: int = 0) -> None:
        """A whole command schedule resident in HBM, queued with one call: Joy batch j (device buffer float[batches][B][n] at
        d_commands) is latched at step j * refresh_steps as a `kind` command ("velocity" = jointVelocities, "position" =
        jointPositions, "force" = setForce); every step's observables go to the record (device buffer) if one is given.
        d_ready: optional device-visible uint32[batches] mailbox (batch j is taken once d_ready[j] != 0).  d_robot_masks
        (Config.perRobotCommands): device buffer uint8[batches][B], batch j reaches only the robots of its mask.
        Uniform-mode handles on the register-resident path run it in ONE launch; every other handle as a chain of launches
        queued back to back (same results: cdpr_update_scheduled_kind)."""
        k = {"velocity": _abi.COMMAND_VELOCITY, "position": _abi.COMMAND_POSITION, "force": _abi.COMMAND_FORCE}[kind]
        self._check(lib().cdpr_update_scheduled_kind(self._h, k, int(nsteps), int(refresh_steps), C.c_void_p(d_commands), C.c_void_p(d_ready) if d_ready else None,
                                                     C.c_void_p(d_robot_masks) if d_robot_masks else None, C.c_void_p(d_record) if d_record else None, int(record_bytes)))

    def update_record(self, nsteps: int, steps_per_launch: int = 10):
        """Advance nsteps world steps (fused launches) and return the observables of EVERY step:
        dict of arrays position/velocity/effort [nsteps, B, n], pose [nsteps, B, 7], twist [nsteps, B, 6]."""
        nbytes = C.c_size_t()
        self._check(lib().cdpr_observable_image_bytes(self._h, C.byref(nbytes)))
        image = int(nbytes.value)
        dptr = C.c_void_p()
        self._check(lib().cdpr_device_malloc(self._h, image * nsteps, C.byref(dptr)))
        try:
            self._check(lib().cdpr_update_record(self._h, int(nsteps), int(steps_per_launch), dptr, image * nsteps))
            raw = np.empty(image * nsteps, dtype=np.uint8)
            self._check(lib().cdpr_device_download(self._h, raw.ctypes.data_as(C.c_void_p), dptr, raw.nbytes))
        finally:
            lib().cdpr_device_free(self._h, dptr)
        f64 = int(self.config.precision) == 64  # precision = 64 handles record doubles and hand out float64 arrays
        out = {k: np.empty((nsteps, self.B, w), dtype=np.float64 if f64 else np.float32) for k, w in (("position", self.n), ("velocity", self.n), ("effort", self.n), ("pose", 7), ("twist", 6))}
        for j in range(nsteps):
            img = raw[j * image:(j + 1) * image]
            if f64:
                self._check(lib().cdpr_decode_observables_f64(self._h, img.ctypes.data_as(C.c_void_p), self._dp(out["position"][j]), self._dp(out["velocity"][j]),
                                                              self._dp(out["effort"][j]), self._dp(out["pose"][j]), self._dp(out["twist"][j])))
            else:
                self._check(lib().cdpr_decode_observables(self._h, img.ctypes.data_as(C.c_void_p), _fp(out["position"][j]), _fp(out["velocity"][j]),
                                                          _fp(out["effort"][j]), _fp(out["pose"][j]), _fp(out["twist"][j])))
        return out

    def decode_observables(self, image: np.ndarray):
        """One downloaded observable image (uint8[observable_image_bytes]) -> (position, velocity, effort, pose7, twist6)."""
        img = np.ascontiguousarray(image, dtype=np.uint8)
        out = [np.empty((self.B, w), dtype=np.float32) for w in (self.n, self.n, self.n, 7, 6)]
        self._check(lib().cdpr_decode_observables(self._h, img.ctypes.data_as(C.c_void_p), *[_fp(o) for o in out]))
        return tuple(out)

    def synchronize(self) -> None:
        self._check(lib().cdpr_synchronize(self._h))

    @property
    def step_count(self) -> int:
        return int(lib().cdpr_step_count(self._h))

    @property
    def kernel_name(self) -> str:
        """The kernel the last step launch of this handle ran on (cdpr_kernel_name)."""
        buf = C.create_string_buffer(256)
        self._check(lib().cdpr_kernel_name(self._h, buf, 256))
        return buf.value.decode()

    @property
    def mapping(self) -> str:
        return {_abi.MAP_LANE_PER_ROBOT: "lane-per-robot", _abi.MAP_LANE_PAIR: "lane-pair", _abi.MAP_LANE_PER_CABLE: "lane-per-cable"}.get(int(lib().cdpr_mapping(self._h)), "auto")

    @property
    def sim_time(self) -> float:
        return self.step_count * self.config.dt

    # -- observables
    def joint_states(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        q, qd, e = (np.empty((self.B, self.n), dtype=np.float32) for _ in range(3))
        self._check(lib().cdpr_get_joint_states(self._h, _fp(q), _fp(qd), _fp(e)))
        return q, qd, e

    def platform_state(self) -> Tuple[np.ndarray, np.ndarray]:
        p, t = np.empty((self.B, 7), dtype=np.float32), np.empty((self.B, 6), dtype=np.float32)
        self._check(lib().cdpr_get_platform_state(self._h, _fp(p), _fp(t)))
        return p, t

    def observables(self):
        """JointState + PlatformState arrays of the last published step in one device round trip (cdpr_get_observables):
        position, velocity, effort [B, n], pose [B, 7], twist [B, 6]."""
        q, qd, e = (np.empty((self.B, self.n), dtype=np.float32) for _ in range(3))
        p, t = np.empty((self.B, 7), dtype=np.float32), np.empty((self.B, 6), dtype=np.float32)
        self._check(lib().cdpr_get_observables(self._h, _fp(q), _fp(qd), _fp(e), _fp(p), _fp(t)))
        return q, qd, e, p, t

    # -- handles created with Config.precision = 64: the same read-outs in double
    @staticmethod
    def _dp(a: Optional[np.ndarray]):
        return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))

    def observables_f64(self):
        """position, velocity, effort [B, n], pose [B, 7], twist [B, 6] of the last published step, float64."""
        q, qd, e = (np.empty((self.B, self.n), dtype=np.float64) for _ in range(3))
        p, t = np.empty((self.B, 7), dtype=np.float64), np.empty((self.B, 6), dtype=np.float64)
        self._check(lib().cdpr_get_observables_f64(self._h, self._dp(q), self._dp(qd), self._dp(e), self._dp(p), self._dp(t)))
        return q, qd, e, p, t

    def raw_state_f64(self) -> Tuple[np.ndarray, np.ndarray]:
        p, t = np.empty((self.B, 7), dtype=np.float64), np.empty((self.B, 6), dtype=np.float64)
        self._check(lib().cdpr_get_raw_state_f64(self._h, self._dp(p), self._dp(t)))
        return p, t

    def set_platform_state_f64(self, pose7=None, twist6=None) -> None:
        p = None if pose7 is None else np.ascontiguousarray(pose7, dtype=np.float64).reshape(self.B, 7)
        t = None if twist6 is None else np.ascontiguousarray(twist6, dtype=np.float64).reshape(self.B, 6)
        self._check(lib().cdpr_set_platform_state_f64(self._h, self._dp(p), self._dp(t)))

    def raw_state(self) -> Tuple[np.ndarray, np.ndarray]:
        p, t = np.empty((self.B, 7), dtype=np.float32), np.empty((self.B, 6), dtype=np.float32)
        self._check(lib().cdpr_get_raw_state(self._h, _fp(p), _fp(t)))
        return p, t

    def pid_debug(self) -> np.ndarray:
        d = np.empty((self.B, _abi.PID_DEBUG_AXES), dtype=np.float32)
        self._check(lib().cdpr_get_pid_debug(self._h, _fp(d)))
        return d

    def fk_state(self):
        p, r, it = np.empty((self.B, 7), dtype=np.float32), np.empty(self.B, dtype=np.float32), np.empty(self.B, dtype=np.int32)
        self._check(lib().cdpr_get_fk_state(self._h, _fp(p), _fp(r), it.ctypes.data_as(C.POINTER(C.c_int32))))
        return p, r, it

    def td_state(self):
        t, f = np.empty((self.B, self.n), dtype=np.float32), np.empty(self.B, dtype=np.int32)
        self._check(lib().cdpr_get_td_state(self._h, _fp(t), f.ctypes.data_as(C.POINTER(C.c_int32))))
        return t, f

    def limit_state(self) -> np.ndarray:
        """Travel limits (Model.travel_lower / travel_upper; cube.sdf:436-437): uint32[B], bit i set where joint i's
        position was outside the range at the last published step."""
        m = np.empty(self.B, dtype=np.uint32)
        self._check(lib().cdpr_get_limit_state(self._h, m.ctypes.data_as(C.POINTER(C.c_uint32))))
        return m

    # -- one-shot batched kinematics on caller data
    def solve_ik(self, pose7, twist6=None):
        """Joint::Position / GetVelocity restated: q[B,n], qdot[B,n], J[B,n,6] for the given poses (and twists)."""
        p = np.ascontiguousarray(pose7, dtype=np.float32).reshape(self.B, 7)
        t = None if twist6 is None else np.ascontiguousarray(twist6, dtype=np.float32).reshape(self.B, 6)
        q, qd = np.empty((self.B, self.n), dtype=np.float32), np.empty((self.B, self.n), dtype=np.float32)
        jac = np.empty((self.B, self.n, 6), dtype=np.float32)
        self._check(lib().cdpr_solve_ik(self._h, _fp(p), _fp(t), _fp(q), _fp(qd), _fp(jac)))
        return q, qd, jac

    def solve_fk(self, lengths, seed7):
        """Newton-Raphson forward kinematics: pose7[B,7], residual[B], iterations[B]."""
        ln = np.ascontiguousarray(lengths, dtype=np.float32).reshape(self.B, self.n)
        sd = np.ascontiguousarray(seed7, dtype=np.float32).reshape(self.B, 7)
        pose, res, it = np.empty((self.B, 7), dtype=np.float32), np.empty(self.B, dtype=np.float32), np.empty(self.B, dtype=np.int32)
        self._check(lib().cdpr_solve_fk(self._h, _fp(ln), _fp(sd), _fp(pose), _fp(res), it.ctypes.data_as(C.POINTER(C.c_int32))))
        return pose, res, it

    def solve_td(self, pose7, wrench6):
        """Tension distribution for the wrench the cables must apply: tension[B,n], infeasible[B]."""
        p = np.ascontiguousarray(pose7, dtype=np.float32).reshape(self.B, 7)
        w = np.ascontiguousarray(wrench6, dtype=np.float32).reshape(self.B, 6)
        t, f = np.empty((self.B, self.n), dtype=np.float32), np.empty(self.B, dtype=np.int32)
        self._check(lib().cdpr_solve_td(self._h, _fp(p), _fp(w), _fp(t), f.ctypes.data_as(C.POINTER(C.c_int32))))
        return t, f

    # -- MPC fan-out
    def rollout_launch(self, commands, ref_position):
        """Queue one rollout and return at once (cdpr_rollout_velocity_launch): commands[B, H, S, n] as a host array
        (uploaded here, freed by rollout_fetch) or a (device_pointer, S, H) tuple for a buffer already in HBM."""
        if self._rollout_pending is not None:
            raise RuntimeError("rollout_launch: the previous rollout has not been fetched (call rollout_fetch first)")
        ref = np.ascontiguousarray(ref_position, dtype=np.float32).reshape(self.B, 3)
        if isinstance(commands, tuple):
            dptr, S, H = commands
            owned = None
        else:
            c = np.ascontiguousarray(commands, dtype=np.float32)
            assert c.ndim == 4 and c.shape[0] == self.B and c.shape[3] == self.n, "commands must be [B, H, S, n]"
            H, S = int(c.shape[1]), int(c.shape[2])
            dptr = owned = self.device_upload(c)
        try:
            self._check(lib().cdpr_rollout_velocity_launch(self._h, int(S), int(H), C.c_void_p(dptr), _fp(ref)))
        except Exception:
            if owned is not None:
                self.device_free(owned)
            raise
        self._rollout_pending = (int(S), owned)

    def rollout_fetch(self) -> np.ndarray:
        """cost[B, S] of the rollout queued by rollout_launch (synchronises the stream)."""
        if self._rollout_pending is None:
            raise RuntimeError("rollout_fetch: no rollout pending (call rollout_launch first)")
        S, owned = self._rollout_pending
        cost = np.empty((self.B, S), dtype=np.float32)
        try:
            self._check(lib().cdpr_rollout_velocity_fetch(self._h, _fp(cost)))
        finally:
            self._rollout_pending = None
            if owned is not None:
                self.device_free(owned)
        return cost

    def rollout_discard(self) -> None:
        """Drop a launched rollout without reading its costs (waits for it; frees an uploaded command buffer)."""
        if self._rollout_pending is None:
            return
        _, owned = self._rollout_pending
        self._rollout_pending = None
        try:
            self.synchronize()
        finally:
            if owned is not None:
                self.device_free(owned)

    def rollout_velocity(self, commands, ref_position) -> np.ndarray:
        """commands[B, H, S, n] (host array, or a (device_pointer, S, H) tuple for a buffer already in HBM),
        ref_position[B, 3] -> cost[B, S]; the engine's own state is left untouched."""
        self.rollout_launch(commands, ref_position)
        return self.rollout_fetch()

    def rollout_velocity_device(self, d_commands: int, samples: int, horizon: int, d_ref: int, d_cost: int) -> None:
        """Device-resident rollout: nothing copied, asynchronous on the engine's stream."""
        self._check(lib().cdpr_rollout_velocity_device(self._h, int(samples), int(horizon), C.c_void_p(d_commands), C.c_void_p(d_ref),
                                                       C.c_void_p(d_cost)))

    # -- caller-owned device buffers (e.g. a schedule of Joy batches resident in HBM)
    def device_upload(self, array: np.ndarray) -> int:
        a = np.ascontiguousarray(array)
        ptr = C.c_void_p()
        self._check(lib().cdpr_device_malloc(self._h, a.nbytes, C.byref(ptr)))
        self._check(lib().cdpr_device_upload(self._h, ptr, a.ctypes.data_as(C.c_void_p), a.nbytes))
        return int(ptr.value)

    def device_upload_into(self, dptr: int, array: np.ndarray) -> None:
        """Host array into an existing device buffer, on THIS handle's stream.  (Not the way to post the mailbox of a schedule that is
        already waiting: the copy may share the waiting launch's hardware queue - keep such words in mapped pinned host memory.)"""
        a = np.ascontiguousarray(array)
        self._check(lib().cdpr_device_upload(self._h, C.c_void_p(dptr), a.ctypes.data_as(C.c_void_p), a.nbytes))

    def device_alloc(self, nbytes: int) -> int:
        ptr = C.c_void_p()
        self._check(lib().cdpr_device_malloc(self._h, int(nbytes), C.byref(ptr)))
        return int(ptr.value)

    def device_download(self, dptr: int, shape, dtype=np.float32) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        self._check(lib().cdpr_device_download(self._h, out.ctypes.data_as(C.c_void_p), C.c_void_p(dptr), out.nbytes))
        return out

    def device_free(self, dptr: int) -> None:
        self._check(lib().cdpr_device_free(self._h, C.c_void_p(dptr)))

    # -- timing (HIP events on the engine's own stream)
    def profile_begin(self) -> None:
        self._check(lib().cdpr_profile_begin(self._h))

    def profile_end(self) -> Tuple[float, int]:
        ms, n = C.c_float(), C.c_uint64()
        self._check(lib().cdpr_profile_end(self._h, C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    def bytes_per_state_step(self) -> int:
        return int(lib().cdpr_bytes_per_state_step(C.byref(self._cfg)))
