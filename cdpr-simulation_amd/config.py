"""Configuration of the batched CDPR step engine.

`Config` carries what `CdprGazeboPlugin::Load` reads from the ROS parameter server
(CdprGazeboPlugin.h:32-54, CdprGazeboPlugin.cpp:57,102-138; shipped values in
launch/cdpr_gazebo.launch:17-39) plus the model constants Gazebo takes from
sdf/cube.sdf and sdf/cube.yaml.  Key names follow the launch file so a parameter
dictionary dumped from the ROS parameter server loads unchanged.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field, replace
from typing import Dict, List, Mapping, Sequence

import numpy as np

from . import _abi

LAUNCH_PARAM_PREFIX = "/cdpr_gazebo_simulator/"  # CdprGazeboPlugin.h:32-54


@dataclass
class FilterParameters:  # Pid.h:64-68
    relCutoff: float = 0.1
    quality: float = 0.707
    cascade: int = 0


@dataclass
class PidParameters:  # Pid.h:70-81
    forwardGain: float = 0.0
    pGain: float = 0.0
    iGain: float = 0.0
    dGain: float = 0.0
    dDegree: int = 2
    dBufferLength: int = 11
    iLimit: float = 100.0
    cmdLimit: float = 100.0
    pFilter: FilterParameters = field(default_factory=FilterParameters)
    dFilter: FilterParameters = field(default_factory=FilterParameters)


def _shipped_velocity_pid() -> PidParameters:  # launch:19-32
    return PidParameters(0.0, 200.0, 20.0, 1.0, 2, 11, 100.0, 100.0, FilterParameters(0.1, 0.707, 0), FilterParameters(0.1, 0.707, 0))


def _shipped_position_pid() -> PidParameters:  # launch:33-39; forward gain / cascades forced 0 at PLG.cpp:123,133
    return PidParameters(0.0, 200.0, 70.0, 80.0, 2, 11, 100.0, 100.0, FilterParameters(0.1, 0.707, 0), FilterParameters(0.1, 0.707, 0))


@dataclass
class Model:
    """Geometry and inertial constants of one CDPR (what cube.sdf / cube.yaml hold)."""

    frame_anchors: np.ndarray  # [n,3] a_i in frame coords   (cube.yaml:21-29 `frame`)
    platform_anchors: np.ndarray  # [n,3] b_i in platform coords (cube.yaml:21-29 `platform`)
    home_position: Sequence[float] = (0.0, 0.0, 0.3)  # cube.sdf:310 (cube.yaml:17 says z=2; the SDF is what is loaded)
    home_quaternion: Sequence[float] = (0.0, 0.0, 0.0, 1.0)  # x y z w
    mass: float = 1.0  # cube.sdf:340
    inertia: Sequence[float] = (1.0, 1.0, 1.0, 0.0, 0.0, 0.0)  # ixx iyy izz ixy ixz iyz, cube.sdf:332-339
    joint_damping: float = 1.0  # cube.sdf:442 / cube.yaml:9
    effort_limit: float = 100.0  # cube.sdf:438 / cube.yaml:9
    velocity_limit: float = -1.0  # cube.sdf:439 has 10; < 0 = not modelled (the contract's reduced model)
    unilateral_cables: bool = False  # [NEW] option: cables cannot push
    # lumped terms for what the massless-cable reduction drops of the 22-link SDF model (all 0: the reduced model)
    passive_damping: float = 0.0    # every passive revolute joint of a leg, cube.sdf:396,425,471,500,515 (0.01)
    leg_inertia: float = 0.0        # virt_X + virt_Y + cable link + virt_Ypf turning about the frame anchor (4 x 0.001)
    cable_axial_mass: float = 0.0   # cable link sliding along the cable axis, cube.sdf:368 (0.001)
    anchor_point_mass: float = 0.0  # virt_Xpf + virt_Ypf at each platform anchor (2 x 0.001)
    anchor_inertia: float = 0.0     # virt_Xpf turning with the platform (0.001)
    # travel limits of the prismatic joints (cube.sdf:436-437: -0.5196 / 0.5196); (0, 0) = none.  travel_stop > 0: model
    # the joint stop itself (inelastic; that many Gauss-Seidel sweeps over the cables per step, 4 hold several joints on
    # their stops at once), not only the flag (Engine.limit_state)
    travel_lower: float = 0.0
    travel_upper: float = 0.0
    travel_stop: int = 0
    f_min: float = 5.0  # cube.yaml:9 `min`
    f_max: float = 100.0  # cube.yaml:9 `effort`

    @property
    def n_cables(self) -> int:
        return int(np.asarray(self.frame_anchors).shape[0])

    def home_pose(self) -> np.ndarray:
        return np.array(list(self.home_position) + list(self.home_quaternion), dtype=np.float64)

    def reference_lengths(self) -> np.ndarray:
        """L0_i: cable length at the spawn pose, where every prismatic joint reads 0
        (gen_cdpr.py:113-118: pp = pf_t + pf_R b, u = (pp - fp)/|pp - fp|)."""
        r = quat_to_matrix(self.home_quaternion)
        pp = np.asarray(self.home_position, dtype=np.float64)[None, :] + np.asarray(self.platform_anchors, dtype=np.float64) @ r.T
        return np.linalg.norm(pp - np.asarray(self.frame_anchors, dtype=np.float64), axis=1)


def quat_to_matrix(q: Sequence[float]) -> np.ndarray:
    x, y, z, w = [float(v) for v in q]
    return np.array(
        [
            [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
            [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
            [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
        ]
    )


def cube_model() -> Model:
    """The shipped 4-cable robot: anchors from cube.yaml:21-29, spawn pose from cube.sdf:310."""
    fa = np.array([[-0.3, -0.3, 0.6], [-0.3, 0.3, 0.6], [0.3, 0.3, 0.6], [0.3, -0.3, 0.6]])
    pa = np.array([[-0.03, -0.03, 0.0], [-0.03, 0.03, 0.0], [0.03, 0.03, 0.0], [0.03, -0.03, 0.0]])
    return Model(fa, pa)


def eight_cable_model() -> Model:
    """Build-defined 8-cable robot (the reference has none).

    Frame anchors are the corners of the frame box (cube.yaml:4-7), platform anchors the corners
    of the platform box (cube.yaml:18, 0.06 x 0.06 x 0.015).  Top cable k goes to platform corner
    c[pi_t(k)] on the BOTTOM face, bottom cable k to c[pi_b(k)] on the TOP face, with
    pi_t = (3,1,2,0), pi_b = (0,1,3,2); this crossing gives rank(J) = 6 and wrench closure.
    """
    corners = [(-1, -1), (-1, 1), (1, 1), (1, -1)]
    pi_t, pi_b = (3, 1, 2, 0), (0, 1, 3, 2)
    fa, pa = [], []
    for k, (sx, sy) in enumerate(corners):  # top cables
        fa.append([0.3 * sx, 0.3 * sy, 0.6])
        cx, cy = corners[pi_t[k]]
        pa.append([0.03 * cx, 0.03 * cy, -0.0075])
    for k, (sx, sy) in enumerate(corners):  # bottom cables
        fa.append([0.3 * sx, 0.3 * sy, 0.0])
        cx, cy = corners[pi_b[k]]
        pa.append([0.03 * cx, 0.03 * cy, 0.0075])
    return Model(np.array(fa), np.array(pa))


def twelve_cable_model() -> Model:
    """Build-defined 12-cable robot (round 6: cube.yaml:21-29 is a free-length `points` list; the engine takes up to 12):
    the eight cables of eight_cable_model plus four from the middle of the frame's vertical edges' height (z = 0.3) at the
    mid-points of its sides to the platform's side mid-points - redundant cables in the horizontal plane, rank(J) = 6."""
    m8 = eight_cable_model()
    fa, pa = list(m8.frame_anchors), list(m8.platform_anchors)
    for sx, sy in ((1, 0), (0, 1), (-1, 0), (0, -1)):
        fa.append([0.3 * sx, 0.3 * sy, 0.3])
        pa.append([0.03 * sx, 0.03 * sy, 0.0])
    return Model(np.array(fa), np.array(pa))


@dataclass
class Config:
    model: Model = field(default_factory=cube_model)
    batch: int = 1
    dt: float = 1e-3  # Gazebo default max_step_size
    gravity: Sequence[float] = (0.0, 0.0, -9.8)  # Gazebo default
    publishPeriod: float = 0.0  # launch:17
    velocityEpsilon: float = -0.001  # launch:18
    velocityController: PidParameters = field(default_factory=_shipped_velocity_pid)
    positionController: PidParameters = field(default_factory=_shipped_position_pid)
    stages: int = 0
    mapping: int = _abi.MAP_AUTO
    fkMaxIterations: int = 4
    fkLambda: float = 1e-9
    fkTolerance: float = 0.0
    precision: int = 32  # 64: the step in the reference's own precision (fp64 kernels: every option of the controller and the physics; Engine.observables_f64)
    perRobotCommands: bool = False  # every robot has its own mode / Pid history: a Joy may reach some robots only ([NEW]: B plugin instances)
    tdFMin: float | None = None  # default: model.f_min
    tdFMax: float | None = None

    # --- ROS parameter spellings (CdprGazeboPlugin.h:32-54) -> attribute paths
    _LAUNCH_KEYS = {
        "publishPeriod": ("publishPeriod",),
        "velocityEpsilon": ("velocityEpsilon",),
        "velocityControllerForward": ("velocityController", "forwardGain"),
        "velocityControllerP": ("velocityController", "pGain"),
        "velocityControllerI": ("velocityController", "iGain"),
        "velocityControllerD": ("velocityController", "dGain"),
        "velocityControllerDdegree": ("velocityController", "dDegree"),
        "velocityControllerDbuffer": ("velocityController", "dBufferLength"),
        "velocityControllerMaxI": ("velocityController", "iLimit"),
        "velocityControllerMaxCmd": ("velocityController", "cmdLimit"),
        "velocityControllerPcutoff": ("velocityController", "pFilter", "relCutoff"),
        "velocityControllerPquality": ("velocityController", "pFilter", "quality"),
        "velocityControllerPcascade": ("velocityController", "pFilter", "cascade"),
        "velocityControllerDcutoff": ("velocityController", "dFilter", "relCutoff"),
        "velocityControllerDquality": ("velocityController", "dFilter", "quality"),
        "velocityControllerDcascade": ("velocityController", "dFilter", "cascade"),
        "positionControllerP": ("positionController", "pGain"),
        "positionControllerI": ("positionController", "iGain"),
        "positionControllerD": ("positionController", "dGain"),
        "positionControllerDdegree": ("positionController", "dDegree"),
        "positionControllerDbuffer": ("positionController", "dBufferLength"),
        "positionControllerMaxI": ("positionController", "iLimit"),
        "positionControllerMaxCmd": ("positionController", "cmdLimit"),
    }

    @classmethod
    def from_launch_params(cls, params: Mapping[str, object], **kw) -> "Config":
        """Build a Config from a ROS-parameter dictionary.  Keys may carry the absolute
        `/cdpr_gazebo_simulator/` prefix or not.  Unlike the reference, which ignores the
        return value of getParam and runs on uninitialised values (PLG.cpp:102-138), unknown
        keys raise and missing keys keep the shipped defaults."""
        cfg = cls(**kw)
        for key, value in params.items():
            short = key[len(LAUNCH_PARAM_PREFIX):] if key.startswith(LAUNCH_PARAM_PREFIX) else key.lstrip("/")
            if short not in cls._LAUNCH_KEYS:
                raise KeyError(f"unknown launch parameter {key!r}")
            path = cls._LAUNCH_KEYS[short]
            obj = cfg
            for attr in path[:-1]:
                obj = getattr(obj, attr)
            cur = getattr(obj, path[-1])
            setattr(obj, path[-1], int(value) if isinstance(cur, int) and not isinstance(cur, bool) else float(value))
        return cfg

    def launch_params(self) -> Dict[str, float]:
        out = {}
        for short, path in self._LAUNCH_KEYS.items():
            obj = self
            for attr in path:
                obj = getattr(obj, attr)
            out[LAUNCH_PARAM_PREFIX + short] = obj
        return out

    def effective_position_pid(self) -> PidParameters:
        """PLG.cpp:123,133: the position PID is built with forward gain 0 and both cascades 0."""
        p = self.positionController
        return replace(p, forwardGain=0.0, pFilter=replace(p.pFilter, cascade=0), dFilter=replace(p.dFilter, cascade=0))

    @property
    def n_cables(self) -> int:
        return self.model.n_cables

    def validate(self) -> None:
        n = self.n_cables
        if not 1 <= n <= _abi.MAX_CABLES:
            raise ValueError(f"invalid joint count {n} (1..{_abi.MAX_CABLES})")  # PLG.cpp:167-168
        if np.asarray(self.model.platform_anchors).shape != (n, 3) or np.asarray(self.model.frame_anchors).shape != (n, 3):
            raise ValueError("anchors must be [n,3]")
        if self.batch < 1:
            raise ValueError("batch must be >= 1")
        if not self.dt > 0:
            raise ValueError("dt must be > 0")
        for name, p in (("velocity", self.velocityController), ("position", self.positionController)):
            if not 2 <= p.dBufferLength <= _abi.MAX_D_BUFFER:
                raise ValueError(f"{name}Controller Dbuffer must be in 2..{_abi.MAX_D_BUFFER}")
            if not 1 <= p.dDegree <= _abi.MAX_D_DEGREE or p.dDegree >= p.dBufferLength:
                raise ValueError(f"{name}Controller Ddegree must be in 1..{_abi.MAX_D_DEGREE} and < Dbuffer")
            if not 0 <= p.pFilter.cascade <= _abi.MAX_CASCADE or not 0 <= p.dFilter.cascade <= _abi.MAX_CASCADE:
                raise ValueError(f"{name}Controller cascade must be in 0..{_abi.MAX_CASCADE}")
        if self.model.travel_lower > self.model.travel_upper or (self.model.travel_stop and self.model.travel_lower == self.model.travel_upper):
            raise ValueError("travel limits need lower < upper (lower == upper == 0: no limits, and then no stop)")
        if not 0 <= int(self.model.travel_stop) <= 64:
            raise ValueError("travel_stop (sweeps of the joint stop) must be in 0..64")
        if int(self.precision) not in (0, 32, 64):
            raise ValueError("precision must be 32 or 64")
        if self.stages & (_abi.STAGE_FK | _abi.STAGE_TD) and n < 6:
            raise ValueError("FK / tension distribution need at least 6 cables")

    def to_struct(self) -> _abi.ConfigStruct:
        self.validate()
        m = self.model
        s = _abi.ConfigStruct()
        s.abi_version = _abi.ABI_VERSION
        s.n_cables = self.n_cables
        s.batch = int(self.batch)
        s.dt = float(self.dt)
        fa = np.asarray(m.frame_anchors, dtype=np.float64)
        pa = np.asarray(m.platform_anchors, dtype=np.float64)
        l0 = m.reference_lengths()
        for i in range(self.n_cables):
            for k in range(3):
                s.frame_anchor[i][k] = fa[i, k]
                s.platform_anchor[i][k] = pa[i, k]
            s.cable_ref_length[i] = l0[i]
        for k, v in enumerate(m.home_pose()):
            s.home_pose[k] = v
        s.mass = float(m.mass)
        for k in range(6):
            s.inertia[k] = float(m.inertia[k])
        for k in range(3):
            s.gravity[k] = float(self.gravity[k])
        s.joint_damping = float(m.joint_damping)
        s.effort_limit = float(m.effort_limit)
        s.velocity_limit = float(m.velocity_limit)
        s.unilateral_cables = 1 if m.unilateral_cables else 0
        s.precision = int(self.precision)
        s.passive_damping = float(m.passive_damping)
        s.leg_inertia = float(m.leg_inertia)
        s.cable_axial_mass = float(m.cable_axial_mass)
        s.anchor_point_mass = float(m.anchor_point_mass)
        s.anchor_inertia = float(m.anchor_inertia)
        s.travel_lower = float(m.travel_lower)
        s.travel_upper = float(m.travel_upper)
        s.travel_stop = int(m.travel_stop)
        _fill_pid(s.velocity_pid, self.velocityController)
        _fill_pid(s.position_pid, self.effective_position_pid())
        s.velocity_epsilon = float(self.velocityEpsilon)
        s.publish_period = float(self.publishPeriod)
        s.stages = int(self.stages)
        s.mapping = int(self.mapping)
        s.fk_max_iterations = int(self.fkMaxIterations)
        s.per_robot_commands = 1 if self.perRobotCommands else 0
        s.fk_lambda = float(self.fkLambda)
        s.fk_tolerance = float(self.fkTolerance)
        s.td_f_min = float(m.f_min if self.tdFMin is None else self.tdFMin)
        s.td_f_max = float(m.f_max if self.tdFMax is None else self.tdFMax)
        return s


def _fill_pid(dst: _abi.PidParams, p: PidParameters) -> None:
    dst.forward_gain = float(p.forwardGain)
    dst.p_gain = float(p.pGain)
    dst.i_gain = float(p.iGain)
    dst.d_gain = float(p.dGain)
    dst.d_degree = int(p.dDegree)
    dst.d_buffer_length = int(p.dBufferLength)
    dst.i_limit = float(p.iLimit)
    dst.cmd_limit = float(p.cmdLimit)
    for f, src in ((dst.p_filter, p.pFilter), (dst.d_filter, p.dFilter)):
        f.rel_cutoff = float(src.relCutoff)
        f.quality = float(src.quality)
        f.cascade = int(src.cascade)
