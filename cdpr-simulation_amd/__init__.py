"""cdpr-simulation_amd — MI355X-native batched CDPR step engine.

Drop-in for the per-step hot path of the `cdpr_gazebo` plugin
(balazs-bamer/cdpr-simulation): IK -> per-cable PID -> platform dynamics, plus
Newton-Raphson FK and tension distribution for >= 6-cable robots, for tens of
thousands of independent robots per GPU.  Host code is Python over a ctypes C-ABI
(include/cdpr.h) onto hand-written gfx950 HIP kernels; there is no CPU fallback.
"""
from . import _abi, stimulus
from .config import Config, FilterParameters, Model, PidParameters, cube_model, eight_cable_model, twelve_cable_model
from .engine import CdprError, Engine, derivative_weights, plan_kernel
from .model_io import load_launch, load_sdf, load_yaml
from .messages import Header, JointState, Joy, KeyValue, PlatformState, Pose, Twist, WireStates
from .plugin import CdprGazeboPlugin, TopicBus
from .sharding import ShardedEngine, shard_range

__all__ = [
    "Config", "FilterParameters", "Model", "PidParameters", "cube_model", "eight_cable_model", "twelve_cable_model",
    "load_launch", "load_sdf", "load_yaml", "Engine", "CdprError", "derivative_weights", "plan_kernel", "CdprGazeboPlugin", "TopicBus", "ShardedEngine", "shard_range",
    "Header", "JointState", "Joy", "KeyValue", "PlatformState", "Pose", "Twist", "WireStates", "stimulus", "_abi",
]  # fmt: skip
