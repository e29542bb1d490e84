"""Message records of the plugin's topics (batched: every array has a leading B dimension).

Field names follow the ROS message definitions the reference uses:
sensor_msgs/Joy (`axes: float32[]`), sensor_msgs/JointState (`name`, `position`,
`velocity`, `effort`), cdpr_gazebo/PlatformState (msg/PlatformState.msg:1-3 =
Header + geometry_msgs/Pose + geometry_msgs/Twist), cdpr_gazebo/WireStates
(msg/WireStates.msg:1-2 = Header + diagnostic_msgs/KeyValue stateChange).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List

import numpy as np


@dataclass
class Header:
    stamp: float = 0.0  # seconds of sim time (launch:5 use_sim_time)
    seq: int = 0
    frame_id: str = ""


@dataclass
class Joy:  # sensor_msgs/Joy; axes are float32 on the wire
    axes: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.float32))
    header: Header = field(default_factory=Header)
    robots: "np.ndarray | None" = None  # batched facade only: mask[B] of the robots this Joy reaches (None = all)

    def __post_init__(self):
        self.axes = np.asarray(self.axes, dtype=np.float32)


@dataclass
class JointState:  # sensor_msgs/JointState, PLG.cpp:188-192,248-256
    name: List[str]
    position: np.ndarray  # [B, n] float64 on the ROS wire; float32 from the GPU
    velocity: np.ndarray
    effort: np.ndarray
    header: Header = field(default_factory=Header)


@dataclass
class Pose:
    position: np.ndarray  # [B, 3] x y z
    orientation: np.ndarray  # [B, 4] x y z w (PLG.cpp:266-269)


@dataclass
class Twist:
    linear: np.ndarray  # [B, 3]
    angular: np.ndarray  # [B, 3]


@dataclass
class PlatformState:  # cdpr_gazebo/PlatformState, PLG.cpp:258-280
    pose: Pose
    velocity: Twist
    header: Header = field(default_factory=Header)


@dataclass
class KeyValue:  # diagnostic_msgs/KeyValue
    key: str = ""
    value: str = ""


@dataclass
class WireStates:  # cdpr_gazebo/WireStates, msg/WireStates.msg:1-2; advertised at PLG.cpp:196, publishing is the TODO of PLG.cpp:230-231
    stateChange: KeyValue = field(default_factory=KeyValue)
    header: Header = field(default_factory=Header)
    robot: int = 0  # batched facade only: which robot of the batch the event belongs to
