"""Facade with the plugin-callback surface of the reference's `CdprGazeboPlugin`.

Same method names as CdprGazeboPlugin.h:92-102 (`Load`, `update`,
`cableVelocityCommandCallback`, `cablePositionCommandCallback`,
`publishJointStates`, `publishPlatformState`) and the same topic strings as
CdprGazeboPlugin.h:23-28, over a tiny in-process topic bus that stands in for
roscpp.  One facade drives B robots; every message carries a leading B
dimension (B = 1 reproduces the reference's single-robot messages).

What `update()` does here is what one Gazebo world iteration does in the
reference: the plugin's WorldUpdateBegin callback (PLG.cpp:202-246) followed by
the physics step, both inside the HIP step kernel.
"""
from __future__ import annotations

from collections import defaultdict, deque
from typing import Callable, Deque, Dict, List, Optional

import numpy as np

from . import _abi
from .config import Config
from .engine import Engine
from .messages import Header, JointState, Joy, PlatformState, Pose, Twist

# CdprGazeboPlugin.h:21-31
cSubscriberQueueSize = 256
cPublisherQueueSize = 256
cPidTopic = "pid"
cVelocityTopic = "jointVelocities"
cPositionTopic = "jointPositions"
cCableStatesTopic = "jointStates"
cWireStatesTopic = "wireStates"
cPlatformPoseTopic = "platformPose"
cSdfNameCable = "cable"
cSdfNameFrame = "frame"
cSdfNamePlatform = "platform"


class TopicBus:
    """In-process stand-in for the ROS master: advertise / subscribe / publish by topic string."""

    def __init__(self):
        self._subs: Dict[str, List[Callable]] = defaultdict(list)
        self.advertised: Dict[str, int] = {}

    def advertise(self, topic: str, queue_size: int = cPublisherQueueSize) -> Callable:
        self.advertised[topic] = queue_size
        return lambda msg: self.publish(topic, msg)

    def subscribe(self, topic: str, callback: Callable, queue_size: int = cSubscriberQueueSize) -> None:
        self._subs[topic].append(callback)

    def publish(self, topic: str, msg) -> None:
        for cb in self._subs.get(topic, ()):
            cb(msg)


class _CallbackQueue:
    """ros::CallbackQueue drained on the physics thread (PLG.h:66,71; PLG.cpp:203-204)."""

    def __init__(self, maxlen: int):
        self._q: Deque = deque(maxlen=maxlen)

    def push(self, fn: Callable, msg) -> None:
        self._q.append((fn, msg))

    def callAvailable(self) -> None:
        while self._q:
            fn, msg = self._q.popleft()
            fn(msg)


class CdprGazeboPlugin:
    def __init__(self, bus: Optional[TopicBus] = None, device: int = 0):
        self.bus = bus or TopicBus()
        self.device = device
        self.engine: Optional[Engine] = None
        self._velocity_queue = _CallbackQueue(cSubscriberQueueSize)
        self._position_queue = _CallbackQueue(cSubscriberQueueSize)
        self.mVelocityCommand: Optional[Joy] = None
        self.mPositionCommand: Optional[Joy] = None
        self.mVelocityCommandReceived = False
        self.mPositionCommandReceived = False
        self.mJointNames: List[str] = []
        self.mPublishPeriod = 0.0
        self.mPreviousProcessingTime = 0.0

    # ---- PLG.cpp:49-65
    def Load(self, config: Config, joint_names: Optional[List[str]] = None) -> None:
        """`joint_names` stands in for the model's joint list: names starting with `cable` are
        indexed by their numeric suffix (PLG.cpp:146-152); a wrong count raises (PLG.cpp:167-168)."""
        n = config.n_cables
        names = joint_names if joint_names is not None else [f"{cSdfNameCable}{i}" for i in range(n)]
        self.mJointNames = [""] * n
        found = 0
        for name in names:
            if name.find(cSdfNameCable) == 0:
                idx = int(name[len(cSdfNameCable):])
                if idx < n:
                    self.mJointNames[idx] = name
                    found += 1
        if found != n:
            raise ValueError("invalid joint count")
        self.config = config
        self.engine = Engine(config, self.device)  # validates, allocates, Position mode with target 0
        self.initCommunication()
        self.mPublishPeriod = config.publishPeriod
        self.mPreviousProcessingTime = 0.0

    # ---- PLG.cpp:176-198
    def initCommunication(self) -> None:
        self.bus.subscribe(cVelocityTopic, lambda m: self._velocity_queue.push(self.cableVelocityCommandCallback, m), cSubscriberQueueSize)
        self.bus.subscribe(cPositionTopic, lambda m: self._position_queue.push(self.cablePositionCommandCallback, m), cSubscriberQueueSize)
        self.mVelocityCommandReceived = False
        self.mPositionCommandReceived = False
        self._pub_joint = self.bus.advertise(cCableStatesTopic, cPublisherQueueSize)
        self._pub_pid = self.bus.advertise(cPidTopic, cPublisherQueueSize)
        self._pub_wire = self.bus.advertise(cWireStatesTopic, cPublisherQueueSize)  # advertised, never published (PLG.cpp:196,230-231)
        self._pub_platform = self.bus.advertise(cPlatformPoseTopic, cPublisherQueueSize)

    def _accepts(self, msg: Joy) -> bool:
        """axes.size() == cWireCount (PLG.cpp:68,77); a batched Joy may carry [B, n]."""
        a = np.asarray(msg.axes)
        n, B = self.config.n_cables, self.config.batch
        return a.size == n or (a.size == n * B and (a.ndim == 1 or a.shape == (B, n)))

    # ---- PLG.cpp:67-74
    def cableVelocityCommandCallback(self, aMsg: Joy) -> None:
        if self._accepts(aMsg):
            self.mVelocityCommand = aMsg
            self.mVelocityCommandReceived = True

    # ---- PLG.cpp:76-83
    def cablePositionCommandCallback(self, aMsg: Joy) -> None:
        if self._accepts(aMsg):
            self.mPositionCommand = aMsg
            self.mPositionCommandReceived = True

    # ---- PLG.cpp:202-246 (+ the world step)
    def update(self, nsteps: int = 1) -> None:
        eng = self.engine
        self._velocity_queue.callAvailable()
        self._position_queue.callAvailable()
        if self.mVelocityCommandReceived:
            eng.set_velocity_command(self.mVelocityCommand.axes)
            self.mVelocityCommandReceived = False
        if self.mPositionCommandReceived:
            eng.set_position_command(self.mPositionCommand.axes)
            self.mPositionCommandReceived = False
        first = eng.step_count
        # several world steps under one held command: fuse them into launches of up to 16 steps (state stays on chip
        # between them; bit-identical to single-step launches, tests/test_gpu_parity.py)
        eng.update(nsteps, min(max(nsteps, 1), 16))
        # stamps of the steps just run: t_k = k * dt; the engine applied the same
        # throttle on the device (PLG.cpp:236-242), here it gates the host-side publish
        now = (first + nsteps - 1) * self.config.dt
        if self.config.stages & _abi.STAGE_PID_DEBUG:
            self._pub_pid(Joy(axes=eng.pid_debug(), header=Header(stamp=now)))
        published = False
        for k in range(first, first + nsteps):
            t = k * self.config.dt
            if (t - self.mPreviousProcessingTime) > self.mPublishPeriod:
                self.mPreviousProcessingTime = t
                published = True
        if published:
            self.publishJointStates(self.mPreviousProcessingTime)
            self.publishPlatformState(self.mPreviousProcessingTime)

    # ---- PLG.cpp:248-256
    def publishJointStates(self, aNow: float) -> JointState:
        q, qd, eff = self.engine.joint_states()
        msg = JointState(name=list(self.mJointNames), position=q, velocity=qd, effort=eff, header=Header(stamp=aNow))
        self._pub_joint(msg)
        return msg

    # ---- PLG.cpp:258-280
    def publishPlatformState(self, aNow: float) -> PlatformState:
        pose, twist = self.engine.platform_state()
        msg = PlatformState(
            pose=Pose(position=pose[:, 0:3], orientation=pose[:, 3:7]),
            velocity=Twist(linear=twist[:, 0:3], angular=twist[:, 3:6]),
            header=Header(stamp=aNow),
        )
        self._pub_platform(msg)
        return msg
