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
from .messages import Header, JointState, Joy, KeyValue, PlatformState, Pose, Twist, WireStates

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
        self.frame_pose = np.array([0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0])
        self._wire_taut: Optional[np.ndarray] = None

    # ---- PLG.cpp:49-65
    def Load(self, config: Config, joint_names: Optional[List[str]] = None, frame_pose=None) -> None:
        """`joint_names` stands in for the model's joint list: names starting with `cable` are
        indexed by their numeric suffix (PLG.cpp:146-152); a wrong count raises (PLG.cpp:167-168).

        `frame_pose` (x y z qx qy qz qw) = WorldPose of the frame link, i.e. where the model was spawned (launch file
        `-x -y -z -R -P -Y`).  The engine simulates in FRAME coordinates — anchors are frame-relative (cube.yaml:21-29)
        and what `platformPose` carries is WorldPose(platform) - WorldPose(frame) (PLG.cpp:262-274), which is the
        engine's state itself — so the only thing the spawn pose changes is the direction of gravity seen from the
        frame: `config.gravity` is taken as the WORLD gravity and rotated into the frame.  `worldPlatformPose()` composes
        the state back into world coordinates."""
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
        if frame_pose is not None:
            from dataclasses import replace

            from scipy.spatial.transform import Rotation

            self.frame_pose = np.asarray(frame_pose, dtype=np.float64).reshape(7)
            g_frame = Rotation.from_quat(self.frame_pose[3:]).inv().apply(np.asarray(config.gravity, dtype=np.float64))
            config = replace(config, gravity=tuple(g_frame))
        self.config = config
        self.engine = Engine(config, self.device)  # validates, allocates, Position mode with target 0
        self._wire_taut = None
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
        self._pub_wire = self.bus.advertise(cWireStatesTopic, cPublisherQueueSize)  # PLG.cpp:196; events: publishWireStates
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

    # ---- JointForceCalculator::setForce (JFC.h:92-95) on every joint.  No topic of the reference plugin reaches it
    #      (UpdateMode::Force is the mode a calculator is constructed in, JFC.h:42, and dead code from Load on); a caller
    #      that computes its own tensions (tension distribution, MPC) drives the joints open loop through this.
    def setForce(self, aMsg: Joy) -> None:
        if self._accepts(aMsg):
            self.mForceCommand = aMsg
            self.mForceCommandReceived = True

    # ---- PLG.cpp:202-246 (+ the world step)
    def update(self, nsteps: int = 1) -> None:
        """`nsteps` Gazebo world iterations under the commands latched now.  Every step whose stamp passes the
        publishPeriod throttle (PLG.cpp:236-242) is published with its own stamp t_k, as the reference does once per
        iteration: for nsteps > 1 on the fast path the steps run as one fused launch chain that keeps every step's
        observables in a trajectory record (cdpr_update_record), so n steps give n messages per topic."""
        eng = self.engine
        self._velocity_queue.callAvailable()
        self._position_queue.callAvailable()
        if self.mVelocityCommandReceived:
            eng.set_velocity_command(self.mVelocityCommand.axes, mask=self.mVelocityCommand.robots)
            self.mVelocityCommandReceived = False
        if self.mPositionCommandReceived:
            eng.set_position_command(self.mPositionCommand.axes, mask=self.mPositionCommand.robots)
            self.mPositionCommandReceived = False
        if getattr(self, "mForceCommandReceived", False):  # [NEW] ordering: after the two Joy topics
            eng.set_force_command(self.mForceCommand.axes, mask=self.mForceCommand.robots)
            self.mForceCommandReceived = False
        first = eng.step_count
        dt = self.config.dt
        debug = bool(self.config.stages & _abi.STAGE_PID_DEBUG)
        record = None
        if nsteps > 1 and self.mPublishPeriod == 0.0 and not debug:
            try:  # fused launches of up to 20 steps, every step's observables kept
                record = eng.update_record(nsteps, min(nsteps, 20))
            except Exception as exc:  # the general controller path has no trajectory record: step one by one
                if getattr(exc, "code", None) != _abi.ERR_UNSUPPORTED:
                    raise
        if record is not None:
            for j in range(nsteps):
                t = (first + j) * dt
                if (t - self.mPreviousProcessingTime) > self.mPublishPeriod:  # never true for step 0: 0 - 0 > 0 is false
                    self.mPreviousProcessingTime = t
                    self._publish(t, record["position"][j], record["velocity"][j], record["effort"][j], record["pose"][j], record["twist"][j])
            return
        for j in range(nsteps):
            eng.update(1)
            t = (first + j) * dt
            if debug:  # the `pid` topic goes out every iteration, unthrottled (PLG.cpp:233-235)
                self._pub_pid(Joy(axes=eng.pid_debug(), header=Header(stamp=t)))
            if (t - self.mPreviousProcessingTime) > self.mPublishPeriod:
                self.mPreviousProcessingTime = t
                # publishJointStates + publishPlatformState, one device round trip; float64 arrays (what the ROS messages
                # carry, PLG.cpp:248-280) when the engine steps in the reference's precision
                self._publish(t, *(eng.observables_f64() if int(self.config.precision) == 64 else eng.observables()))

    def _publish(self, t, q, qd, eff, pose, twist) -> None:
        self._pub_joint(JointState(name=list(self.mJointNames), position=q, velocity=qd, effort=eff, header=Header(stamp=t)))
        self._pub_platform(PlatformState(pose=Pose(position=pose[:, 0:3], orientation=pose[:, 3:7]),
                                         velocity=Twist(linear=twist[:, 0:3], angular=twist[:, 3:6]), header=Header(stamp=t)))
        self.publishWireStates(t, eff)

    # ---- [NEW] the TODO of PLG.cpp:230-231: "develop logic for wire state publishing"
    def publishWireStates(self, aNow: float, effort: np.ndarray) -> List[WireStates]:
        """One cdpr_gazebo/WireStates event per cable whose state changed since the last published step: a cable is
        `slack` while the force applied to it is <= 0 (it would have to push), `taut` otherwise.  The reference only
        advertises the topic (PLG.cpp:196) and leaves the logic as a TODO, so this is this build's definition."""
        taut = np.asarray(effort) > 0.0
        events: List[WireStates] = []
        if self._wire_taut is not None:
            for b, i in zip(*np.nonzero(taut != self._wire_taut)):
                msg = WireStates(stateChange=KeyValue(key=self.mJointNames[i], value="taut" if taut[b, i] else "slack"),
                                 header=Header(stamp=aNow), robot=int(b))
                self._pub_wire(msg)
                events.append(msg)
        self._wire_taut = taut
        return events

    def worldPlatformPose(self):
        """WorldPose(platform) = WorldPose(frame) o platformPose: position [B, 3] and quaternion [B, 4] (x y z w)."""
        from scipy.spatial.transform import Rotation

        pose, _ = self.engine.platform_state()
        rf = Rotation.from_quat(self.frame_pose[3:])
        return self.frame_pose[:3] + rf.apply(pose[:, :3].astype(np.float64)), (rf * Rotation.from_quat(pose[:, 3:7].astype(np.float64))).as_quat()

    # ---- PLG.cpp:248-256
    def publishJointStates(self, aNow: float) -> JointState:
        q, qd, eff = self.engine.joint_states()
        msg = JointState(name=list(self.mJointNames), position=q, velocity=qd, effort=eff, header=Header(stamp=aNow))
        self._pub_joint(msg)
        self.publishWireStates(aNow, eff)
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
