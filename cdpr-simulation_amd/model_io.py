"""Model loaders: read a CDPR description in the reference's own file formats instead of transcribed constants.

* `load_yaml`  — the upstream `cdpr` package layout that `sdf/cube.yaml` uses (cube.yaml:1-29): `points: [{frame,
  platform}]`, `platform: {mass, inertia, position: {xyz, rpy}}`, `joints: {actuated: {damping, effort, min}}`.
  Any cable count 1..8, so an 8-cable robot is data, not code.
* `load_launch` — the roslaunch file that starts the reference (`launch/cdpr_gazebo.launch`): the controller parameters
  under the `cdpr_gazebo_simulator` node (launch:17-39, the keys `CdprGazeboPlugin::Load` reads, PLG.h:32-54), the spawn
  pose of the model (`-x -y -z -R -P -Y`, launch:16) and the SDF / YAML files it names.
* `load_sdf`   — the SDF that `gen_cdpr.py` emits and Gazebo loads (`sdf/cube.sdf`): the platform link's pose and
  inertial (cube.sdf:309-342), the frame-side anchor = pose of link `virt_X<i>` (gen_cdpr.py:140-150 puts it at the
  frame attach point), the platform-side anchor = pose of link `virt_Xpf<i>` at spawn, damping / effort of the
  prismatic joint `cable<i>` (cube.sdf:429-445).  The spawn pose in the SDF is authoritative (cube.yaml's z = 2 is
  not what is loaded).
"""
from __future__ import annotations

import math
import re
import xml.etree.ElementTree as ET
from typing import Optional, Sequence

import numpy as np

from .config import Model, quat_to_matrix


def rpy_to_quat(roll: float, pitch: float, yaw: float) -> np.ndarray:
    """Fixed-axis roll-pitch-yaw (SDF / `transformations.euler_matrix(..., 'sxyz')`) to quaternion x y z w."""
    cr, sr = math.cos(roll / 2), math.sin(roll / 2)
    cp, sp = math.cos(pitch / 2), math.sin(pitch / 2)
    cy, sy = math.cos(yaw / 2), math.sin(yaw / 2)
    return np.array([sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy])


def load_yaml(path_or_text: str, home_xyz: Optional[Sequence[float]] = None, travel_limits: bool = True) -> Model:
    """The prismatic joints' travel range is what gen_cdpr.py derives from the frame box (gen_cdpr.py:104,182-183:
    +-0.5 |frame.upper - frame.lower|, the 0.51961524 of cube.sdf:436-437); `travel_limits=False` leaves it out.
    `home_xyz` overrides `platform.position.xyz` (for cube.yaml pass (0, 0, 0.3): the SDF that is actually loaded
    has the platform there, cube.sdf:310, while the yaml still says z = 2)."""
    import yaml

    text = open(path_or_text).read() if "\n" not in path_or_text else path_or_text
    y = yaml.safe_load(text)
    pts = y["points"]
    fa = np.array([p["frame"] for p in pts], dtype=np.float64)
    pa = np.array([p["platform"] for p in pts], dtype=np.float64)
    plat = y.get("platform", {})
    pos = plat.get("position", {})
    xyz = list(home_xyz) if home_xyz is not None else list(pos.get("xyz", [0.0, 0.0, 0.0]))
    rpy = list(pos.get("rpy", [0.0, 0.0, 0.0]))
    act = y.get("joints", {}).get("actuated", {})
    frame = y.get("frame", {}) or {}
    half = 0.0
    if travel_limits and "lower" in frame and "upper" in frame:
        half = 0.5 * float(np.linalg.norm(np.asarray(frame["upper"], dtype=np.float64) - np.asarray(frame["lower"], dtype=np.float64)))
    return Model(
        frame_anchors=fa,
        platform_anchors=pa,
        home_position=tuple(float(v) for v in xyz),
        home_quaternion=tuple(rpy_to_quat(*[float(v) for v in rpy])),
        mass=float(plat.get("mass", 1.0)),
        inertia=tuple(float(v) for v in plat.get("inertia", [1, 1, 1, 0, 0, 0])),
        joint_damping=float(act.get("damping", 1.0)),
        effort_limit=float(act.get("effort", 100.0)),
        f_min=float(act.get("min", 5.0)),
        f_max=float(act.get("effort", 100.0)),
        travel_lower=-half,
        travel_upper=half,
    )


def _pose(elem) -> np.ndarray:
    p = elem.find("pose")
    return np.array([float(v) for v in p.text.split()]) if p is not None else np.zeros(6)


def load_sdf(path_or_text: str, f_min: float = 5.0, lumped_links: bool = False, travel_limits: bool = True, travel_stop: int = 0) -> Model:
    """The prismatic joints' travel limits `<lower>` / `<upper>` (cube.sdf:436-437) go into the model (`travel_limits=False`
    leaves them out): the engine then flags every joint that leaves the range (`Engine.limit_state`); `travel_stop=k`
    also models the stop (k Gauss-Seidel sweeps per step; 4 is plenty).  `lumped_links=True` also reads what the massless-cable reduction drops into the model's lumped
    leg terms: the damping of the passive revolute joints (`rev_X<i>`, cube.sdf:396) and the masses / inertias of the
    five small links of every leg (cube.sdf:359-369, 372-382, ...): virt_X + virt_Y + cable + virt_Ypf turn with the leg
    (leg_inertia), the cable link slides along the axis (cable_axial_mass), virt_Xpf + virt_Ypf ride on the platform
    anchor (anchor_point_mass), virt_Xpf turns with the platform (anchor_inertia).  Off by default: the contract's
    reduced model."""
    text = open(path_or_text).read() if "<" not in path_or_text else path_or_text
    root = ET.fromstring(text)
    model = root.find("model") if root.tag == "sdf" else root
    links = {l.get("name"): l for l in model.findall("link")}
    joints = {j.get("name"): j for j in model.findall("joint")}
    if "platform" not in links:
        raise ValueError("SDF has no link named 'platform' (CdprGazeboPlugin.h:31)")
    frame_pose = _pose(links["frame"]) if "frame" in links else np.zeros(6)
    if np.abs(frame_pose).max() > 0:
        raise ValueError("frame link must sit at the model origin (the engine simulates in frame coordinates)")
    plat = links["platform"]
    ppose = _pose(plat)
    q = rpy_to_quat(*ppose[3:])
    r = quat_to_matrix(q)
    inertial = plat.find("inertial")
    inert = inertial.find("inertia")
    mass = float(inertial.find("mass").text)
    inertia = tuple(float(inert.find(k).text) for k in ("ixx", "iyy", "izz", "ixy", "ixz", "iyz"))
    # actuated joints: names starting with `cable`, index = numeric suffix (CdprGazeboPlugin.cpp:146-152)
    idx = sorted(int(re.sub(r"^cable", "", n)) for n, j in joints.items() if n.startswith("cable") and j.get("type") == "prismatic")
    if not idx or idx != list(range(len(idx))):
        raise ValueError("invalid joint count")  # CdprGazeboPlugin.cpp:167-168
    fa, pa, damping, effort = [], [], None, None
    lower = upper = 0.0
    for i in idx:
        fa.append(_pose(links[f"virt_X{i}"])[:3])
        world_attach = _pose(links[f"virt_Xpf{i}"])[:3]
        pa.append(r.T @ (world_attach - ppose[:3]))
        axis = joints[f"cable{i}"].find("axis")
        damping = float(axis.find("dynamics/damping").text)
        effort = float(axis.find("limit/effort").text)
        lo, hi = axis.find("limit/lower"), axis.find("limit/upper")
        if travel_limits and lo is not None and hi is not None and i == idx[0]:
            lower, upper = float(lo.text), float(hi.text)
        elif travel_limits and lo is not None and hi is not None and (float(lo.text), float(hi.text)) != (lower, upper):
            raise ValueError("the engine takes one travel range for all cables; the SDF gives different ones")
    lumped = {}
    if lumped_links:
        def inertial(name):
            el = links[name].find("inertial") if name in links else None
            if el is None:
                return 0.0, 0.0
            return float(el.find("mass").text), float(el.find("inertia/ixx").text)

        m_c, i_c = inertial("cable0")
        (m_x, i_x), (m_y, i_y), (m_xp, i_xp), (m_yp, i_yp) = (inertial(f"{k}0") for k in ("virt_X", "virt_Y", "virt_Xpf", "virt_Ypf"))
        passive = joints.get("rev_X0")
        lumped = dict(
            passive_damping=float(passive.find("axis/dynamics/damping").text) if passive is not None else 0.0,
            leg_inertia=i_x + i_y + i_c + i_yp,
            cable_axial_mass=m_c,
            anchor_point_mass=m_xp + m_yp,
            anchor_inertia=i_xp,
        )
    return Model(
        frame_anchors=np.array(fa),
        platform_anchors=np.array(pa),
        home_position=tuple(ppose[:3]),
        home_quaternion=tuple(q),
        mass=mass,
        inertia=inertia,
        joint_damping=damping,
        effort_limit=effort,
        f_min=f_min,
        f_max=effort,
        travel_lower=lower,
        travel_upper=upper,
        travel_stop=int(travel_stop) if lower < upper else 0,
        **lumped,
    )


def load_launch(path_or_text: str):
    """Read the reference's roslaunch file (launch/cdpr_gazebo.launch).  Returns a dict:
      params      {"/cdpr_gazebo_simulator/<key>": value}: what the plugin reads from the parameter server (PLG.cpp:57,102-138);
                  feed it to `Config.from_launch_params`
      frame_pose  [x, y, z, qx, qy, qz, qw] of the spawned model (the `-x -y -z -R -P -Y` of the spawn_model node,
                  launch:16) = `CdprGazeboPlugin.Load(config, frame_pose=...)`
      sdf_file, model_yaml   the files the launch names (`$(find pkg)/` prefixes kept as written), or None
    Only this file's own `<param>` tags count (commented-out alternatives, launch:40-46, are comments to XML too)."""
    from .config import LAUNCH_PARAM_PREFIX

    text = open(path_or_text).read() if "<" not in path_or_text else path_or_text
    root = ET.fromstring(text)
    out = {"params": {}, "frame_pose": [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0], "sdf_file": None, "model_yaml": None}
    for node in root.iter("node"):
        if node.get("type") != "spawn_model":
            continue
        ns = "/" + node.get("name", "").strip("/") + "/"
        for prm in node.findall("param"):
            raw = prm.get("value", "")
            try:
                val = int(raw)
            except ValueError:
                val = float(raw)
            out["params"][(LAUNCH_PARAM_PREFIX if ns == LAUNCH_PARAM_PREFIX else ns) + prm.get("name")] = val
        argstr = node.get("args", "")
        m = re.search(r"-file\s+((?:\$\([^)]*\)|\S)+)", argstr)  # a roslaunch substitution `$(find pkg)` contains a blank
        if m:
            out["sdf_file"] = m.group(1)
        args = argstr.split()
        spawn = {"-x": 0.0, "-y": 0.0, "-z": 0.0, "-R": 0.0, "-P": 0.0, "-Y": 0.0}
        for i, tok in enumerate(args[:-1]):
            if tok in spawn:
                spawn[tok] = float(args[i + 1])
        out["frame_pose"] = [spawn["-x"], spawn["-y"], spawn["-z"]] + [float(v) for v in rpy_to_quat(spawn["-R"], spawn["-P"], spawn["-Y"])]
    for rp in root.iter("rosparam"):
        if rp.get("command") == "load" and rp.get("file"):
            out["model_yaml"] = rp.get("file")
    return out
