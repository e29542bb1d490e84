"""Model loaders: the upstream YAML layout and the generated SDF, n cables as data."""
import os

import numpy as np
import pytest

REF = "/root/reference/src/cdpr_gazebo/sdf"

EIGHT_YAML = """
cable: {radius: 0.005}
frame: {type: box, lower: [-0.3, -0.3, 0], upper: [0.3, 0.3, 0.6]}
joints:
  actuated: {damping: 1, effort: 100, min: 5, velocity: 10}
  passive: {damping: 0.01, effort: 100, velocity: 10}
platform:
  mass: 1
  inertia: [1, 1, 1, 0, 0, 0]
  position: {rpy: [0, 0, 0], xyz: [0, 0, 0.3]}
  size: [0.06, 0.06, 0.015]
points:
- {frame: [-0.3, -0.3, 0.6], platform: [0.03, -0.03, -0.0075]}
- {frame: [-0.3, 0.3, 0.6], platform: [-0.03, 0.03, -0.0075]}
- {frame: [0.3, 0.3, 0.6], platform: [0.03, 0.03, -0.0075]}
- {frame: [0.3, -0.3, 0.6], platform: [-0.03, -0.03, -0.0075]}
- {frame: [-0.3, -0.3, 0.0], platform: [-0.03, -0.03, 0.0075]}
- {frame: [-0.3, 0.3, 0.0], platform: [-0.03, 0.03, 0.0075]}
- {frame: [0.3, 0.3, 0.0], platform: [0.03, -0.03, 0.0075]}
- {frame: [0.3, -0.3, 0.0], platform: [0.03, 0.03, 0.0075]}
"""


def mini_sdf(anchors_f, anchors_p_world, plat_pose):
    links = "".join(
        f'<link name="virt_X{i}"><pose>{a[0]} {a[1]} {a[2]} 0 0 0</pose></link>'
        f'<link name="virt_Xpf{i}"><pose>{b[0]} {b[1]} {b[2]} 0 0 0</pose></link>'
        f'<joint name="cable{i}" type="prismatic"><parent>virt_Y{i}</parent><child>cable{i}</child><axis><xyz>0 0 1</xyz>'
        f"<limit><lower>-0.5</lower><upper>0.5</upper><effort>80</effort><velocity>10</velocity></limit>"
        f"<dynamics><damping>0.7</damping></dynamics></axis></joint>"
        for i, (a, b) in enumerate(zip(anchors_f, anchors_p_world))
    )
    return (
        '<?xml version="1.0"?><sdf version="1.4"><model name="m"><link name="frame"><pose>0 0 0 0 0 0</pose></link>'
        f'<link name="platform"><pose>{" ".join(str(v) for v in plat_pose)}</pose><inertial><inertia><ixx>1.5</ixx><iyy>2</iyy><izz>2.5</izz>'
        "<ixy>0.1</ixy><ixz>0</ixz><iyz>0</iyz></inertia><mass>3</mass></inertial></link>" + links + "</model></sdf>"
    )


def test_yaml_eight_cable_equals_builtin_model(pkg):
    m = pkg.load_yaml(EIGHT_YAML)
    ref = pkg.eight_cable_model()
    assert m.n_cables == 8
    assert np.allclose(m.frame_anchors, ref.frame_anchors) and np.allclose(m.platform_anchors, ref.platform_anchors)
    assert (m.mass, m.joint_damping, m.effort_limit, m.f_min, m.f_max) == (1.0, 1.0, 100.0, 5.0, 100.0)
    assert np.allclose(m.reference_lengths(), ref.reference_lengths())
    pkg.Config(model=m, stages=3).to_struct()  # validates


def test_sdf_loader_recovers_anchors_through_a_rotated_spawn_pose(pkg):
    rng = np.random.default_rng(0)
    fa = rng.uniform(-0.3, 0.3, (6, 3)) + [0, 0, 0.5]
    pb = rng.uniform(-0.05, 0.05, (6, 3))
    pose = [0.01, -0.02, 0.3, 0.1, -0.2, 0.3]
    q = pkg.model_io.rpy_to_quat(*pose[3:])
    r = pkg.config.quat_to_matrix(q)
    world = pose[:3] + pb @ r.T
    m = pkg.load_sdf(mini_sdf(fa, world, pose))
    assert m.n_cables == 6 and np.allclose(m.frame_anchors, fa) and np.allclose(m.platform_anchors, pb, atol=1e-12)
    assert m.mass == 3.0 and m.inertia == (1.5, 2.0, 2.5, 0.1, 0.0, 0.0) and m.joint_damping == 0.7 and m.effort_limit == 80.0
    assert np.allclose(m.home_quaternion, q) and np.allclose(m.reference_lengths(), np.linalg.norm(world - fa, axis=1))
    # the prismatic joints' travel range comes along (cube.sdf:436-437 layout: <limit><lower/><upper/>), flag only by default
    assert (m.travel_lower, m.travel_upper, m.travel_stop) == (-0.5, 0.5, 0)
    assert pkg.load_sdf(mini_sdf(fa, world, pose), travel_stop=4).travel_stop == 4
    assert pkg.load_sdf(mini_sdf(fa, world, pose), travel_limits=False).travel_upper == 0.0
    s = pkg.Config(model=m, stages=1).to_struct()
    assert (s.travel_lower, s.travel_upper, s.travel_stop) == (-0.5, 0.5, 0)


def test_travel_limits_match_the_golden_model_data(pkg):
    """tests/golden/cube_model.json holds what cube.sdf says about every prismatic joint (data read out of the reference's
    file by make_golden.py): the loader's range for EIGHT_YAML's frame box is the same +-0.51961524 (gen_cdpr.py:104)."""
    import json

    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "cube_model.json")))
    lows, ups = [], []

    def walk(o):
        if isinstance(o, dict):
            if isinstance(o.get("lower"), float) and isinstance(o.get("upper"), float):
                lows.append(o["lower"]), ups.append(o["upper"])
            for v in o.values():
                walk(v)
        elif isinstance(o, list):
            for v in o:
                walk(v)

    walk(gold)
    assert len(lows) == 4 and set(lows) == {-0.51961524} and set(ups) == {0.51961524}
    m = pkg.load_yaml(EIGHT_YAML)
    assert abs(m.travel_upper - 0.51961524) < 1e-8 and abs(m.travel_lower + 0.51961524) < 1e-8
    with pytest.raises(ValueError):
        bad = pkg.eight_cable_model()
        bad.travel_lower, bad.travel_upper = 0.1, -0.1
        pkg.Config(model=bad).to_struct()


def test_sdf_without_contiguous_cable_joints_is_rejected(pkg):
    bad = mini_sdf([[0, 0, 1]], [[0, 0, 0.3]], [0, 0, 0.3, 0, 0, 0]).replace('name="cable0"', 'name="cable3"')
    with pytest.raises(ValueError, match="invalid joint count"):
        pkg.load_sdf(bad)


def test_rpy_convention_matches_scipy(pkg):
    from scipy.spatial.transform import Rotation

    for rpy in ([0.3, -0.2, 1.1], [-2.408778, 0.589592, -1.338805]):
        q = pkg.model_io.rpy_to_quat(*rpy)
        assert np.allclose(pkg.config.quat_to_matrix(q), Rotation.from_euler("xyz", rpy).as_matrix(), atol=1e-12)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree only exists in the build container")
def test_reference_files_load_to_the_transcribed_model(pkg):
    ref = pkg.cube_model()
    for m in (pkg.load_sdf(os.path.join(REF, "cube.sdf")), pkg.load_yaml(os.path.join(REF, "cube.yaml"), home_xyz=(0, 0, 0.3))):
        assert m.n_cables == 4
        assert np.allclose(m.frame_anchors, ref.frame_anchors, atol=1e-6) and np.allclose(m.platform_anchors, ref.platform_anchors, atol=1e-6)
        assert np.allclose(m.home_position, ref.home_position) and np.allclose(m.home_quaternion, ref.home_quaternion)
        assert (m.mass, tuple(m.inertia), m.joint_damping, m.effort_limit) == (ref.mass, tuple(ref.inertia), ref.joint_damping, ref.effort_limit)
        assert np.allclose(m.reference_lengths(), 0.485592422, atol=1e-6)
    assert pkg.load_yaml(os.path.join(REF, "cube.yaml")).home_position == (0.0, 0.0, 2.0)  # what the yaml itself says (cube.yaml:17)
    # travel limits: the SDF's own numbers (cube.sdf:436-437) and the yaml's via gen_cdpr.py:104,182-183 (half the frame diagonal)
    sdf, yml = pkg.load_sdf(os.path.join(REF, "cube.sdf")), pkg.load_yaml(os.path.join(REF, "cube.yaml"))
    assert (sdf.travel_lower, sdf.travel_upper) == (-0.51961524, 0.51961524)
    assert abs(yml.travel_upper - 0.51961524) < 1e-8 and yml.travel_lower == -yml.travel_upper


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree only exists in the build container")
def test_reference_sdf_lumped_link_terms(pkg):
    """The five 0.001 kg / 0.001 kg m^2 links and the 0.01 passive damping of every leg of cube.sdf, as lumped terms."""
    m = pkg.load_sdf(os.path.join(REF, "cube.sdf"), lumped_links=True)
    assert (m.passive_damping, m.cable_axial_mass) == (0.01, 0.001)
    assert abs(m.leg_inertia - 0.004) < 1e-15 and abs(m.anchor_point_mass - 0.002) < 1e-15 and m.anchor_inertia == 0.001
    plain = pkg.load_sdf(os.path.join(REF, "cube.sdf"))
    assert plain.passive_damping == plain.leg_inertia == plain.anchor_point_mass == 0.0  # off by default: the contract's reduced model
    s = pkg.Config(model=m).to_struct()
    assert (s.passive_damping, s.leg_inertia, s.cable_axial_mass, s.anchor_point_mass, s.anchor_inertia) == (0.01, m.leg_inertia, 0.001, m.anchor_point_mass, 0.001)


LAUNCH_TEXT = """<?xml version="1.0"?>
<launch>
  <node name="cdpr_gazebo_simulator" pkg="gazebo_ros" type="spawn_model" respawn="false" output="screen"
        args="-sdf -model cube -file $(find cdpr_gazebo)/sdf/cube.sdf -x 0.1 -y -0.2 -z 0.3 -R 0.0 -P 0.0 -Y 1.5707963267948966">
    <param name="publishPeriod"              value="0.002" />
    <param name="velocityEpsilon"            value="-0.001" />
    <param name="velocityControllerP"        value="150"/>
    <param name="velocityControllerDbuffer"  value="9"/>
    <param name="positionControllerD"        value="40.5" />
<!-- an alternative tuning, commented out as in the reference's launch file>
    <param name="positionControllerP" value="20.0" />
    <param name="positionControllerMaxCmd" value="1.0" /-->
  </node>
  <rosparam file="$(find cdpr_gazebo)/sdf/cube.yaml" command="load" ns="model"/>
</launch>"""


def test_launch_file_loader(pkg):
    """launch/cdpr_gazebo.launch layout: controller parameters under the spawn_model node, spawn pose, file names."""
    d = pkg.load_launch(LAUNCH_TEXT)
    assert d["sdf_file"].endswith("sdf/cube.sdf") and d["model_yaml"].endswith("sdf/cube.yaml")
    assert d["params"]["/cdpr_gazebo_simulator/velocityControllerP"] == 150 and d["params"]["/cdpr_gazebo_simulator/positionControllerD"] == 40.5
    assert "/cdpr_gazebo_simulator/positionControllerMaxCmd" not in d["params"]  # inside the XML comment
    cfg = pkg.Config.from_launch_params(d["params"])
    assert (cfg.publishPeriod, cfg.velocityController.pGain, cfg.velocityController.dBufferLength, cfg.positionController.dGain) == (0.002, 150.0, 9, 40.5)
    assert cfg.positionController.pGain == 200.0  # untouched keys keep the shipped defaults
    assert np.allclose(d["frame_pose"][:3], [0.1, -0.2, 0.3]) and np.allclose(d["frame_pose"][3:], [0, 0, np.sqrt(0.5), np.sqrt(0.5)])


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree only exists in the build container")
def test_reference_launch_file_gives_the_shipped_configuration(pkg):
    """The reference's own launch file (launch/cdpr_gazebo.launch:16-39) loads to exactly the defaults `Config()` carries."""
    d = pkg.load_launch(os.path.join(REF, "..", "launch", "cdpr_gazebo.launch"))
    assert len(d["params"]) == 23  # PLG.h:32-54: publishPeriod, velocityEpsilon, 14 velocity-controller and 7 position-controller keys
    assert pkg.Config.from_launch_params(d["params"]).launch_params() == pkg.Config().launch_params()
    assert d["frame_pose"] == [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0] and d["sdf_file"].endswith("sdf/cube.sdf")


def test_yaml_with_twelve_points_loads_and_is_routed(pkg):
    """sdf/cube.yaml:21-29 is a free-length `points` list: twelve entries (round 6: CDPR_MAX_CABLES 12) load to the built-in
    twelve-cable model, validate with FK + TD, and are routed to the first-generation lane-per-robot kernel; thirteen are refused."""
    ref = pkg.twelve_cable_model()
    pts = "\n".join(f"  - frame: [{a[0]}, {a[1]}, {a[2]}]\n    platform: [{b[0]}, {b[1]}, {b[2]}]" for a, b in zip(ref.frame_anchors, ref.platform_anchors))
    text = "platform:\n  mass: 1.0\n  position:\n    xyz: [0, 0, 0.3]\n    rpy: [0, 0, 0]\npoints:\n" + pts + "\n"
    m = pkg.load_yaml(text)
    assert m.n_cables == 12 and np.allclose(m.frame_anchors, ref.frame_anchors) and np.allclose(m.platform_anchors, ref.platform_anchors)
    cfg = pkg.Config(model=m, batch=1000, stages=3)
    cfg.to_struct()
    assert pkg.plan_kernel(cfg, 1) == "cdpr_step_kernel<12, true, true, SINGLE>" and pkg.plan_kernel(cfg, 10) == "cdpr_step_kernel<12, true, true>"
    one_more = text + "  - frame: [0.1, 0.1, 0.6]\n    platform: [0.0, 0.0, 0.0]\n"
    with pytest.raises(ValueError):
        pkg.Config(model=pkg.load_yaml(one_more), batch=1).to_struct()
