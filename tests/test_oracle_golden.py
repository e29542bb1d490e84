"""Pins the CPU oracle against the golden vectors (tests/golden/, made by make_golden.py
from the reference's own data files, its transformations.py and its Filter.h)."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return json.load(open(os.path.join(GOLD, name)))


def test_biquad_matches_reference_filter_h(oracle):
    """oracle BiQuad restatement == reference Filter.h BiQuad<double>, bit for bit."""
    for case in load("biquad.json")["cases"]:
        f = oracle.OracleBiquad(case["fc"], case["fs"], case["q"])
        if "preset" in case:
            f.set_value(case["preset"])
        y = [f.process(x) for x in case["input"]]
        assert y == case["output"], (case["fc"], case["q"], case["input_name"])


@pytest.mark.skipif(not os.path.exists("/root/reference"), reason="reference tree only exists in the build container")
def test_biquad_against_live_reference_build(oracle):
    """Same check against oracle/_ref built from the header where it lies (not only the fixture)."""
    rng = np.random.default_rng(3)
    ref, mine = oracle.RefBiquad(0.13, 1.0, 0.9), oracle.OracleBiquad(0.13, 1.0, 0.9)
    for x in rng.standard_normal(200):
        assert ref.process(float(x)) == mine.process(float(x))


def test_model_constants_match_reference_files(pkg):
    """cube_model() transcribes cube.yaml anchors and cube.sdf spawn pose / inertial / joint constants."""
    g = load("cube_model.json")
    m = pkg.cube_model()
    for i, pt in enumerate(g["yaml"]["points"]):
        assert list(m.frame_anchors[i]) == pt["frame"]
        assert list(m.platform_anchors[i]) == pt["platform"]
    assert list(m.home_position) == g["sdf_platform_pose"][:3]  # 0 0 0.3, not cube.yaml's z = 2
    inert = g["sdf_platform_inertia"]
    assert m.mass == inert["mass"]
    assert list(m.inertia) == [inert[k] for k in ("ixx", "iyy", "izz", "ixy", "ixz", "iyz")]
    for c in g["sdf_cables"]:
        assert m.joint_damping == c["damping"] and m.effort_limit == c["effort"]
    assert m.f_min == g["yaml"]["joints"]["actuated"]["min"] and m.f_max == g["yaml"]["joints"]["actuated"]["effort"]


def test_ik_matches_generator_geometry(pkg, oracle):
    """oracle IK at the spawn pose == gen_cdpr.py:113-118 evaluated with the reference's transformations.py,
    and == the numbers the generator wrote into cube.sdf (6 printed digits)."""
    geo, sdf = load("geometry.json"), load("cube_model.json")
    cfg = pkg.Config()
    s = cfg.to_struct()
    q, qd, ln, jac = oracle.ik(s, cfg.model.home_pose())
    assert np.allclose(q, 0.0, atol=1e-15)
    for i, c in enumerate(geo["cables"]):
        assert abs(ln[i] - c["length"]) < 1e-15
        assert np.allclose(jac[i], c["jacobian_row"], atol=1e-15)
        assert abs(s.cable_ref_length[i] - c["length"]) < 1e-15
        # prismatic axis in cube.sdf = -u scaled by 0.15 (hand edit; Gazebo normalises)
        ax = np.array(sdf["sdf_cables"][i]["axis_xyz"])
        assert np.allclose(ax / np.linalg.norm(ax), -jac[i, :3], atol=2e-6)
        # link pose written by the generator: cp = pp - a (pp - fp), rpy from euler_from_matrix
        assert np.allclose(sdf["sdf_cables"][i]["link_pose"][:3], c["link_position"], atol=1e-6)
        assert np.allclose(sdf["sdf_cables"][i]["link_pose"][3:], c["rpy"], atol=1e-6)
    assert np.linalg.matrix_rank(jac) == geo["rank_J_home"] == 3


def test_survey_home_geometry_values(pkg, oracle):
    kat = load("pid_kat.json")["home_geometry"]
    cfg = pkg.Config()
    q, qd, ln, jac = oracle.ik(cfg.to_struct(), cfg.model.home_pose())
    assert np.allclose(ln, kat["L0"], atol=1e-9)
    assert np.allclose(jac[0], kat["J_row0"], atol=1e-9)
    assert abs(9.8 / (-jac[:, 2].sum()) - kat["static_tension"]) < 1e-9
    c8 = pkg.Config(model=pkg.eight_cable_model())
    j8 = oracle.ik(c8.to_struct(), c8.model.home_pose())[3]
    sv = np.linalg.svd(j8, compute_uv=False)
    assert np.allclose(sv, kat["eight_cable_singular_values"], rtol=2e-3)
    assert np.linalg.matrix_rank(j8) == 6
