"""The C-ABI library loads on a CPU-only box, exports every symbol include/cdpr.h declares, agrees with the
ctypes mirror on the struct layout, and refuses to run without a GPU (there is no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "cdpr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cdpr_[a-z0-9_]+)\s*\(", text)))


def test_exports_every_declared_symbol(pkg):
    from cdpr_simulation_amd._native import EXPORTS, lib

    L = lib()
    declared = header_functions()
    assert declared, "no declarations parsed from include/cdpr.h"
    for name in declared:
        assert hasattr(L, name), f"libcdpr_hip.so does not export {name}"
    assert sorted(EXPORTS) == declared  # the Python prototype table covers the whole header


def test_struct_layout_and_version(pkg):
    from cdpr_simulation_amd._native import lib

    assert lib().cdpr_abi_version() == pkg._abi.ABI_VERSION
    assert lib().cdpr_config_size() == C.sizeof(pkg._abi.ConfigStruct)


def test_bytes_per_state_step_formula(pkg):
    from cdpr_simulation_amd._native import lib

    assert lib().cdpr_bytes_per_state_step(C.byref(pkg.Config().to_struct())) == 604  # n = 4
    assert lib().cdpr_bytes_per_state_step(C.byref(pkg.Config(model=pkg.eight_cable_model()).to_struct())) == 1052


def test_derivative_weights_rejects_bad_arguments(pkg):
    with pytest.raises(pkg.CdprError):
        pkg.derivative_weights(3, 3)
    with pytest.raises(pkg.CdprError):
        pkg.derivative_weights(100, 2)


def test_invalid_configuration_is_rejected_before_touching_the_gpu(pkg):
    from cdpr_simulation_amd._native import lib

    s = pkg.Config().to_struct()
    s.n_cables = 13  # PLG.cpp:167-168: wrong joint count throws at Load (the engine takes 1..12)
    h = C.c_void_p()
    assert lib().cdpr_create(C.byref(s), 0, C.byref(h)) == pkg._abi.ERR_INVALID
    assert b"invalid joint count" in lib().cdpr_last_error(None)
    s = pkg.Config().to_struct()
    s.stages = pkg._abi.STAGE_FK
    assert lib().cdpr_create(C.byref(s), 0, C.byref(h)) == pkg._abi.ERR_INVALID  # FK needs >= 6 cables
    with pytest.raises(ValueError):
        pkg.Config(stages=pkg._abi.STAGE_TD).to_struct()


def test_no_cpu_fallback(pkg):
    """Without a GPU cdpr_create fails loudly; it never computes on the host."""
    from cdpr_simulation_amd._native import lib

    if lib().cdpr_device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(pkg.CdprError) as ei:
        pkg.Engine(pkg.Config())
    assert ei.value.code == pkg._abi.ERR_DEVICE and "no HIP device" in str(ei.value)


def test_product_does_not_reference_the_oracle():
    """Nothing under the package or the C-ABI sources may import, include or link oracle/."""
    pkg_dir = os.path.join(ROOT, "cdpr-simulation_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower().replace("no cpu fallback", ""), f"{f} mentions the oracle"


def test_header_is_valid_c99_and_the_c_example_links(tmp_path):
    """include/cdpr.h is a C header (not only C++): examples/c_abi_demo.c compiles with gcc -std=c99 -pedantic without a
    warning and links against the library; without a GPU the program fails loudly at cdpr_create (no CPU fallback)."""
    import subprocess

    exe = tmp_path / "c_abi_demo"
    libdir = os.path.join(ROOT, "cdpr-simulation_amd")
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O2", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "c_abi_demo.c"), "-L" + libdir, "-lcdpr_hip", "-lm", "-Wl,-rpath," + libdir, "-o", str(exe)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    from cdpr_simulation_amd._native import lib

    if lib().cdpr_device_count() == 0:
        run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
        assert run.returncode == 1 and "cdpr_create" in run.stderr and not run.stdout
