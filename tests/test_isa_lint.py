"""The shipped kernels must not contain the compiler defect that made results depend on the code layout.

Root cause of VERDICT r04 "What's weak" 2 (found in round 5, scripts/exec_lint.py): on this toolchain
the register allocator sometimes inserts its live-range-split copies (v_accvgpr_write_b32, v_mov_b32, scratch stores) at the
top of the JOIN block of a divergent `if`, in front of the `s_or_b64 exec, exec, <saved>` that ends the region.  The copies
then run under the branch's partial exec mask and the lanes that skipped the branch lose the value.  Whether it happens
depends on register pressure and block layout, not on the source's meaning - so it is checked on the ISA of every
translation unit, with the flags of the shipped build (no GPU needed: hipcc cross-compiles).  The build variants of
tests/test_gpu_build_variants.py are linted by scripts/build_variants.sh when they are built."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lint_recognises_the_pattern(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import exec_lint

    bad = tmp_path / "bad.s"
    bad.write_text(
        "kern:\n"
        "\ts_and_saveexec_b64 s[0:1], vcc\n"
        "\ts_cbranch_execz .LBB0_2\n"
        "; %bb.1:\n"
        "\tds_write_b32 v1, v2\n"
        ".LBB0_2:                                ;   in Loop: Header=BB0_1 Depth=1\n"
        "\tv_accvgpr_write_b32 a3, v7\n"
        "\ts_mov_b32 s12, s64\n"
        "\ts_or_b64 exec, exec, s[0:1]\n"
        "\ts_endpgm\n"
    )
    good = tmp_path / "good.s"
    good.write_text(
        "kern:\n"
        "\ts_and_saveexec_b64 s[0:1], vcc\n"
        "\ts_cbranch_execz .LBB0_2\n"
        "; %bb.1:\n"
        "\tglobal_load_dword v1, v[2:3], off\n"   # a `then` body merged with its join: entered by fall-through only
        "\ts_or_b64 exec, exec, s[0:1]\n"
        "\ts_branch .LBB0_3\n"
        ".LBB0_2:\n"
        "\tv_writelane_b32 v255, s4, 3\n"          # ignores exec: harmless
        "\ts_or_b64 exec, exec, s[0:1]\n"
        "\tv_accvgpr_write_b32 a3, v7\n"
        ".LBB0_3:\n"
        "\ts_endpgm\n"
    )
    assert len(exec_lint.lint(str(bad))) == 1
    assert exec_lint.lint(str(good)) == []
    haz = tmp_path / "haz.s"
    haz.write_text(
        "kern:\n"
        "\tbuffer_store_dwordx4 v[58:61], v191, s[48:51], s5 offen nt\n"
        "\tv_cndmask_b32_e32 v59, v63, v169, vcc\n"           # rule (2): SGPR soffset, data register rewritten at once
        "\tbuffer_store_dwordx4 v[10:13], v191, s[48:51], s5 offen nt\n"
        "\ts_nop 0\n"
        "\tv_mov_b32_e32 v11, v0\n"                            # one wait state: fine
        "\tbuffer_store_dwordx4 v[20:23], v191, s[48:51], 0 offen nt\n"
        "\tv_mov_b32_e32 v21, v0\n"                            # immediate soffset: LLVM's own hazard handling covers it
        "\ts_endpgm\n"
    )
    found = exec_lint.lint_store_hazard(str(haz))
    assert len(found) == 1 and "v59" in found[0][4]


def test_no_vector_work_before_an_exec_restore_in_the_shipped_kernels():
    env = dict(os.environ, EXEC_LINT_JOBS="8")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "exec_lint.py"), "--build"], capture_output=True, text=True, env=env, timeout=1500)
    assert r.returncode == 0, r.stdout[-6000:] + r.stderr[-2000:]
    assert "0 finding(s) in 19 file(s)" in r.stdout
