"""ISA lint for two defects of this toolchain on gfx950 that make results depend on code layout and timing.

(1) Vector work placed in front of the exec restore of a join block.

After a divergent `if`, the join block starts with `s_or_b64 exec, exec, <saved mask>` (the end of the control-flow
region).  LLVM's register allocator must insert its live-range-split copies and spills AFTER that instruction; on this
toolchain (ROCm 7.2, clang 22) it sometimes puts them BEFORE it when scalar copies already sit at the top of the block.
The copies (v_accvgpr_write_b32 / v_mov_b32 / scratch stores of values that are live in every lane) then run under the
`then` branch's partial exec mask, and the lanes that skipped the branch later read back whatever the destination register
held before: results that change from run to run and with the code layout (the three "left alone"
anomalies).  This script finds every such place in `hipcc -S` output:

  a basic block that is the target of an `s_cbranch_execz` (the skip edge of a divergent `if`: its join block) in which a
  vector instruction that WRITES a register or memory precedes the `s_or_b64 exec, exec, s[..]`, with no other exec write
  or branch in between.  (A `then` body that was merged with its join block also ends in the exec restore, but it is
  entered by fall-through or `s_cbranch_execnz`, never by the skip edge.)

(2) A 128-bit buffer store whose soffset operand is an SGPR, followed with NO wait state by a VALU write of one of its data
registers.  The ISA manual exempts BUFFER_STORE_DWORDX3/X4 with an SGPR soffset from the "VMEM store > 64 bits, then VALU
write of the write data" hazard and LLVM's hazard recognizer follows it; MI355X does not: now and then lanes 12-15 of every
16 store the NEW register value (scripts/micro/store_hazard.hip -> profiles/r05_store_hazard.txt: 880 of 786 432 lanes with
zero wait states, none with one; with an immediate soffset the part needs two and LLVM inserts two).  Round 2's "one buffer
descriptor per buffer" experiment failed bit-identity through this, and the general controller path's record stores
(GenBuf::store4_if) carried it until round 5.  Found here as: buffer_store_dwordx3/x4 v[a:b], v, s[..], sN ... with a VALU
instruction writing v[a..b] as the very next instruction.

  python scripts/exec_lint.py file.s [file.s ...]       exit code 1 if anything is found
  python scripts/exec_lint.py --build [unit ...]        compile the units of csrc/ to assembly first (default: all)"""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cdpr-simulation_amd", "csrc")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function --cuda-device-only -S"
NOVC_UNITS = {"k_step", "k_gen_one", "k_gen_step", "k_gen_roll", "k_gen_step32", "k_gen_roll32"}
UNITS = ["k_step", "k_pr", "k_onestep", "k_pair", "k_cable", "k_f64", "k_f64_phys", "k_f64_long", "k_f64_hold_long", "k_gen_one", "k_gen_split", "k_gen_step", "k_gen_roll", "k_gen_step32", "k_gen_roll32", "cdpr_engine", "cdpr_engine_f64", "cdpr_engine_rollout", "cdpr_engine_solvers"]

VECTOR = re.compile(r"^(v_|ds_|buffer_|global_|scratch_|flat_)")
HARMLESS = re.compile(r"^(v_cmp|v_cmpx|v_readlane|v_writelane|v_readfirstlane|v_nop|s_)")  # (v_writelane ignores exec)  # write no vector register / memory under exec
EXEC_RESTORE = re.compile(r"^s_or_b64\s+exec,\s*exec,")
EXEC_WRITE = re.compile(r"^s_\w+\s+exec\b|saveexec")


def lint(path):
    findings = []
    func, block, pending = None, None, []
    lines = open(path).read().split("\n")
    skip_targets = set()  # (function, label) reached by an execz skip edge
    f = None
    for raw in lines:
        line = raw.strip()
        m = re.match(r"^([A-Za-z_$][\w.$]*):", line)
        if m and not line.startswith(".L"):
            f = m.group(1)
        m = re.match(r"^s_cbranch_execz\s+(\S+)", line)
        if m:
            skip_targets.add((f, m.group(1)))
    for ln, raw in enumerate(lines, 1):
        line = raw.strip()
        if line.startswith("; %bb."):  # a fall-through block without a label
            block, pending = line[2:].split(":")[0], []
            continue
        line = line.split(";")[0].strip()  # drop trailing comments (".LBB1_2:   ; in Loop ...")
        if not line or line.startswith("//"):
            continue
        m = re.match(r"^([A-Za-z_.$][\w.$]*):$", line)
        if m:  # a label: new basic block (function entry or .LBB)
            if not m.group(1).startswith(".L"):
                func = m.group(1)
            block, pending = m.group(1), []
            continue
        if line.startswith("."):
            continue
        op = line.split()[0]
        if EXEC_RESTORE.match(line):
            if pending and (func, block) in skip_targets:
                findings.append((path, func, block, ln, list(pending)))
            pending = None  # only the block's prologue matters
            continue
        if pending is None:
            continue
        if EXEC_WRITE.search(line) or op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc", "s_barrier", "s_waitcnt")):
            if op.startswith("s_waitcnt"):
                continue
            pending = None
            continue
        if VECTOR.match(op) and not HARMLESS.match(op):
            pending.append((ln, line))
    return findings


def _vregs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def lint_store_hazard(path, need_wait_states=1):
    """Rule (2): (path, function, line, store, overwriting instruction)."""
    findings = []
    func = None
    ins = []
    for ln, raw in enumerate(open(path), 1):
        line = raw.split(";")[0].strip()
        m = re.match(r"^([A-Za-z_$][\w.$]*):", line)
        if m and not line.startswith(".L"):
            func = m.group(1)
        if line.endswith(":"):
            ins.append((ln, func, None))
        elif raw.startswith("\t") and line and not line.startswith("."):
            ins.append((ln, func, line))
    for i, (ln, f, l) in enumerate(ins):
        m = re.match(r"buffer_store_dwordx[34]\s+(v\[\d+:\d+\]),\s*\S+,\s*s\[\d+:\d+\],\s*(s\d+|m0|vcc_lo|vcc_hi)\b", l or "")
        if not m:
            continue
        data, ws = _vregs(m.group(1)), 0
        for j in range(i + 1, min(i + 6, len(ins))):
            lj = ins[j][2]
            if lj is None or ws >= need_wait_states:
                break
            op = lj.split()[0]
            if op == "s_nop":
                ws += int(lj.split()[1]) + 1
                continue
            if op.startswith("v_") and not op.startswith(("v_cmp", "v_readlane", "v_readfirstlane")) and _vregs(lj.split()[1].rstrip(",")) & data:
                findings.append((path, f, ln, l, lj))
            ws += 1
    return findings


def build(units, outdir="/tmp/exec_lint"):
    os.makedirs(outdir, exist_ok=True)

    def one(u):
        out = os.path.join(outdir, u + ".s")
        novc = ["-mllvm", "-disable-vector-combine"] if u in NOVC_UNITS else []  # (as the Makefile builds these units)
        r = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS.split(), *novc, *os.environ.get("EXEC_LINT_EXTRA", "").split(), "-o", out, u + ".hip"], cwd=CSRC, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"{u}: {r.stderr[-2000:]}")
        return out

    with ThreadPoolExecutor(int(os.environ.get("EXEC_LINT_JOBS", "6"))) as ex:
        return list(ex.map(one, units))


def main(argv):
    if argv and argv[0] == "--build":
        files = build(argv[1:] or UNITS)
    else:
        files = argv
    total = 0
    for f in files:
        found = lint(f)
        total += len(found)
        for path, func, block, ln, pend in found:
            dn = subprocess.run(["c++filt", func or "?"], capture_output=True, text=True).stdout.strip()
            print(f"{os.path.basename(path)}:{ln}: {dn[:120]} [{block}]: {len(pend)} vector instruction(s) before the exec restore, e.g. `{pend[0][1]}`")
        haz = lint_store_hazard(f)
        total += len(haz)
        for path, func, ln, store, over in haz:
            dn = subprocess.run(["c++filt", func or "?"], capture_output=True, text=True).stdout.strip()
            print(f"{os.path.basename(path)}:{ln}: {dn[:120]}: `{store}` then `{over}` with no wait state (SGPR soffset: LLVM inserts none)")
    print(f"exec_lint: {total} finding(s) in {len(files)} file(s)")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
