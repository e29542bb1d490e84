#!/usr/bin/env python3
"""Vector-instruction budget of a step kernel BY SOURCE REGION.

The translation unit is compiled to assembly with line tables (-gline-tables-only: same code, `.loc` directives added);
every instruction is attributed to the device function its `.loc` line lies in (the innermost inlined callee: `.loc` carries the
callee's file and line), functions are grouped into the step's regions, and instructions inside the Newton loop are
weighted by its trip count (4).  Printed per region: vector instructions per step (packed / plain / transcendental / moves and
selects / accumulation-register moves), next to the arithmetic MINIMUM of the region - the packed fma / mul / add count
its formulas need when every pair of cables shares an instruction (MINIMUM below: counted from the formulas of DESIGN.md section 1).

  python scripts/region_budget.py [unit] [kernel-substring]      default: k_step, the rollout kernel <8, FK, TD>
  -> profiles/r05_region_budget.txt"""
import os
import re
import subprocess
import sys
from collections import Counter, defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cdpr-simulation_amd", "csrc")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function --cuda-device-only -S -gline-tables-only -DCDPR_LEAF_NODEBUG"

# device function -> region of the step
REGION_OF = [
    (r"ik_pairs|ik_rows|quat_to_rot", "IK / structure matrix"),
    (r"pid_pairs|pid_coef|ring_push|ring_row", "PID (FIR + terms)"),
    (r"normal_matrix|gram_partial|jt_times|jt_partial", "J^T J and J^T r"),
    (r"chol_|normal_solve|hsum", "Cholesky + substitutions"),
    (r"quat_apply_rotvec", "quaternion update (FK)"),
    (r"integrate|apply_travel_stop", "world step"),
    (r"store_slot|load_slot|row_to_lds", "row loads / stores"),
    (r"travel_mask|pack_flags|step_published|sched_wait|mailbox_wait", "flags / bookkeeping"),
]
# arithmetic minimum per step at n = 8 (4 cable pairs), packed where two cables share an instruction
MINIMUM = {
    "IK / structure matrix": "6 evaluations x (4 pairs x 27 pk + 18 rot) = 756",
    "PID (FIR + terms)": "4 pairs x (11 FIR + 14 terms) = 100",
    "J^T J and J^T r": "5 x (21 x 4 pk fma + 21 hsum) + 6 x (6 x 4 + 6) = 705",
    "Cholesky + substitutions": "5 x (22 pk + 12 pk scale + 6 rsq + 27 hsum-free subst ~ 70) = 350",
    "quaternion update (FK)": "4 x ~40 = 160",
    "world step": "~90",
}


def function_ranges(path):
    """[(first_line, last_line, name)] of the device functions of a header, from a crude brace scan."""
    out = []
    lines = open(path).read().split("\n")
    i = 0
    while i < len(lines):
        m = re.match(r"^(?:template\s*<[^>]*>\s*)?(?:CDPR_DEV|__device__|__global__|static __global__)[^;{]*?\b(\w+)\s*\(", lines[i])
        if not m and i + 1 < len(lines) and lines[i].startswith("template"):
            m = re.match(r"^(?:CDPR_DEV|__device__|__global__)[^;{]*?\b(\w+)\s*\(", lines[i + 1])
        if m:
            name = m.group(1)
            depth, j, seen = 0, i, False
            while j < len(lines):
                depth += lines[j].count("{") - lines[j].count("}")
                seen = seen or "{" in lines[j]
                if seen and depth <= 0:
                    break
                j += 1
            out.append((i + 1, j + 1, name))
            i = j + 1
        else:
            i += 1
    return out


def main():
    unit = sys.argv[1] if len(sys.argv) > 1 else "k_step"
    pat = sys.argv[2] if len(sys.argv) > 2 else "cdpr_step_kernelILi8ELb1ELb1ELb0ELb1ELb0ELb0ELb0"  # the rollout kernel, n = 8, FK + TD
    asm = f"/tmp/region_{unit}.s"
    # the unit's own flags (csrc/Makefile: NOVC_UNITS, NOSLP_UNITS), so that the budget is that of the shipped code
    mk = open(os.path.join(CSRC, "Makefile")).read()
    extra = []
    for var, flags in (("NOVC_UNITS", ["-mllvm", "-disable-vector-combine"]), ("NOSLP_UNITS", ["-fno-slp-vectorize"])):
        m = re.search(rf"^{var} := (.*)$", mk, re.M)
        if m and unit in m.group(1).split():
            extra += flags
    r = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS.split(), *extra, "-o", asm, unit + ".hip"], cwd=CSRC, capture_output=True, text=True)
    if r.returncode != 0:
        sys.exit(r.stderr[-2000:])
    text = open(asm).read().split("\n")
    files = {}
    for l in text:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[int(m.group(1))] = os.path.basename(m.group(3) or m.group(2))
    ranges = {f: function_ranges(os.path.join(CSRC, f)) for f in set(files.values()) if os.path.exists(os.path.join(CSRC, f))}
    start = next(i for i, l in enumerate(text) if l.startswith("_ZN4cdpr") and pat in l)
    end = next(i for i in range(start, len(text)) if text[i].startswith(".Lfunc_end"))
    # the Newton loop: the innermost loop that contains v_rsq (the Cholesky) - blocks between its header label and back edge
    body = text[start:end]
    loop_w = [1] * len(body)
    labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    loops = []
    for i, l in enumerate(body):
        m = re.match(r"\s+s_(?:cbranch_\w+|branch)\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:  # a back edge
            lo = labels[m.group(1)]
            if sum(1 for x in body[lo:i] if "v_rsq_f32" in x) >= 6:  # holds a 6 x 6 Cholesky
                loops.append((i - lo, lo, i))
    if loops:  # the smallest such loop is the Newton iteration (the step loop around it holds it too)
        _, lo, hi = min(loops)
        for k in range(lo, hi + 1):
            loop_w[k] = 4  # fk_max_iterations of the contract
    cur = ("?", 0)
    sources = {}
    last_fn = "kernel: prologue (loads, Joy, window)"
    per_fn = defaultdict(Counter)
    for i, l in enumerate(body):
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
            continue
        t = l.strip()
        if not l.startswith("\t") or not t or t.startswith((".", ";")):
            continue
        op = t.split()[0]
        if not op.startswith("v_"):
            continue
        fn = "kernel body"
        if cur[1] and not os.path.exists(os.path.join(CSRC, cur[0])):  # a HIP header's inline (rsq, fmin, ...): the region around it
            per_fn[last_fn][("accvgpr" if "accvgpr" in op else "packed" if op.startswith("v_pk_") else "transcendental" if re.match(r"v_(rsq|rcp|sqrt|sin|cos|exp|log)", op)
                             else "move/select" if re.match(r"v_(mov|cndmask|readlane|writelane|readfirstlane|perm|swap)", op) else "compare" if op.startswith("v_cmp") else "plain")] += loop_w[i]
            continue
        if cur[1] == 0:
            fn = "(no line: merged / hoisted by the optimiser)"
        for lo, hi, name in ranges.get(cur[0], []) if cur[1] else []:
            if lo <= cur[1] <= hi:
                fn = name
        if cur[1] and (fn in ("__launch_bounds__", "kernel body") or fn.startswith("cdpr_")):  # the kernel itself: by its "// ----" section headers
            src = sources.setdefault(cur[0], open(os.path.join(CSRC, cur[0])).read().split("\n") if os.path.exists(os.path.join(CSRC, cur[0])) else [])
            fn = "kernel: prologue (loads, Joy, window)"
            for k in range(min(cur[1], len(src)) - 1, 0, -1):
                m2 = re.match(r"\s*// ---- (.*)", src[k])
                if m2:
                    fn = "kernel: " + re.split(r"[(:\[]", m2.group(1))[0].strip()[:44]
                    break
                if re.search(r"__global__", src[k]):
                    break
        kind = ("accvgpr" if "accvgpr" in op else "packed" if op.startswith("v_pk_") else "transcendental" if re.match(r"v_(rsq|rcp|sqrt|sin|cos|exp|log)", op)
                else "move/select" if re.match(r"v_(mov|cndmask|readlane|writelane|readfirstlane|perm|swap)", op) else "compare" if op.startswith("v_cmp") else "plain")
        per_fn[fn][kind] += loop_w[i]
        if cur[1]:
            last_fn = fn
    regions = defaultdict(Counter)
    for fn, c in per_fn.items():
        reg = next((name for rx, name in REGION_OF if re.search(rx, fn)), fn if fn.startswith("kernel:") else "other: " + fn)
        regions[reg].update(c)
    total = sum(sum(c.values()) for c in regions.values())
    print(f"{unit}: {[l for l in body[:1]][0].split(':')[0]}")
    print(f"vector instructions per step (Newton loop x 4), by source region; total {total}")
    print(f"{'region':46s} {'all':>6s} {'packed':>7s} {'plain':>6s} {'transc':>6s} {'mov/sel':>7s} {'accvgpr':>7s} {'cmp':>5s}   arithmetic minimum")
    for reg, c in sorted(regions.items(), key=lambda kv: -sum(kv[1].values())):
        print(f"{reg:46s} {sum(c.values()):6d} {c['packed']:7d} {c['plain']:6d} {c['transcendental']:6d} {c['move/select']:7d} {c['accvgpr']:7d} {c['compare']:5d}   {MINIMUM.get(reg, '')}")
    print("by device function:")
    for fn, c in sorted(per_fn.items(), key=lambda kv: -sum(kv[1].values()))[:18]:
        print(f"   {fn:32s} {sum(c.values()):6d}  {dict(c)}")


if __name__ == "__main__":
    main()
