#!/usr/bin/env python3
"""Instructions ONE world step costs one wave, by class, from the ISA of a several-steps kernel: the straight-line blocks of
the step loop (between labels / branches), longest first.  For a wave that has its SIMD to itself (BASELINE config 2: 128
waves on 1 024 SIMDs) every instruction of every class is an issue slot of ~5 cycles: this count IS the step time.
Usage: step_copy_budget.py file.s kernel-symbol-substring [blocks to show]"""
import re, sys, collections

asm, pat = sys.argv[1], sys.argv[2]
show = int(sys.argv[3]) if len(sys.argv) > 3 else 12
text = open(asm).read().split("\n")
start = next(i for i, l in enumerate(text) if re.match(rf"^\S*{re.escape(pat)}\S*:", l))
end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
CLS = [("VALU packed", r"v_pk_"), ("VALU move/select", r"v_mov_|v_cndmask|v_accvgpr|v_readlane|v_writelane|v_readfirstlane"),
       ("VALU transcendental", r"v_rsq|v_rcp|v_sqrt|v_sin|v_cos|v_exp|v_log"), ("VALU compare", r"v_cmp"), ("VALU other", r"v_"),
       ("SALU", r"s_(?!waitcnt|nop|cbranch|branch|barrier|sleep|endpgm)"), ("wait / nop", r"s_waitcnt|s_nop|s_sleep"),
       ("branch", r"s_cbranch|s_branch"), ("memory", r"buffer_|global_|flat_|scratch_"), ("LDS", r"ds_")]
blocks, cur = [], collections.Counter()
for l in text[start + 1:end + 1]:
    l = l.strip()
    if not l or l.startswith((";", ".", "//")) and not re.match(r"^\.LBB", l):
        continue
    if re.match(r"^\.LBB\S*:", l):
        if sum(cur.values()):
            blocks.append(cur)
        cur = collections.Counter()
        continue
    op = l.split()[0]
    for name, rx in CLS:
        if re.match(rx, op):
            cur[name] += 1
            break
    if re.match(r"s_cbranch|s_branch|s_endpgm|s_setpc", op):
        blocks.append(cur)
        cur = collections.Counter()
blocks = [b for b in blocks if sum(b.values())]
blocks.sort(key=lambda b: -sum(b.values()))
names = [n for n, _ in CLS]
print(f"{pat}: {len(blocks)} straight-line blocks, {sum(sum(b.values()) for b in blocks)} instructions in all; the {show} longest:")
print("  total  " + "  ".join(f"{n:>19s}" for n in names))
for b in blocks[:show]:
    print(f"  {sum(b.values()):5d}  " + "  ".join(f"{b.get(n, 0):19d}" for n in names))
