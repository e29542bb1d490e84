"""Static instruction counts of the role-split fp64 kernel BY PHASE: the translation unit compiled with -DCDPR_STAMPS to assembly, the
instructions between consecutive stamp sites (CDPR_F64_STAMP: s_memrealtime + a store whose offset is the stamp's index x 8) counted
by class.  Loop bodies count once (trip counts in the notes).   python scripts/f64_phase_budget.py <asm> <kernel symbol>"""
import re, sys
NAMES = {0: "est: entry", 1: "est: rows requested", 2: "est: measured lengths done", 3: "est: Newton + TD matrix and factor done", 4: "est: passed barrier 1", 5: "est: tensions out, exit",
         6: "(shader-clock stamp)", 7: "(shader-clock stamp)", 8: "ctl: entry", 9: "ctl: first rows requested", 10: "ctl: IK + PID done, forces out", 11: "ctl: passed barrier 2",
         12: "ctl: limits, late observables, world step done", 13: "ctl: stores issued", 14: "ctl: stores acknowledged"}
asm = open(sys.argv[1]).read().split("\n")
name = sys.argv[2]
start = next(i for i, l in enumerate(asm) if l.startswith(name + ":"))
end = next(i for i in range(start, len(asm)) if asm[i].startswith(".Lfunc_end"))
ins = []
for l in asm[start:end]:
    s = l.strip()
    if re.match(r"^([.\w$]+):", s) or not s or s[0] in ";.":
        continue
    ins.append(s)
sites = []
for k, s in enumerate(ins):
    if s.startswith("s_memrealtime") or s.startswith("s_memtime"):
        for j in range(k + 1, min(k + 16, len(ins))):
            if "global_store" in ins[j]:
                m = re.search(r"offset:(\d+)", ins[j])
                sites.append((k, (int(m.group(1)) if m else 0) // 8))
                break
print(f"{name}: {len(ins)} instructions; segments in program order (the compiler lays the controller wave's code out first)")
prev_k, prev_i = 0, None
for k, i in sites + [(len(ins), None)]:
    seg = ins[prev_k:k]
    c = dict(valu=0, salu=0, vmem=0, lds=0, wait=0, branch=0)
    for s in seg:
        op = s.split()[0]
        if op.startswith("s_waitcnt") or op == "s_nop": c["wait"] += 1
        elif op.startswith("s_cbranch") or op == "s_branch" or op.startswith("s_swappc") or op.startswith("s_setpc"): c["branch"] += 1
        elif op.startswith("v_"): c["valu"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): c["vmem"] += 1
        elif op.startswith("ds_"): c["lds"] += 1
        else: c["salu"] += 1
    a = "kernel entry" if prev_i is None else f"{prev_i:2d} {NAMES.get(prev_i, '?')}"
    b = "end of code" if i is None else f"{i:2d} {NAMES.get(i, '?')}"
    if len(seg) > 24:
        print(f"  {len(seg):5d}  [{a}] -> [{b}]: " + ", ".join(f"{v} {n}" for n, v in c.items()))
    prev_k, prev_i = k, i
