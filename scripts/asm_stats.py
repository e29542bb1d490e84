#!/usr/bin/env python3
"""Instruction mix of the step kernels from hipcc's -S output (static counts, loops counted once)."""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [(m.start(), m.group(1)) for m in re.finditer(r"\n(_ZN4cdpr\w+):", s)]
for (pos, name), nxt in zip(starts, starts[1:] + [(len(s), "")]):
    if pat and pat not in name: continue
    body = s[pos:nxt[0]]
    body = body[: body.find("s_endpgm")]
    lines = [l.strip() for l in body.split("\n")]
    ins = [l for l in lines if l and not l.startswith((".", ";", "//")) and not l.endswith(":") and not l.startswith("_ZN")]
    c = Counter(l.split()[0] for l in ins)
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    print(f"{name}: total {len(ins)} valu {valu} accvgpr {sum(v for k,v in c.items() if 'accvgpr' in k)} "
          f"s_waitcnt {c['s_waitcnt']} s_load {sum(v for k,v in c.items() if k.startswith('s_load'))} "
          f"global {sum(v for k,v in c.items() if k.startswith('global_'))} scratch {sum(v for k,v in c.items() if k.startswith('scratch_'))} "
          f"branches {sum(v for k,v in c.items() if k.startswith('s_cbranch'))}")
    print("   ", c.most_common(16))
