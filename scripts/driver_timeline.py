#!/usr/bin/env python3
"""Per-dispatch timeline of a short bench run under `rocprofv3 --kernel-trace` (the driver's `--steps 20 --warmup 5`):
start offset, duration and the gap to the previous dispatch's end for every dispatch of the step kernel - where the extra
microseconds per step of a 20-step invocation are (first launches after an idle queue, clock ramp, gaps between launches).
Usage: driver_timeline.py <dir with *_kernel_trace.csv> [kernel name substring]"""
import csv, glob, os, sys

d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "cdpr_split_kernel"
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = [i for i, r in enumerate(rows) if pat in r["Kernel_Name"]]
if not first:
    sys.exit(f"no dispatch of {pat} in {f}")
i0 = first[0]
# the first run of consecutive step-kernel dispatches (warm-up + timed steps of the headline leg; other kernels - latches,
# copies - are listed by name when they fall in between)
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = None
n = 0
durs = []
print(f"# {f}\n# dispatch  start_us  dur_us  gap_us  kernel")
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"]
    short = name.split("(")[0][-60:]
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{n:4d} {(s - t0) / 1e3:10.2f} {(e - s) / 1e3:8.2f} {gap:8.2f}  {short}")
    if pat in name:
        durs.append((e - s) / 1e3)
    prev_end = e
    n += 1
    if n >= int(os.environ.get("TIMELINE_MAX", "45")):
        break
print(f"# step-kernel durations: first 5 {['%.2f' % x for x in durs[:5]]}, median of the rest {sorted(durs[5:])[len(durs[5:]) // 2] if len(durs) > 5 else float('nan'):.2f}")
