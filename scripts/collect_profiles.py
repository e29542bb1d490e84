#!/usr/bin/env python3
"""Copies what scripts/profile_round.sh left under gpurun_out/ (scratch) into profiles/ (tracked): per workload the rocprofv3
kernel-stats CSV and the PMC summary, the bench line that ran under the tracer, and the traffic table rebuilt from them.
    python scripts/collect_profiles.py <tag>"""
import glob, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
out = os.path.join(ROOT, "profiles")
n = 0
for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{tag}*"))):
    for f in glob.glob(os.path.join(d, "summary", "*")):
        b = os.path.basename(f)
        if b == "traffic.json" or b.endswith("_summary.json") and b.count("bench_pmc") == 0 and d.endswith(f"prof_{tag}"):
            continue  # (the bench's summary is kept under its *_bench_pmc_summary.json name; the table comes from prof_<tag>_table)
        shutil.copy(f, os.path.join(out, b)); n += 1
    bt = os.path.join(d, "bench_trace.json")
    if os.path.exists(bt) and os.path.getsize(bt) > 0:
        shutil.copy(bt, os.path.join(out, f"{tag}_bench_under_rocprofv3.json")); n += 1
t = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_table", "traffic.json")
if os.path.exists(t):
    shutil.copy(t, os.path.join(out, "traffic.json")); n += 1
print(f"{n} files -> profiles/")
