#!/usr/bin/env python3
"""Condenses the rocprofv3 output of scripts/profile_gpu.sh into small CSV/JSON summaries (copied to profiles/)."""
import csv, glob, json, os, sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]
summary = {"tag": tag}

def find(pattern):
    f = glob.glob(os.path.join(out, pattern), recursive=True)
    return f[0] if f else None

ks = find("trace/**/*kernel_stats.csv")
if ks:
    rows = list(csv.DictReader(open(ks)))
    summary["kernel_stats"] = [{k: r[k] for k in ("Name", "Calls", "AverageNs", "MinNs", "MaxNs", "Percentage")} for r in rows]
    os.makedirs(os.path.join(out, "summary"), exist_ok=True)
    open(os.path.join(out, "summary", f"{tag}_kernel_stats.csv"), "w").write(open(ks).read())

counters = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc_*/**/*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if "cdpr_" not in name:
            continue
        counters[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
pm = {}
for kname, cs in counters.items():
    pm[kname] = {c: {"mean": sum(v) / len(v), "n": len(v)} for c, v in cs.items()}
summary["pmc_per_dispatch"] = pm
# HBM traffic per launch, corrected as MI355X_MICROARCH.md prescribes for gfx950:
# FETCH_SIZE (KiB) counts 128-B requests at 64 B -> x2 for wide coalesced reads; WRITE_SIZE (KiB) is exact.
for kname, cs in pm.items():
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        rd = cs["FETCH_SIZE"]["mean"] * 1024 * 2
        wr = cs["WRITE_SIZE"]["mean"] * 1024
        summary.setdefault("traffic_bytes_per_launch", {})[kname] = {"read_corrected": rd, "write": wr, "total": rd + wr}
json.dump(summary, open(os.path.join(out, "summary", f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1)[:3000])
