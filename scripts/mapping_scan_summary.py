"""rocprofv3 CSVs of scripts/mapping_scan.py -> one table: per (config, batch, mapping) the average dispatch duration and
the vector instructions per robot.  Usage: mapping_scan_summary.py <trace_dir> <pmc_dir>"""
import csv, glob, os, sys
from collections import defaultdict

def rows(d, pat):
    for f in glob.glob(os.path.join(d, "**", pat), recursive=True):
        yield from csv.DictReader(open(f))

def short(name):
    for key, tag in (("cdpr_step_kernel_cable", "cable"), ("cdpr_step_kernel_pair", "pair"), ("cdpr_split_kernel", "robot"), ("cdpr_onestep_kernel", "robot"), ("cdpr_step_kernel<", "robot")):
        if key in name:
            return tag, ("n8" if "<8" in name or "(8" in name or "8," in name.split("<")[1][:3] else "n4")
    return None, None

dur = defaultdict(list)
for r in rows(sys.argv[1], "*kernel_trace.csv"):
    tag, cfg = short(r["Kernel_Name"])
    if tag is None:
        continue
    grid = int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0)
    wg = int(r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or 64)
    dur[(cfg, tag, grid, wg)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
valu = defaultdict(lambda: defaultdict(list))
if len(sys.argv) > 2:
    for r in rows(sys.argv[2], "*counter_collection.csv"):
        tag, cfg = short(r["Kernel_Name"])
        if tag is None:
            continue
        grid = int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0)
        wg = int(r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or 64)
        valu[(cfg, tag, grid, wg)][r["Counter_Name"]].append(float(r["Counter_Value"]))
robots_per_wg = {("n8", "robot"): 64, ("n8", "pair"): 32, ("n8", "cable"): 8, ("n4", "robot"): 64, ("n4", "pair"): 32, ("n4", "cable"): 16}
print(f"{'config':6} {'batch':>7} {'mapping':8} {'us/launch':>10} {'min':>7} {'calls':>6} {'waves':>7} {'VALU/64rob':>11}")
for (cfg, tag, grid, wg), d in sorted(dur.items(), key=lambda kv: (kv[0][0], kv[0][2] // kv[0][3] * robots_per_wg[(kv[0][0], kv[0][1])], kv[0][1])):
    nwg = grid // wg
    batch_hi = nwg * robots_per_wg[(cfg, tag)]
    d = sorted(d)[len(d) // 10:]  # drop nothing but keep order stable; report mean of the steady 90 %
    c = valu.get((cfg, tag, grid, wg), {})
    insts = (sum(c["SQ_INSTS_VALU"]) / len(c["SQ_INSTS_VALU"])) if c.get("SQ_INSTS_VALU") else float("nan")
    waves = (sum(c["SQ_WAVES"]) / len(c["SQ_WAVES"])) if c.get("SQ_WAVES") else float("nan")
    print(f"{cfg:6} {'<=' + str(batch_hi):>7} {tag:8} {sum(d) / len(d):10.2f} {d[0]:7.2f} {len(d):6d} {waves:7.0f} {insts * 64 / max(batch_hi, 1):11.1f}")
