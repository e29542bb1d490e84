#!/usr/bin/env python3
"""Rewrites profiles/traffic.json's entry for one workload from a rocprofv3 PMC summary (scripts/summarize_profile.py):
the HBM bytes per launch of the dominant step kernel, FETCH_SIZE x 2 + WRITE_SIZE as MI355X_MICROARCH.md prescribes for
gfx950.  Called by scripts/profile_gpu.sh right after the PMC passes (on the GPU box the result lands in
gpurun_out/prof_<tag>/summary/traffic.json; copy it over profiles/traffic.json together with the summary), or by hand:
    python scripts/update_traffic.py profiles/r04final_bench_pmc_summary.json n8_b65536_spl1 [traffic_in.json] [traffic_out.json]
tests/test_bench_helpers.py fails when an entry's source is older than the newest *_bench_pmc_summary.json in profiles/."""
import json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dominant_step_kernel(summary):
    """(name, bytes per launch) of the step kernel with the most HBM traffic in a summary (read-out kernels excluded)."""
    best = None
    for name, t in summary.get("traffic_bytes_per_launch", {}).items():
        if "unpack" in name or "publish" in name or "latch" in name:
            continue
        if best is None or t["total"] > best[1]:
            best = (name, t["total"])
    return best


def main():
    src, key = sys.argv[1], sys.argv[2]
    tin = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles", "traffic.json")
    tout = sys.argv[4] if len(sys.argv) > 4 else tin
    summary = json.load(open(src))
    dom = dominant_step_kernel(summary)
    if dom is None:
        raise SystemExit(f"{src}: no FETCH_SIZE / WRITE_SIZE pair for a step kernel")
    table = json.load(open(tin)) if os.path.exists(tin) else {}
    table[key] = dom[1]
    table.setdefault("_sources", {})[key] = {"file": "profiles/" + os.path.basename(src), "kernel": dom[0]}
    # L2 hit rate of the same kernel (separate PMC pass: TCC_HIT_sum / TCC_MISS_sum), keyed without the steps per launch:
    # bench.py's roofline.residency
    c = summary.get("pmc_per_dispatch", {}).get(dom[0], {})
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        hit, miss = c["TCC_HIT_sum"]["mean"], c["TCC_MISS_sum"]["mean"]
        rkey = key.rsplit("_", 1)[0] if key.rsplit("_", 1)[-1].startswith(("spl", "sched")) else key
        table.setdefault("_residency", {})[rkey] = {"tcc_hit_rate": hit / max(hit + miss, 1.0), "file": "profiles/" + os.path.basename(src)}
    json.dump(table, open(tout, "w"), indent=1)
    print(f"{key}: {dom[1]:.0f} B per launch ({dom[0]}) <- {src}")


if __name__ == "__main__":
    main()
