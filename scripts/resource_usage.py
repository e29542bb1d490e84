"""Register / scratch / spill table of every kernel in the given translation units, from hipcc's
-Rpass-analysis=kernel-resource-usage remarks (no GPU needed).

  python scripts/resource_usage.py [--all] [--flags "..."] k_step k_onestep ...

Default: only kernels with scratch or spills are listed; --all lists every kernel.  Output is one line per kernel:
demangled name, SGPRs, VGPRs, AGPRs, scratch bytes per lane, occupancy, spilled SGPRs / VGPRs, LDS bytes per block."""
import argparse
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cdpr-simulation_amd", "csrc")
BASE_FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function"
UNITS = ["k_step", "k_pr", "k_onestep", "k_pair", "k_cable", "k_f64", "k_f64_phys", "k_f64_long", "k_f64_hold_long", "k_gen_one", "k_gen_split", "k_gen_step", "k_gen_roll", "k_gen_step32", "k_gen_roll32"]


NOVC_UNITS = {"k_step", "k_gen_one", "k_gen_step", "k_gen_roll", "k_gen_step32", "k_gen_roll32"}  # (the Makefile's: built without VectorCombine)


def remarks(unit, flags):
    extra = ["-mllvm", "-disable-vector-combine"] if unit in NOVC_UNITS else []
    cmd = ["/opt/rocm/bin/hipcc", *flags.split(), *extra, "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null", unit + ".hip"]
    return subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr


def parse(text):
    out = []
    for block in re.split(r"remark: Function Name: ", text)[1:]:
        name = block.split()[0]

        def g(key):
            m = re.search(re.escape(key) + r": (\d+)", block)
            return int(m.group(1)) if m else -1

        out.append(dict(name=name, sgpr=g("TotalSGPRs"), vgpr=g("VGPRs"), agpr=g("AGPRs"), scratch=g("ScratchSize [bytes/lane]"), occ=g("Occupancy [waves/SIMD]"),
                        spill_s=g("SGPRs Spill"), spill_v=g("VGPRs Spill"), lds=g("LDS Size [bytes/block]")))
    return out


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return r.stdout.split("\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("units", nargs="*", default=UNITS)
    ap.add_argument("--all", action="store_true")
    ap.add_argument("--flags", default=BASE_FLAGS)
    ap.add_argument("--jobs", type=int, default=6)
    a = ap.parse_args()
    with ThreadPoolExecutor(a.jobs) as ex:
        texts = list(ex.map(lambda u: remarks(u, a.flags), a.units))
    seen = set()
    for unit, text in zip(a.units, texts):
        rows = [r for r in parse(text) if r["name"] not in seen]
        seen.update(r["name"] for r in rows)
        names = demangle([r["name"] for r in rows])
        shown = 0
        for r, dn in zip(rows, names):
            if not a.all and r["scratch"] <= 0 and r["spill_s"] <= 0 and r["spill_v"] <= 0:
                continue
            dn = re.sub(r"^void cdpr::", "", dn).replace("cdpr::", "")
            dn = re.sub(r"\((StepArgs|F64Args|GenCtl|SolveArgs|[A-Za-z0-9_]+Args)(, GenCtl)?\)$", "", dn)
            print(f"{unit:13s} {dn[:84]:84s} S{r['sgpr']:<4d} V{r['vgpr']:<4d} A{r['agpr']:<4d} scratch {r['scratch']:<5d} occ {r['occ']} spillS {r['spill_s']:<4d} spillV {r['spill_v']:<4d} lds {r['lds']}")
            shown += 1
        print(f"# {unit}: {len(rows)} kernels, {shown} listed", file=sys.stderr)


if __name__ == "__main__":
    main()
