#!/bin/bash
# Dynamic instruction mix and wait cycles of the fp64 role-split kernel (two counter passes per case), into gpurun_out/f64_counters.txt
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/f64_counters; mkdir -p $OUT
for CASE in "64 -1.0" "64 0.001" "65536 -1.0" "65536 0.001"; do
  set -- $CASE; TAG=b$1_$( [ "$2" = "-1.0" ] && echo plain || echo hold )
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES -d $OUT/${TAG}_a -o c --output-format csv -- python3 $ROOT/scripts/f64_counters.py $1 $2 > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU -d $OUT/${TAG}_b -o c --output-format csv -- python3 $ROOT/scripts/f64_counters.py $1 $2 > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
out = open("$ROOT/gpurun_out/f64_counters.txt", "w")
for d in sorted(glob.glob("$OUT/*")):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "split_kernel_f64" in r.get("Kernel_Name", ""):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    line = d.split("/")[-1] + ": " + ", ".join(f"{k} {sum(v[-100:]) / len(v[-100:]):.0f}" for k, v in sorted(acc.items()))
    print(line); out.write(line + "\n")
PY
