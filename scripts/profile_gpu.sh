#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + separate PMC passes for bench.py's timed kernel.
# Usage: scripts/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 300 --warmup 30 --no-cpu-baseline --no-secondary $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.log
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc_$C.log
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_SQ -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc_SQ.log
rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_L2 -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc_L2.log
python3 $ROOT/scripts/summarize_profile.py $OUT $TAG
# the traffic table bench.py reads, rewritten from the PMC passes just run (default workload: config 3; --config 2 etc.: pass TRAFFIC_KEY).
# Copy summary/traffic.json over profiles/traffic.json together with summary/${TAG}_summary.json -> profiles/${TAG}_bench_pmc_summary.json.
cp $OUT/summary/${TAG}_summary.json $OUT/summary/${TAG}_bench_pmc_summary.json
python3 $ROOT/scripts/update_traffic.py $OUT/summary/${TAG}_bench_pmc_summary.json ${TRAFFIC_KEY:-n8_b65536_spl1} $ROOT/profiles/traffic.json $OUT/summary/traffic.json
