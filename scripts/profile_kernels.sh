#!/bin/bash
# Runs on the GPU box (via gpurun): for each case of scripts/profile_driver.py, kernel-trace stats and separate PMC
# passes (never --pmc together with a trace).  Usage: scripts/profile_kernels.sh <tag> case [case...]
set -u
TAG=${1:-r02}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for CASE in "$@"; do
  OUT=$ROOT/gpurun_out/prof_${TAG}_$CASE
  mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/scripts/profile_driver.py $CASE > $OUT/trace.log 2>&1
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $ROOT/scripts/profile_driver.py $CASE 40 > $OUT/pmc_$C.log 2>&1
  done
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_SQ -- python3 $ROOT/scripts/profile_driver.py $CASE 40 > $OUT/pmc_SQ.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_MFMA -- python3 $ROOT/scripts/profile_driver.py $CASE 40 > $OUT/pmc_MFMA.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $OUT/pmc_L2 -- python3 $ROOT/scripts/profile_driver.py $CASE 40 > $OUT/pmc_L2.log 2>&1
  python3 $ROOT/scripts/summarize_profile.py $OUT ${TAG}_$CASE > $OUT/summary.log 2>&1
done
