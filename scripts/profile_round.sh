#!/bin/bash
# One measurement pass of a round, on the GPU box (via gpurun): bench.py under rocprofv3 (kernel trace, then separate PMC
# passes), the secondary workloads of scripts/profile_driver.py the same way, and profiles/traffic.json rebuilt from what was
# just measured.  Everything lands in gpurun_out/prof_<tag>*/summary; scripts/collect_profiles.py <tag> copies the summaries
# into profiles/.      Usage: scripts/profile_round.sh <tag> [case ...]
set -u
TAG=${1:-r05}; shift || true
CASES=${*:-"lowreg general65k general65k_sw fused rollout config2m2"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
bash $ROOT/scripts/profile_gpu.sh ${TAG}
cp $ROOT/gpurun_out/prof_${TAG}/summary/traffic.json $ROOT/profiles/traffic.json
bash $ROOT/scripts/profile_kernels.sh ${TAG} $CASES
for C in $CASES; do
  S=$ROOT/gpurun_out/prof_${TAG}_$C/summary/${TAG}_${C}_summary.json
  [ -f $S ] || continue
  case $C in
    lowreg) KEY=n8_b524288_spl1 ;;
    general65k) KEY=general_n8_b65536_spl1 ;;
    general) KEY=general_n8_b16384_spl1 ;;
    config2m2) KEY=n4_b4096_spl1 ;;
    *) continue ;;
  esac
  cp $S $ROOT/profiles/${TAG}_${C}_summary.json
  python3 $ROOT/scripts/update_traffic.py $ROOT/profiles/${TAG}_${C}_summary.json $KEY $ROOT/profiles/traffic.json $ROOT/profiles/traffic.json
done
mkdir -p $ROOT/gpurun_out/prof_${TAG}_table
cp $ROOT/profiles/traffic.json $ROOT/gpurun_out/prof_${TAG}_table/traffic.json
