#!/bin/bash
# Experimental builds of the fp64 unit only (k_f64.o with extra -D flags, every other object from the regular build directory):
#   scripts/f64_variants.sh name1 "-DFLAG=.." name2 "-D.."   ->  cdpr-simulation_amd/libcdpr_f64var_<name>.so   (A/B: scripts/f64_lib_ab.py)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); C=$ROOT/cdpr-simulation_amd/csrc
make -C $C > /dev/null
pids=()
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( mkdir -p $C/build_f64var_$name
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function $flags -c -o $C/build_f64var_$name/k_f64.o $C/k_f64.hip
    objs=$(ls $C/build/*.o | grep -v k_f64.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/cdpr-simulation_amd/libcdpr_f64var_$name.so $objs $C/build_f64var_$name/k_f64.o
    echo built $name ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
