#!/bin/bash
# Experimental builds of ONE unit (its object with extra -D flags, every other object from the regular build directory):
#   scripts/unit_variants.sh <unit> name1 "-DFLAG=.." name2 "-D.."   ->  cdpr-simulation_amd/libcdpr_var_<name>.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); C=$ROOT/cdpr-simulation_amd/csrc
UNIT=$1; shift
make -C $C > /dev/null
X=""; case " k_step k_gen_one k_gen_step k_gen_roll k_gen_step32 k_gen_roll32 " in *" $UNIT "*) X="-mllvm -disable-vector-combine";; esac
case " k_pair " in *" $UNIT "*) X="-fno-slp-vectorize";; esac
pids=()
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( mkdir -p $C/build_uvar_$name
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function $X $flags -c -o $C/build_uvar_$name/$UNIT.o $C/$UNIT.hip
    objs=$(ls $C/build/*.o | grep -v "/$UNIT.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/cdpr-simulation_amd/libcdpr_var_$name.so $objs $C/build_uvar_$name/$UNIT.o
    echo built $name ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
