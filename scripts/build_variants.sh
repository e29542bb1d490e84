#!/bin/bash
# Builds libcdpr_hip.so a few more ways: same sources, semantically neutral changes of the compilation (optimisation level,
# scheduler strategy, where spilled scalars go, a branch-layout hint).  Every variant must produce bit-identical states and
# observables to the shipped build (tests/test_gpu_build_variants.py, scripts/variant_digest.py): a digest that moves
# means a result depends on something undefined (a missing wait, a clobbered lane, a compiler bug).
#
#   scripts/build_variants.sh [name ...]        default: every variant below
#
# Output: cdpr-simulation_amd/libcdpr_hip_var_<name>.so (objects in csrc/build_var_<name>/; both git-ignored, both travel
# to the GPU box with the snapshot).
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/cdpr-simulation_amd/csrc

flags_of() {
  case "$1" in
    o2)      echo "-O2" ;;                                          # after the Makefile's -O3: the last -O wins
    maxilp)  echo "-mllvm -amdgpu-sched-strategy=max-ilp" ;;
    sgprmem) echo "-mllvm -amdgpu-spill-sgpr-to-vgpr=0" ;;          # spilled scalars go to scratch memory, not to VGPR lanes
    expect)  echo "-DCDPR_EXPECT_STEADY" ;;                         # __builtin_expect on the general kernel's steady-state branch
    # diagnosis only (not in the test's list): hypotheses about the expect variant's deviation
    expect_nsa)   echo "-DCDPR_EXPECT_STEADY -fno-strict-aliasing" ;;
    expect_fence) echo "-DCDPR_EXPECT_STEADY -DCDPR_HYP_STEP_FENCE" ;;
    descbuf)      echo "-DCDPR_DESC_PER_BUFFER" ;;
    noslp)        echo "-fno-slp-vectorize" ;;                      # round 6 A/B: every unit without LLVM's SLP vectorizer (the shipped build: k_pair only)                  # round 2's experiment: one buffer descriptor per buffer in store_slot
    *) echo "unknown variant $1" >&2; exit 2 ;;
  esac
}

# variants whose flags only reach the general controller kernels recompile those units and link the shipped build's other objects
units_of() {
  case "$1" in
    expect*) echo "k_gen_one k_gen_split k_gen_step k_gen_roll k_gen_step32 k_gen_roll32" ;;
    *) echo "" ;;
  esac
}

NAMES=${*:-"o2 maxilp sgprmem expect"}
for v in $NAMES; do
  extra=$(flags_of "$v")
  units=$(units_of "$v")
  echo "== variant $v: $extra ${units:+(units: $units)}"
  if [ -n "$units" ]; then
    make -C "$CSRC" all   # the shipped build's objects, up to date
    mkdir -p "$CSRC/build_var_$v"
    for o in "$CSRC"/build/*.o; do
      b=$(basename "$o" .o)
      case " $units " in *" $b "*) ;; *) cp -p "$o" "$CSRC/build_var_$v/" ;; esac
    done
  fi
  # (the variants are test libraries that travel to the GPU box with every snapshot: their code objects are compressed in the fat
  #  binary - 17 MB -> ~9 MB each; the shipped library stays as hipcc makes it)
  make -C "$CSRC" OUT=../libcdpr_hip_var_$v.so OBJDIR=build_var_$v EXTRA="$extra --offload-compress" all
done
