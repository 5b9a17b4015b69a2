#!/bin/bash
# Randomised differential runs against the oracle on the GPU box (tests/sweeps/fuzz_parity.py), every kernel class a leg of its
# own, four legs at a time:   tools/fuzz_all.sh [scale: cases = scale x the per-leg base, default 1] [seed base, default 600] [out-dir]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
S=${1:-1}; B=${2:-600}; O=${3:-gpurun_out/fuzz}; mkdir -p $O
leg() {  # name cases seed env...
  name=$1; cases=$(( $2 * S )); seed=$3; shift 3
  env "$@" timeout 3300 python tests/sweeps/fuzz_parity.py $cases $seed > $O/$name.log 2>&1
  echo "$name: $(tail -1 $O/$name.log)"
}
leg fuzz_general        6000 $((B+1)) X=1 &
leg fuzz_sigma_60_128   5000 $((B+2)) NL_FUZZ_MODES=2 NL_FUZZ_N=60,128 NL_FUZZ_WEIGHTED=0.0 &
leg fuzz_winsor_1_128   5000 $((B+3)) NL_FUZZ_MODES=3 NL_FUZZ_N=1,128 NL_FUZZ_WEIGHTED=0.05 &
leg fuzz_linfit_97_128  3000 $((B+6)) NL_FUZZ_MODES=5 NL_FUZZ_N=97,128 &
wait
leg fuzz_deep           1000 $((B+4)) NL_FUZZ_MODES=2,3 NL_FUZZ_N=129,512 NL_FUZZ_WEIGHTED=0.3 &
leg fuzz_other_modes    3000 $((B+5)) NL_FUZZ_MODES=0,1,4,5 &
leg fuzz_linfit_1_96    3000 $((B+7)) NL_FUZZ_MODES=5 NL_FUZZ_N=1,96 &
leg fuzz_linfit_deep     400 $((B+8)) NL_FUZZ_MODES=5 NL_FUZZ_N=129,512 &
wait
