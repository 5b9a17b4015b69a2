#!/bin/bash
# round 5: randomised differential runs on the final library (certificate of the winsorization loops in every one-lane kernel)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/fuzz5; mkdir -p $O
NL_FUZZ_MODES=3 NL_FUZZ_N=1,128 NL_FUZZ_WEIGHTED=0.1 timeout 1500 python tests/sweeps/fuzz_parity.py 4000 51 > $O/fuzz_winsor_1_128.log 2>&1; tail -2 $O/fuzz_winsor_1_128.log
NL_FUZZ_MODES=3 NL_FUZZ_N=12,40 NL_FUZZ_WEIGHTED=0.0 timeout 900 python tests/sweeps/fuzz_parity.py 3000 52 > $O/fuzz_winsor_12_40.log 2>&1; tail -2 $O/fuzz_winsor_12_40.log
timeout 1500 python tests/sweeps/fuzz_parity.py 3000 53 > $O/fuzz_general.log 2>&1; tail -2 $O/fuzz_general.log
NL_FUZZ_MODES=2,3 NL_FUZZ_N=257,512 NL_FUZZ_WEIGHTED=0.6 timeout 1500 python tests/sweeps/fuzz_parity.py 800 54 > $O/fuzz_weighted_deep.log 2>&1; tail -2 $O/fuzz_weighted_deep.log
