#!/bin/bash
# round 5, final library: randomised differential runs with a random frame stride per case and a second pass per clip-mode case
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/fuzz5b; mkdir -p $O
timeout 2400 python tests/sweeps/fuzz_parity.py 20000 81 > $O/fuzz_general.log 2>&1; tail -2 $O/fuzz_general.log
NL_FUZZ_MODES=2 NL_FUZZ_N=60,128 NL_FUZZ_WEIGHTED=0.0 timeout 1800 python tests/sweeps/fuzz_parity.py 15000 82 > $O/fuzz_sigma_60_128.log 2>&1; tail -2 $O/fuzz_sigma_60_128.log
NL_FUZZ_MODES=3 NL_FUZZ_N=1,128 NL_FUZZ_WEIGHTED=0.05 timeout 2400 python tests/sweeps/fuzz_parity.py 20000 83 > $O/fuzz_winsor_1_128.log 2>&1; tail -2 $O/fuzz_winsor_1_128.log
NL_FUZZ_MODES=2,3 NL_FUZZ_N=129,512 NL_FUZZ_WEIGHTED=0.3 timeout 2400 python tests/sweeps/fuzz_parity.py 3000 84 > $O/fuzz_deep.log 2>&1; tail -2 $O/fuzz_deep.log
grep -c "protocol 3" $O/fuzz_sigma_60_128.log
