#!/bin/bash
# round 5, final library, the long run: tools/job_fuzz5b.sh with other seeds and three times the cases
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/fuzz5c; mkdir -p $O
timeout 3000 python tests/sweeps/fuzz_parity.py 60000 91 > $O/fuzz_general.log 2>&1; tail -1 $O/fuzz_general.log
NL_FUZZ_MODES=2 NL_FUZZ_N=60,128 NL_FUZZ_WEIGHTED=0.0 timeout 1800 python tests/sweeps/fuzz_parity.py 40000 92 > $O/fuzz_sigma_60_128.log 2>&1; tail -1 $O/fuzz_sigma_60_128.log
NL_FUZZ_MODES=3 NL_FUZZ_N=1,128 NL_FUZZ_WEIGHTED=0.05 timeout 2400 python tests/sweeps/fuzz_parity.py 40000 93 > $O/fuzz_winsor_1_128.log 2>&1; tail -1 $O/fuzz_winsor_1_128.log
NL_FUZZ_MODES=2,3 NL_FUZZ_N=129,512 NL_FUZZ_WEIGHTED=0.3 timeout 2400 python tests/sweeps/fuzz_parity.py 8000 94 > $O/fuzz_deep.log 2>&1; tail -1 $O/fuzz_deep.log
NL_FUZZ_MODES=0,1,4,5 timeout 2400 python tests/sweeps/fuzz_parity.py 20000 95 > $O/fuzz_other_modes.log 2>&1; tail -1 $O/fuzz_other_modes.log
true
