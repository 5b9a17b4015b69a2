#!/bin/bash
# round 5: both libraries through the whole GPU suite (default; experiments build via NLSTACK_LIB)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/hyg; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests_default.log 2>&1; echo "rc=$?" >> $O/tests_default.log
tail -3 $O/tests_default.log
NLSTACK_LIB=$PWD/nightlight_amd/libnlstack_exp.so timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests_exp.log 2>&1; echo "rc=$?" >> $O/tests_exp.log
tail -3 $O/tests_exp.log
NLSTACK_LIB=$PWD/nightlight_amd/libnlstack_exp.so NL_LFG=1 timeout 600 python -m pytest tests -m gpu -x -q -k "linear or linfit or c4" 2>&1 | tail -2
NLSTACK_LIB=$PWD/nightlight_amd/libnlstack_exp.so NL_CHUNKS=40,30,20,10 timeout 900 python -m pytest tests -m gpu -x -q -k "sigma or winsor or sweep" 2>&1 | tail -2
