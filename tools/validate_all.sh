#!/bin/bash
# Validation of a final library beyond the pytest suite (ON THE GPU BOX, from the repo root): repeatability soak, every-pixel parity sweep,
# randomised differential runs.   tools/validate_all.sh [fuzz scale, default 2] [out-dir]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
S=${1:-2}; O=${2:-gpurun_out/validate}; mkdir -p $O
( timeout 1500 python tools/soak.py > $O/soak.txt 2>&1; echo "soak rc=$?"; tail -2 $O/soak.txt ) &
( timeout 3000 python tests/sweeps/parity_sweep.py > $O/parity_sweep.txt 2>&1; echo "parity sweep rc=$?"; tail -2 $O/parity_sweep.txt ) &
wait
tools/fuzz_all.sh $S 700 $O/fuzz
