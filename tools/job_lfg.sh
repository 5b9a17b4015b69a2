#!/bin/bash
# round 5: guarded linear fit -- parity tests, then A/B against the bit-exact cascade (developer switch 4096)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/lfg; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "linear or linfit or mode_matches or ties_and or infinite or newton or c4" > $O/tests_lf.log 2>&1; echo "rc=$?" >> $O/tests_lf.log
tail -8 $O/tests_lf.log
for n in 128 100 64 32 25; do
  timeout 600 python tools/ab_flags.py 5 $n 4096 0 4096 3 0,4096 2>&1 | tee -a $O/ab.log
done
python - <<'PY' 2>&1 | tee -a gpurun_out/lfg/ab.log
import sys; sys.path.insert(0, ".")
from nightlight_amd import StackHandle
for n in (128, 32):
    with StackHandle(n, 4096, 4096) as st:
        st.fill_synthetic(seed=1)
        st.run(5, 3.0, 3.0, fetch=False)
        print(n, "stage counts", st.linfit_stage_counts, st.last_kernel_name)
PY
