python -m pytest tests -m gpu -x -q 2>&1 | tail -1
NL_FUZZ_MODES=3 NL_FUZZ_N=9,128 python tests/sweeps/fuzz_parity.py 12000 919 2>&1 | tail -1
for n in 17 18 49 50 52 60 113 114 120 128; do python tools/ab_flags.py 3 $n 4096 0 4096 1 0 | cut -c1-150; done
