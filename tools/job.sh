python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python tools/ab_flags.py 2 512 4096 0 4096 3 0 | cut -c1-170
NL_FUZZ_N=497,512 NL_FUZZ_MODES=2 python tests/sweeps/fuzz_parity.py 3000 61 2>&1 | tail -1
