NL_FUZZ_N=497,512 NL_FUZZ_MODES=2 python tests/sweeps/fuzz_parity.py 2500 11 2>&1 | tail -4
python tests/sweeps/fuzz_parity.py 1500 12 2>&1 | tail -3
