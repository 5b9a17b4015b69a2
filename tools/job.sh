python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python tools/ab_flags.py 3 512 512 1536 4096 3 0,4
python tools/ab_flags.py 3 128 4096 0 4096 3 0,4
python tools/ab_flags.py 3 300 1024 0 4096 2 0,4
NL_FUZZ_N=20,512 NL_FUZZ_MODES=3 python tests/sweeps/fuzz_parity.py 6000 51 2>&1 | tail -2
