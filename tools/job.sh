python -m pytest tests -m gpu -x -q 2>&1 | tail -2
NL_FUZZ_MODES=2,3 NL_FUZZ_N=30,140 python tests/sweeps/fuzz_parity.py 15000 2024 2>&1 | tail -1
NL_FUZZ_MODES=2,3 NL_FUZZ_N=120,520 python tests/sweeps/fuzz_parity.py 3000 2025 2>&1 | tail -1
