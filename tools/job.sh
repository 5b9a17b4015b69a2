python -m pytest tests -m gpu -x -q 2>&1 | tail -1
NL_FUZZ_WEIGHTED=1 NL_FUZZ_MODES=3 NL_FUZZ_N=33,128 python tests/sweeps/fuzz_parity.py 6000 1001 2>&1 | tail -1
python tests/sweeps/fuzz_parity.py 15000 1002 2>&1 | tail -1
python tests/sweeps/parity_sweep.py 2>&1 | tail -1
python bench.py --no-cpu --steps 20 --warmup 5 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*' | tr '\n' ' '
python bench.py --weighted --mode 3 --steps 3 --warmup 1 --no-also 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"bit_exact": [a-z]*\|"clip_counters_equal": [a-z]*' | tr '\n' ' '
