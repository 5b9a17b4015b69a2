NL_FUZZ_WEIGHTED=1 NL_FUZZ_MODES=2,3 NL_FUZZ_N=129,512 python tests/sweeps/fuzz_parity.py 5000 5151 2>&1 | tail -2
NL_FUZZ_WEIGHTED=1 NL_FUZZ_MODES=2,3 NL_FUZZ_N=20,140 python tests/sweeps/fuzz_parity.py 8000 5152 2>&1 | tail -1
python bench.py --weighted --frames 512 --steps 2 --warmup 1 --preheat-steps 1 --no-cpu --no-also 2>/dev/null | cut -c1-2000 | grep -o '"ms_per_step": [0-9.]*\|"bit_exact": [a-z]*\|"clip_counters_equal": [a-z]*'
