python -m pytest tests -m gpu -x -q 2>&1 | tail -1
NL_FUZZ_MODES=2,3 NL_FUZZ_N=9,128 python tests/sweeps/fuzz_parity.py 15000 718 2>&1 | tail -1
NL_FUZZ_WEIGHTED=1 NL_FUZZ_MODES=2,3 NL_FUZZ_N=33,128 python tests/sweeps/fuzz_parity.py 5000 719 2>&1 | tail -1
python tests/sweeps/parity_sweep.py 2>&1 | tail -1
for n in 12 16 17 25 49 65 113 128; do python tools/ab_flags.py 2 $n 4096 0 4096 1 0 | cut -c1-150; done
for n in 15 16 17 49 52; do python tools/ab_flags.py 3 $n 4096 0 4096 1 0 | cut -c1-150; done
python bench.py --no-cpu --steps 20 --warmup 5 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*' | tr '\n' ' '
