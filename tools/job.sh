for n in 300 200 160 384 448 272; do python tools/ab_flags.py 2 $n 4096 0 4096 2 0; done
python tools/ab_flags.py 3 300 1024 0 4096 2 0
NL_FUZZ_N=129,496 NL_FUZZ_MODES=2,3 python tests/sweeps/fuzz_parity.py 2500 21 2>&1 | tail -2
