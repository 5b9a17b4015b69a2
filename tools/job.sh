python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for n in 512 300 200 160 384 448 256; do python tools/ab_flags.py 2 $n 4096 0 4096 2 0 | cut -c1-110; done
python tools/ab_flags.py 3 512 512 1536 4096 2 0 | cut -c1-110
python tools/ab_flags.py 3 300 1024 0 4096 2 0 | cut -c1-110
NL_FUZZ_N=129,512 NL_FUZZ_MODES=2,3 python tests/sweeps/fuzz_parity.py 6000 71 2>&1 | tail -1
