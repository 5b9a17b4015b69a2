python -m pytest tests -m gpu -x -q 2>&1 | tail -1
python tests/sweeps/fuzz_parity.py 10000 4321 2>&1 | tail -1
for m in 2 3; do python bench.py --weighted --mode $m --steps 5 --warmup 2 --no-cpu --no-also 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; done
python tools/ab_flags.py 3 512 512 1536 4096 3 0 | cut -c1-110
python tools/ab_flags.py 2 512 4096 0 4096 2 0 | cut -c1-110
python tools/ab_flags.py 2 128 512 1536 4096 3 0 | cut -c1-110
