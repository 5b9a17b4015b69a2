python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python tests/sweeps/fuzz_parity.py 6000 95 2>&1 | tail -1
for m in 2 3; do python bench.py --weighted --mode $m --steps 5 --warmup 2 --no-cpu --no-also 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; done
