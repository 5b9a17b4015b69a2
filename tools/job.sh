python -m pytest tests -m gpu -x -q 2>&1 | tail -1
for m in 2 3; do python bench.py --weighted --mode $m --steps 5 --warmup 2 --no-cpu --no-also 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; done
for f in 64 96; do python bench.py --weighted --frames $f --steps 3 --warmup 1 --no-cpu --no-also 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; done
python bench.py --weighted --frames 600 --height 512 --steps 2 --warmup 1 --no-cpu --no-also 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
python tools/ab_flags.py 3 512 512 1536 4096 3 0 | cut -c1-110
python tools/ab_flags.py 2 512 4096 0 4096 2 0 | cut -c1-110
