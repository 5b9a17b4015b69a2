python -m pytest tests -m gpu -x -q 2>&1 | tail -2
export NL_COOP_BLOCK=1
python bench.py --weighted --steps 5 --warmup 2 --no-cpu --no-also 2>/dev/null | cut -c100-200
python bench.py --weighted --frames 256 --steps 3 --warmup 1 --no-cpu --no-also 2>/dev/null | cut -c100-200
python tools/ab_flags.py 3 512 512 1536 4096 3 0 | cut -c1-110
python tools/ab_flags.py 2 512 4096 0 4096 2 0 | cut -c1-110
python tools/ab_flags.py 2 128 512 1536 4096 3 0 | cut -c1-110
