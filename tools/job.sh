python -m pytest tests -m gpu -x -q 2>&1 | tail -1
python tools/wall_probe.py 2 128 512 32
python tools/wall_probe.py 2 32 4096 32
python tools/wall_probe.py 2 128 4096 32
python bench.py --no-cpu --steps 20 --warmup 5 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*\|"pass_ms": [0-9.]*' | tr '\n' ' '
