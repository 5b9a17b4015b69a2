python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "wave_per_pixel" 2>&1 | tail -5
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python tools/ab_flags.py 2 128 512 1536 4096 5 0
python tools/ab_flags.py 2 128 4096 0 4096 3 0
python tools/ab_flags.py 2 512 4096 0 4096 2 0
python tools/ab_flags.py 3 512 512 1536 4096 2 0
bash tools/qb.sh "--weighted --no-cpu" "--weighted --mode 3 --no-cpu --height 1024"
