bash tools/qb.sh "--weighted --no-cpu" "--weighted --mode 3 --no-cpu --height 1024" "--frames 600 --height 256 --no-cpu" "--frames 600 --mode 0 --height 256 --no-cpu"
python -m pytest tests -m gpu -x -q -k "wave_per_pixel or weighted or more_than_512 or large_stacks" 2>&1 | tail -3
