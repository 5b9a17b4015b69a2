python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python tools/replay_probe2.py 2 128 1024 1; python tools/replay_probe2.py 3 128 1024 1; python tools/replay_probe2.py 2 96 512 1
bash tools/qb.sh "--weighted --no-cpu" "--weighted --mode 3 --no-cpu" "--weighted --frames 96 --no-cpu" "--weighted --frames 64 --no-cpu"
