python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python tools/ab_flags.py 2 128 512 1536 4096 5 0
python tools/ab_flags.py 2 128 4096 0 4096 3 0
python tools/ab_flags.py 2 32 4096 0 4096 3 0
python tools/ab_flags.py 2 512 4096 0 4096 2 0
python tools/ab_flags.py 3 512 512 1536 4096 2 0
bash tools/timeline.sh --frames 128 --height 512 --image-height 4096 --row0 1536 > /dev/null 2>&1; tail -9 gpurun_out/timeline.txt | head -6 | cut -c1-120
