python -m pytest tests -m gpu -x -q 2>&1 | tail -2
P=tools/gpu_profile.sh
timeout 300 $P sigma128 > /dev/null 2>&1
timeout 300 $P sigma128tile --height 512 --row0 1536 --image-height 4096 > /dev/null 2>&1
timeout 300 $P sigma32 --frames 32 > /dev/null 2>&1
timeout 300 $P winsor128 --mode 3 > /dev/null 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
