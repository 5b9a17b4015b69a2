python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -3 gpurun_out/bench_default.err; cat gpurun_out/bench_default.json
python tools/ab_flags.py 2 128 512 1536 4096 5 0,1
python tools/ab_flags.py 2 128 4096 0 4096 4 0,1
python tools/ab_flags.py 3 128 4096 0 4096 3 0,1
python tools/ab_flags.py 2 512 4096 0 4096 3 0,1
