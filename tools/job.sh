python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "four_pixels or wave_per_pixel" 2>&1 | tail -15
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -6 gpurun_out/pytest_gpu.log
for c4 in 0 1; do
export NL_COOP4=$c4; echo "== NL_COOP4=$c4"
python tools/ab_flags.py 3 512 512 1536 4096 2 0
python tools/ab_flags.py 2 512 4096 0 4096 2 0
python tools/ab_flags.py 2 128 512 1536 4096 3 0
python tools/ab_flags.py 3 128 4096 0 4096 2 0
python tools/ab_flags.py 2 128 4096 0 4096 2 0
done
