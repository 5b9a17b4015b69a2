python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -6 gpurun_out/pytest_gpu.log
free -g | head -2
python tools/batches_probe.py 96 32 2>&1 | grep -v amdgpu.ids | tee gpurun_out/batches_probe.txt
