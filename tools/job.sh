python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "partition_corner" 2>&1 | tail -5
