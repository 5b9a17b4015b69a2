python -m pytest tests -m gpu -x -q 2>&1 | tail -2
NL_FUZZ_N=30,130 NL_FUZZ_MODES=2,3 python tests/sweeps/fuzz_parity.py 30000 41 2>&1 | tail -2
python tests/sweeps/fuzz_parity.py 20000 42 2>&1 | tail -2
timeout 300 tools/gpu_profile.sh wsigma128 --weighted > /dev/null 2>&1
timeout 400 tools/gpu_profile.sh wwinsor128 --weighted --mode 3 > /dev/null 2>&1
