P=tools/gpu_profile.sh
timeout 300 $P sigma300 --frames 300 > /dev/null 2>&1
timeout 300 $P sigma200 --frames 200 > /dev/null 2>&1
timeout 300 $P sigma256 --frames 256 > /dev/null 2>&1
timeout 300 $P winsor300tile --mode 3 --frames 300 --height 1024 > /dev/null 2>&1
timeout 300 $P sigma384 --frames 384 > /dev/null 2>&1
python tests/sweeps/parity_sweep.py > gpurun_out/parity_sweep.txt 2>&1; tail -2 gpurun_out/parity_sweep.txt
python tests/sweeps/fuzz_parity.py 60000 31 > gpurun_out/fuzz_a.txt 2>&1; tail -2 gpurun_out/fuzz_a.txt
NL_FUZZ_N=129,512 NL_FUZZ_MODES=2,3 python tests/sweeps/fuzz_parity.py 20000 32 > gpurun_out/fuzz_b.txt 2>&1; tail -2 gpurun_out/fuzz_b.txt
