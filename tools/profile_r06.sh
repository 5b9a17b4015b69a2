#!/bin/bash
# Round-6 profile set (run on the GPU box from the repo root), then tools/collect_profiles.py r06 <tags>.
set -u
P=tools/gpu_profile.sh
timeout 300 $P sigma128
timeout 300 $P sigma128tile --height 512 --row0 1536 --image-height 4096
timeout 300 $P sigma32 --frames 32
timeout 300 $P sigma512 --frames 512
timeout 300 $P winsor16 --mode 3 --frames 16
timeout 300 $P winsor24 --mode 3 --frames 24
timeout 300 $P winsor32 --mode 3 --frames 32
timeout 300 $P winsor64 --mode 3 --frames 64
timeout 300 $P winsor128 --mode 3
timeout 300 $P winsor512mid --mode 3 --frames 512 --height 512 --row0 1536 --image-height 4096
timeout 300 $P linfit128 --mode 5
timeout 300 $P linfit32 --mode 5 --frames 32
timeout 300 $P median64 --mode 0 --frames 64 --width 6000 --height 4000
timeout 300 $P median128 --mode 0
timeout 300 $P mad128 --mode 4
timeout 300 $P mean128 --mode 1
timeout 300 $P wsigma128 --weighted
timeout 300 $P wwinsor128 --weighted --mode 3
timeout 600 $P wsigma512 --weighted --frames 512 --height 1024 --preheat-steps 8
timeout 600 $P wwinsor512 --weighted --mode 3 --frames 512 --height 1024 --preheat-steps 8
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -c 300 gpurun_out/bench_default.json
