#!/bin/bash
# Refresh every workload summarised under profiles/ (run on the GPU box from the repo root), then
# tools/collect_profiles.py <tags> copies the summaries.
set -u
P=tools/gpu_profile.sh
timeout 300 $P sigma128
timeout 300 $P sigma128tile --height 512 --row0 1536 --image-height 4096
timeout 300 $P sigma32 --frames 32
timeout 300 $P sigma300 --frames 300
timeout 300 $P sigma200 --frames 200
timeout 300 $P sigma256 --frames 256
timeout 300 $P sigma512 --frames 512
timeout 300 $P winsor128 --mode 3
timeout 300 $P winsor512tile --mode 3 --frames 512 --height 512
timeout 300 $P winsor512mid --mode 3 --frames 512 --height 512 --row0 1536 --image-height 4096
timeout 300 $P winsor300tile --mode 3 --frames 300 --height 1024
timeout 300 $P linfit128 --mode 5
timeout 400 $P linfit256 --mode 5 --frames 256
timeout 300 $P median64 --mode 0 --frames 64 --width 6000 --height 4000
timeout 300 $P median128 --mode 0
timeout 300 $P median512 --mode 0 --frames 512
timeout 300 $P mad128 --mode 4
timeout 300 $P mean128 --mode 1
timeout 300 $P wsigma128 --weighted
timeout 300 $P wwinsor128 --weighted --mode 3
timeout 300 $P sigma384 --frames 384
timeout 300 $P sigma160 --frames 160
timeout 300 $P sigma100 --frames 100
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -c 400 gpurun_out/bench_default.json
