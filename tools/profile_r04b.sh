#!/bin/bash
# Round-4 profile refresh of the workloads the last kernel changes touched (run on the GPU box), then collect_profiles.py
set -u
P=tools/gpu_profile.sh
timeout 300 $P sigma128
timeout 300 $P sigma128tile --height 512 --row0 1536 --image-height 4096
timeout 300 $P sigma32 --frames 32
timeout 300 $P winsor16 --mode 3 --frames 16
timeout 300 $P winsor24 --mode 3 --frames 24
timeout 300 $P winsor64 --mode 3 --frames 64
timeout 300 $P winsor120 --mode 3 --frames 120
timeout 300 $P winsor128 --mode 3
timeout 300 $P sigma25 --frames 25
