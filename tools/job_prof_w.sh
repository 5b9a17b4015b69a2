#!/bin/bash
cd $GRAFT_REPO_ROOT
P=tools/gpu_profile.sh
timeout 300 $P wsigma128 --weighted > /dev/null 2>&1
timeout 300 $P wwinsor128 --weighted --mode 3 > /dev/null 2>&1
timeout 600 $P wsigma512 --weighted --frames 512 --height 1024 --preheat-steps 8 > /dev/null 2>&1
for t in wsigma128 wwinsor128 wsigma512; do cat gpurun_out/${t}_traffic.json; done
