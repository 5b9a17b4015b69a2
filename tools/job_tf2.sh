#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/tf2
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
NLSTACK_LIB=$PWD/nightlight_amd/libnlstack_exp.so timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
NL_STRIDE_PAD=1092 timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_parity.py::test_default_frame_stride_and_lent_frames -k "not padded_frame_stride" 2>&1 | tail -3
timeout 1200 python tools/soak.py 2>&1 | tail -3
NL_FUZZ_MODES=2 NL_FUZZ_N=60,128 NL_FUZZ_WEIGHTED=0.0 timeout 900 python tests/sweeps/fuzz_parity.py 3000 71 2>&1 | tail -2
for rep in 1 2; do for w in "2 128 512" "2 100 512" "2 80 512" "2 128 1024"; do python tools/wall_probe.py $w 8192 2>&1 | grep -v amdgpu; done; done | tee gpurun_out/tf2/wall.txt
timeout 300 tools/gpu_profile.sh sigma128tile --height 512 --row0 1536 --image-height 4096 > /dev/null 2>&1; head -8 gpurun_out/sigma128tile_kernel_stats.txt | cut -c1-150
