#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "padded_frame_stride or lent_frames" 2>&1 | tail -5
NLSTACK_LIB=$PWD/nightlight_amd/libnlstack_exp.so timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "padded_frame_stride or lent_frames or four_pixels or split_lds" 2>&1 | tail -5
