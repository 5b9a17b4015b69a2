#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "padded_frame_stride or lent_frames" 2>&1 | tail -30
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
NL_STRIDE_PAD=1092 timeout 2400 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_parity.py::test_default_frame_stride_and_lent_frames -k "not padded_frame_stride" 2>&1 | tail -5
python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_pad_default.json; python -c "
import json; d=json.load(open('gpurun_out/bench_pad_default.json')); print(d['value'], d['ms_per_step'], d['roofline']); print({k:(v.get('ms_per_step'), v.get('roofline',{}).get('frac')) for k,v in d.get('also',{}).items()})"
