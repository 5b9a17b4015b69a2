#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for rows in 4096 512; do
  r0=$(( (4096 - rows) / 2 / 512 * 512 ))
  python bench.py --steps 50 --warmup 10 --no-cpu --no-also --height $rows --row0 $r0 --image-height 4096 --force-dist 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('rows $rows: ms_per_step %.4f  pass %.4f  kernel %.4f' % (d['ms_per_step'], d['roofline']['pass_ms'], d['roofline']['kernel_ms']))
"
done; done
timeout 900 python -m pytest tests/test_gpu_dist.py -m gpu -x -q 2>&1 | tail -2
