#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for extra in "" "--force-dist"; do
 for rep in 1 2; do
  python bench.py --steps 50 --warmup 10 --no-cpu --no-also --height 512 --row0 1536 --image-height 4096 $extra 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('tile 512 $extra: ms_per_step %.4f  pass %.4f  kernel %.4f' % (d['ms_per_step'], d['roofline']['pass_ms'], d['roofline']['kernel_ms']))
"
 done
done
NL_DEV_FLAGS=32 python bench.py --steps 50 --warmup 10 --no-cpu --no-also --height 512 --row0 1536 --image-height 4096 --force-dist 2>&1 | tail -2 | cut -c1-300
