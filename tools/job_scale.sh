#!/bin/bash
# round 5: one GPU's share of the headline stack for N = 1, 2, 4, 8 GPUs (strong scaling: rows 4096 / N), the pass as the
# scaling run drives it (world size 1, RCCL hop on the pass's stream: --force-dist), three runs each
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rows in 4096 2048 1024 512; do
  r0=$(( (4096 - rows) / 2 / 512 * 512 ))
  for rep in 1 2 3; do
    python bench.py --steps 50 --warmup 10 --no-cpu --no-also --height $rows --row0 $r0 --image-height 4096 --force-dist 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('rows $rows row0 $r0: ms_per_step %.4f  pass %.4f  kernel %.4f  sync %.4f' % (d['ms_per_step'], d['roofline']['pass_ms'], d['roofline']['kernel_ms'], d['ms_per_step_synchronous']))
"
  done
done
for n in 32 512; do
 for rows in 4096 512; do
  r0=$(( (4096 - rows) / 2 / 512 * 512 ))
  python bench.py --steps 30 --warmup 5 --no-cpu --no-also --frames $n --height $rows --row0 $r0 --image-height 4096 --force-dist 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('frames $n rows $rows: ms_per_step %.4f  pass %.4f  kernel %.4f' % (d['ms_per_step'], d['roofline']['pass_ms'], d['roofline']['kernel_ms']))
"
 done
done
