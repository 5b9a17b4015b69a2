#!/bin/bash
# quick bench lines (no profiler): tools/qb.sh "<bench args>" "<bench args>" ...   -> one summary line each
for a in "$@"; do
  python bench.py --steps 10 --warmup 3 --no-also $a 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if not l.startswith('{'): continue
    d=json.loads(l); r=d['roofline']; c=d.get('cpu_baseline') or {}
    print('$a'.ljust(44), 'pass %.3f ms  kernel %.3f ms  frac %.3f/%.3f  redone %s  %s  parity %s' % (d['ms_per_step'], r['kernel_ms'], r['pass_frac'], r['frac'], r.get('pixels_redone_by_exact_kernel'), r['kernel'], (c.get('parity_with_gpu') or {})))
"
done
