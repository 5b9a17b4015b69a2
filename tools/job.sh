python -m pytest tests -m gpu -x -q 2>&1 | tail -1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        b=json.loads(l); print(b['value'], b['ms_per_step'], b['roofline']['frac'], b['roofline']['pass_frac'], b['cpu_baseline']['parity_with_gpu']['clip_counters_equal'], [ (a['ms_per_step'], a['pass_frac']) for a in b['also']])
"
