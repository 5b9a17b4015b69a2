python -m pytest tests -m gpu -x -q 2>&1 | tail -3
bash tools/qb.sh "--weighted --no-cpu" "--weighted --mode 3 --no-cpu" "--weighted --frames 96 --no-cpu" "--weighted --frames 64 --no-cpu" "--weighted --frames 64 --mode 3 --no-cpu"
NL_WDECIDE=0 bash tools/qb.sh "--weighted --no-cpu" "--weighted --mode 3 --no-cpu"
python bench.py --weighted --steps 5 --warmup 2 --no-also 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['cpu_baseline']['parity_with_gpu'])"
python bench.py --weighted --mode 3 --steps 3 --warmup 1 --no-also 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['cpu_baseline']['parity_with_gpu'])"
