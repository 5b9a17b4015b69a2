python -m pytest tests -m gpu -x -q 2>&1 | tail -1
for rep in 1 2 3; do
for lib in libnlstack.so libnlstack_c.so; do
echo -n "$lib: "; NLSTACK_LIB=$PWD/nightlight_amd/$lib python bench.py --weighted --steps 5 --warmup 2 --no-cpu --no-also 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
done; done
