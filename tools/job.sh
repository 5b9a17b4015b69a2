python -m pytest tests -m gpu -x -q 2>&1 | tail -2
NL_FUZZ_MODES=2,3 NL_FUZZ_N=120,520 python tests/sweeps/fuzz_parity.py 4000 3031 2>&1 | tail -2
for f in 160 256 512; do for m in 2 3; do for fl in 4 0; do
echo -n "frames $f mode $m flags $fl: "; NL_DEV_FLAGS=$fl python bench.py --weighted --mode $m --frames $f --height 1024 --steps 2 --warmup 1 --preheat-steps 2 --no-cpu --no-also 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
done; done; done
python tools/ab_flags.py 2 512 4096 0 4096 2 0 | cut -c1-110
