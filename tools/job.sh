python -m pytest tests -m gpu -x -q 2>&1 | tail -1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python tests/sweeps/fuzz_parity.py 30000 9001 2>&1 | tail -1
python tests/sweeps/parity_sweep.py 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 200 gpurun_out/bench_default.json
