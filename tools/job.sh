python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for cap in 0 8 11 14; do echo "cap $cap"; NL_WINSOR_CAP=$cap bash tools/qb.sh "--mode 3 --no-cpu" "--mode 3 --frames 24 --no-cpu" "--mode 3 --frames 64 --no-cpu" "--mode 3 --frames 512 --height 512 --no-cpu" "--mode 3 --frames 300 --height 1024 --no-cpu" | cut -c1-118; done
