bash tools/timeline.sh 2>&1 | tail -9 | head -6
bash tools/timeline.sh --frames 512 2>&1 | tail -9 | head -6
