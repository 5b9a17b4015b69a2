#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/gpu_profile.sh r05_wsigma512_pf8 --weighted --frames 512 --height 1024 > /dev/null 2>&1
bash tools/gpu_profile.sh r05_wwinsor512_pf8 --weighted --mode 3 --frames 512 --height 1024 > /dev/null 2>&1
NL_COOP_PF=0 bash tools/gpu_profile.sh r05_wwinsor512_pf2 --weighted --mode 3 --frames 512 --height 1024 > /dev/null 2>&1
for t in r05_wsigma512_pf8 r05_wwinsor512_pf8 r05_wwinsor512_pf2; do echo "== $t"; head -5 gpurun_out/${t}_kernel_stats.txt; cat gpurun_out/${t}_traffic.json; python3 -c "
import json
for l in open('gpurun_out/${t}_bench.json'):
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['roofline']['kernel'])
"; done
