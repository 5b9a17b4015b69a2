#!/bin/bash
# round 5: kernel timelines of the passes whose tails are on the list (tile of the headline stack, winsor 16 / 24, sigma 512)
cd $GRAFT_REPO_ROOT
bash tools/timeline2.sh tile128 2 128 512 1536 4096 0 > /dev/null
bash tools/timeline2.sh winsor16 3 16 4096 0 4096 0 > /dev/null
bash tools/timeline2.sh winsor24 3 24 4096 0 4096 0 > /dev/null
bash tools/timeline2.sh sigma512 2 512 4096 0 4096 0 > /dev/null
bash tools/timeline2.sh c3tile 3 512 512 1536 4096 0 > /dev/null
for t in tile128 winsor16 winsor24 sigma512 c3tile; do echo "== $t"; tail -14 gpurun_out/timeline_$t.txt; done
