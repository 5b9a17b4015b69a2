cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for t in "w16 --mode 3 --frames 16" "w24 --mode 3 --frames 24" "c3tile --mode 3 --frames 512 --height 512 --row0 1536 --image-height 4096" "c4 --mode 5" "sigma512 --frames 512" "headline "; do
  set -- $t; tag=$1; shift
  tools/timeline.sh "$@" > /dev/null 2>&1
  cp gpurun_out/timeline.txt gpurun_out/timeline_r6_$tag.txt
done
ls gpurun_out/timeline_r6_*
