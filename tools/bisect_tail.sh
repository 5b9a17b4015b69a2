#!/bin/bash
# Round 6, item 1 of VERDICT r05: which of the buffer-cache / stream-pool commits grew the pass tail of the C3 tile and of
# winsor 24?  Default bench line (--no-cpu) under each variant, interleaved, twice, on ONE box:
#   pre     library of d9feae8 (before 2da1693 / d720fce / 5b852c3 / ec3beb3), build/pre/libnlstack_pre.so
#   head    this tree
#   nopool  this tree, NL_STREAM_POOL=0            (5b852c3 off)
#   blk16   this tree, NL_CACHE_BLOCKS=16          (d720fce off; the per-device accounting of 2da1693 is the same on one GPU)
#   nocache this tree, NL_MEM_CACHE_MB=0
# Run on the GPU box from the repo root: tools/bisect_tail.sh [out-dir]
# (build/pre/libnlstack_pre.so is built in the BUILD container first -- only nlstack_api.hip and nlstack_group.hip differ:
#    for f in nlstack_api nlstack_group; do git show d9feae8:nightlight_amd/csrc/$f.hip > nightlight_amd/csrc/pre_$f.hip; done
#    hipcc <Makefile FLAGS> -c pre_<f>.hip -o build/pre/<f>.o; link with the default build's other objects; the leg is skipped when the
#    library is not there)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/bisect}; mkdir -p $O
run() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu > $O/$tag.json 2> $O/$tag.err
  python3 - $O/$tag.json $tag <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        b = json.loads(l)
        row = {a["tag"].split(" ")[0] + (" " + a["tag"].split(" ")[1] if a["tag"].startswith("C3") else ""): a for a in b["also"]}
        def f(t):
            a = row[t]; return "%s %.3f/%.3f/%.3f" % (t, a["ms_per_step"], a["pass_ms"], a["kernel_ms"])
        gs = row["C3 tile"].get("goal_seek", {})
        print("%-10s head %.4f | %s | %s | %s | %s | goal-seek %.1f ms / %d" % (sys.argv[2], b["ms_per_step"], f("C3 tile"), f("winsor24"), f("winsor16"), f("sigma512"),
              gs.get("total_ms", 0), gs.get("passes", 0)))
PY
}
for rep in 1 2; do
  [ -f $GRAFT_REPO_ROOT/build/pre/libnlstack_pre.so ] && run pre_$rep NLSTACK_LIB=$GRAFT_REPO_ROOT/build/pre/libnlstack_pre.so
  run head_$rep X=1
  run nopool_$rep NL_STREAM_POOL=0
  run blk16_$rep NL_CACHE_BLOCKS=16
  run nocache_$rep NL_MEM_CACHE_MB=0
done
