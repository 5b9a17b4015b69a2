#!/usr/bin/env python3
"""Static opcode histogram of one kernel in a hipcc -S listing, per basic block
(developer utility: where do the VALU slots of a register-resident kernel go?).
usage: isa_hist.py listing.s mangled-name-substring"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if re.match(r"_Z\w+:", l) and key in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
HALF = ("v_min_f32", "v_max_f32", "v_med3", "v_min3", "v_max3", "v_cmp", "v_min_i32", "v_max_i32",
        "v_min_u32", "v_max_u32", "v_bfe", "v_perm", "v_alignbit")
blocks, name, cur = [], "entry", collections.Counter()
for l in lines[start + 1:end]:
    l = l.strip()
    if re.match(r"\.LBB\d+_\d+:", l):
        blocks.append((name, cur))
        name, cur = l, collections.Counter()
        continue
    m = re.match(r"([vs]_\w+|ds_\w+|buffer_\w+|global_\w+|flat_\w+|scratch_\w+)", l)
    if m:
        cur[m.group(1)] += 1
blocks.append((name, cur))
total = collections.Counter()
for name, c in blocks:
    total.update(c)
    valu = sum(n for op, n in c.items() if op.startswith("v_"))
    slots = sum(n * (2 if op.startswith(HALF) else 1) for op, n in c.items() if op.startswith("v_"))
    if valu > 20:
        print("%-12s valu %5d  slots %5d  %s" % (name, valu, slots, c.most_common(7)))
valu = sum(n for op, n in total.items() if op.startswith("v_"))
slots = sum(n * (2 if op.startswith(HALF) else 1) for op, n in total.items() if op.startswith("v_"))
print("TOTAL valu %d slots %d" % (valu, slots))
print(total.most_common(25))
