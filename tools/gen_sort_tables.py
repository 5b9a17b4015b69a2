#!/usr/bin/env python3
"""Generate nightlight_amd/csrc/sort_tables.inc: the sorting networks of the register-resident
kernels as lists of 2- and 3-input min / med / max operations.

Starting point is Batcher's odd-even merge sort (the network fast_common.hpp also builds at
compile time), optionally pruned to the ranks a kernel needs (ZonalNetwork).  gfx950 issues
v_min_f32 / v_max_f32 and v_min3 / v_med3 / v_max3 at the same (half) rate, so a network costs
its instruction count.  A compare-exchange CE(x, y) followed by CE(min(x,y), z) equals

    min3(x, y, z)   and   med3(x, y, z)        provided  z <= max(x, y)  always holds,

(and CE(max(x,y), z) = med3, max3 provided z >= min(x, y)), which saves the instruction for the
intermediate value.  Whether the side condition holds for every input is decided with the 0-1
principle: inside merge level p every block of 2p wires receives two sorted runs, i.e. one of
(p+1)^2 zero-one patterns; min/med/max commute with thresholds, so a relation that holds on all of
them holds for all real inputs.  A value is inlined only into a consumer of the same merge level,
the consumer's other input and both inputs of the producer stay materialised.  About a quarter of
the instructions disappear (2942 -> 2184 for the full 128-element sort).

Output values live in numbered slots (register names for the compiler); kOut[k] is the slot that
ends up holding rank k.

usage: gen_sort_tables.py [--check]     (--check: verify the committed file is current)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "nightlight_amd", "csrc", "sort_tables.inc")

MIN2, MAX2, MIN3, MED3, MAX3 = range(5)

K_ZONE, K_PAD_MAX, K_MEDIAN_PAD = 8, 8, 16
SIZES = (8, 16, 24, 32, 48, 64, 80, 96, 112, 128)


def oem(ns):
    """Comparators (level, lo, hi) of the odd-even merge sort for ns elements, in the order
    OemNetwork<NS> (fast_common.hpp) emits them."""
    p2 = 1
    while p2 < ns:
        p2 *= 2
    ces = []
    p = 1
    while p < p2:
        k = p
        while k >= 1:
            j = k % p
            while j + k < p2:
                for i in range(k):
                    if (i + j) // (2 * p) == (i + j + k) // (2 * p) and i + j + k < ns:
                        ces.append((p, i + j, i + j + k))
                j += 2 * k
            k //= 2
        p *= 2
    return ces


def prune(ns, ces, e):
    """ZonalNetwork: keep[] flags; positions [0,E0), [E1,E2), [E3,NS) need exact ranks."""
    e0, e1, e2, e3 = e

    def stretch(i):
        return 0 if i < e0 else (1 if i < e1 else (0 if i < e2 else (2 if i < e3 else 0)))

    touched = [False] * ns
    keep = [False] * len(ces)
    for c in range(len(ces) - 1, -1, -1):
        _, lo, hi = ces[c]
        sl, sh = stretch(lo), stretch(hi)
        if sl == sh and sl != 0 and not touched[lo] and not touched[hi]:
            continue
        keep[c] = True
        touched[lo] = touched[hi] = True
    return keep


def block_domain(p):
    """zero-one patterns of a block of 2p wires holding two ascending runs, as one bitset per wire"""
    w = [0] * (2 * p)
    idx = 0
    for i in range(p + 1):
        for j in range(p + 1):
            for k in range(p):
                if k >= p - i:
                    w[k] |= 1 << idx
                if k >= p - j:
                    w[p + k] |= 1 << idx
            idx += 1
    return w


def bitonic(ns, keep_ends):
    """Comparators of the half-cleaner cascade that sorts a bitonic sequence of ns elements
    (distances ns/2 .. 1), as half_clean / half_clean_ends (fast_ml_common.hpp) emit them;
    keep_ends > 0: only the blocks that can reach the keep_ends lowest / highest positions."""
    ces = []
    d = ns // 2
    while d >= 1:
        for t in range(ns // 2):
            i = ((t & ~(d - 1)) << 1) | (t & (d - 1))
            blk = i & ~(2 * d - 1)
            if not keep_ends or blk < keep_ends or blk + 2 * d > ns - keep_ends:
                ces.append((ns // 2, i, i | d))          # one "level": the block is the whole sequence
        d //= 2
    return ces


def bitonic_domain(ns):
    """zero-one bitonic sequences 0^a 1^b 0^c and 1^a 0^b 1^c, one bitset per wire"""
    w = [0] * ns
    idx = 0
    for a in range(ns + 1):
        for b in range(ns + 1 - a):
            for k in range(a, a + b):
                w[k] |= 1 << idx
            idx += 1
            for k in range(0, a):
                w[k] |= 1 << idx
            for k in range(a + b, ns):
                w[k] |= 1 << idx
            idx += 1
    return w


def build(ns, e):
    """-> (ops, out_slot, n_slots, n_ce); ops = (kind, dst, a, b, c)"""
    ces = oem(ns)
    keep = prune(ns, ces, e) if any(e) else [True] * len(ces)
    return build_net(ns, ces, keep, block_domain)


def build_bitonic(ns, keep_ends):
    ces = bitonic(ns, keep_ends)
    dom = bitonic_domain(ns)
    return build_net(ns, ces, [True] * len(ces), lambda p: dom)


def build_net(ns, ces, keep, domain_of):
    # ---- nodes with the 0-1 images of their inputs, producer links inside a merge level ----
    nodes = []
    cur_p, last, state, dom = None, {}, {}, None
    for idx, (p, lo, hi) in enumerate(ces):
        if p != cur_p:
            cur_p, last, state, dom = p, {}, {}, domain_of(p)
        blk = lo // (2 * p) * (2 * p)
        for w in (lo, hi):
            if w not in state:
                state[w] = dom[w - blk]
        x, y = state[lo], state[hi]
        state[lo], state[hi] = x & y, x | y
        if not keep[idx]:
            continue
        nodes.append(dict(lo=lo, hi=hi, x=x, y=y, src_lo=last.get(lo), src_hi=last.get(hi)))
        last[lo] = (len(nodes) - 1, "lo")
        last[hi] = (len(nodes) - 1, "hi")
    # ---- which intermediate values can be inlined into their consumer ----
    cands = []
    for d, n in enumerate(nodes):
        for side in ("lo", "hi"):
            src = n["src_" + side]
            if src is None:
                continue
            c, which = src
            z = n["y"] if side == "lo" else n["x"]
            px, py = nodes[c]["x"], nodes[c]["y"]
            ok = (z & ~(px | py)) == 0 if which == "lo" else ((px & py) & ~z) == 0
            if ok:
                cands.append((c, which, d, side))
    role = {}                     # node -> "fused" | "plain" (must stay plain: one of its outputs is inlined)
    inlined = {}                  # consumer node -> (producer node, which output, consumer side)
    gone = set()                  # (producer node, which) never materialised
    for c, which, d, side in sorted(cands, key=lambda t: -t[2]):
        if d in role or role.get(c) == "fused":
            continue
        role[d] = "fused"
        role[c] = "plain"
        inlined[d] = (c, which, side)
        gone.add((c, which))
    # ---- emit operations on SSA values ----
    wire_val = list(range(ns))    # value currently on each wire
    n_val = ns
    node_in = {}                  # node -> (value on lo wire, value on hi wire) at its inputs
    sops = []                     # (kind, dst value, a, b, c)
    for d, n in enumerate(nodes):
        a, b = wire_val[n["lo"]], wire_val[n["hi"]]
        node_in[d] = (a, b)
        if d in inlined:
            c, which, side = inlined[d]
            px, py = node_in[c]
            z = b if side == "lo" else a
            lo_v, hi_v = n_val, n_val + 1
            n_val += 2
            if which == "lo":
                sops.append((MIN3, lo_v, px, py, z))
                sops.append((MED3, hi_v, px, py, z))
            else:
                sops.append((MED3, lo_v, px, py, z))
                sops.append((MAX3, hi_v, px, py, z))
        else:
            lo_v = hi_v = None
            if (d, "lo") not in gone:
                lo_v = n_val
                n_val += 1
                sops.append((MIN2, lo_v, a, b, a))
            if (d, "hi") not in gone:
                hi_v = n_val
                n_val += 1
                sops.append((MAX2, hi_v, a, b, a))
        wire_val[n["lo"]], wire_val[n["hi"]] = lo_v, hi_v
    assert all(v is not None for v in wire_val)
    sops = schedule(sops, ns, wire_val)
    # ---- values -> slots (read sources, release the dead ones, then write) ----
    last_use = {}
    for i, (_, _, a, b, c) in enumerate(sops):
        for s in (a, b, c):
            last_use[s] = i
    for v in wire_val:
        last_use[v] = len(sops)
    slot = {v: v for v in range(ns)}
    free, n_slots = [], ns
    ops = []
    for i, (kind, dst, a, b, c) in enumerate(sops):
        sa, sb, sc = slot[a], slot[b], slot[c]
        for s in {a, b, c}:
            if last_use[s] == i:
                free.append(slot[s])
        if free:
            free.sort()
            slot[dst] = free.pop(0)
        else:
            slot[dst] = n_slots
            n_slots += 1
        ops.append((kind, slot[dst], sa, sb, sc))
    return ops, [slot[v] for v in wire_val], n_slots, len(nodes)


def schedule(sops, ns, final_vals):
    """Reorder the operations (dependencies kept) so that few values are alive at any time: an
    inlined value keeps both inputs of its producer alive until the consumer has run, so the
    consumer should follow closely.  Greedy list scheduling over units (a unit = the two 3-input
    operations of one fused comparator, or a single 2-input operation): among the ready units
    take the one that frees the most registers, oldest first."""
    units = []
    i = 0
    while i < len(sops):
        if sops[i][0] >= MIN3:
            units.append([sops[i], sops[i + 1]])
            i += 2
        else:
            units.append([sops[i]])
            i += 1
    uses = {}
    for u in units:
        for s in {x for op in u for x in op[2:]}:
            uses[s] = uses.get(s, 0) + 1
    for v in final_vals:
        uses[v] = uses.get(v, 0) + 1
    producer = {op[1]: k for k, u in enumerate(units) for op in u}
    srcs_of = [{x for op in u for x in op[2:]} for u in units]
    waiting, readers = [], {}
    for k, srcs in enumerate(srcs_of):
        dep = {s for s in srcs if s in producer}
        waiting.append(len(dep))
        for s in dep:
            readers.setdefault(s, []).append(k)
    ready = [k for k in range(len(units)) if waiting[k] == 0]
    order = []
    while ready:
        best, best_key = None, None
        for k in ready:
            delta = len(units[k]) - sum(1 for s in srcs_of[k] if uses[s] == 1)
            key = (delta, k)
            if best_key is None or key < best_key:
                best, best_key = k, key
        ready.remove(best)
        order.append(best)
        for s in srcs_of[best]:
            uses[s] -= 1
        for op in units[best]:
            for r in readers.get(op[1], []):
                waiting[r] -= 1
                if waiting[r] == 0:
                    ready.append(r)
    assert len(order) == len(units)
    return [op for k in order for op in units[k]]


def simulate(ops, out, n_slots, x):
    """x: [trials, ns] -> [trials, ns] in rank order"""
    w = np.full((x.shape[0], n_slots), np.nan, dtype=x.dtype)
    w[:, :x.shape[1]] = x
    for kind, dst, a, b, c in ops:
        if kind == MIN2:
            r = np.minimum(w[:, a], w[:, b])
        elif kind == MAX2:
            r = np.maximum(w[:, a], w[:, b])
        else:
            t = np.sort(np.stack([w[:, a], w[:, b], w[:, c]]), axis=0)
            r = t[kind - MIN3]
        w[:, dst] = r
    return w[:, out]


def needed(ns, e):
    if not any(e):
        return [(0, ns)], []
    e0, e1, e2, e3 = e
    return [(0, e0), (e1, e2), (e3, ns)], [(e0, e1), (e2, e3)]


def verify(ns, e, ops, out, n_slots, trials=600, seed=1):
    rng = np.random.default_rng(seed + ns)
    xs = [rng.standard_normal((trials, ns)),
          rng.integers(0, 4, (trials, ns)).astype(np.float64),          # many ties
          (rng.random((trials, ns)) < rng.random((trials, 1))).astype(np.float64)]   # zero-one
    x = np.concatenate(xs).astype(np.float32)
    x[::7, ns - 3:] = np.inf                                             # pads
    y = simulate(ops, out, n_slots, x)
    ref = np.sort(x, axis=1)
    exact, sets = needed(ns, e)
    for a, b in exact:
        assert np.array_equal(y[:, a:b], ref[:, a:b]), (ns, e, "ranks", a, b)
    for a, b in sets:
        assert np.array_equal(np.sort(y[:, a:b], axis=1), ref[:, a:b]), (ns, e, "set", a, b)


def verify_bitonic(ns, keep_ends, ops, out, n_slots, trials=400, seed=3):
    rng = np.random.default_rng(seed + ns + keep_ends)
    rows = []
    for kind in range(3):
        x = rng.standard_normal((trials, ns))
        if kind == 1:
            x = rng.integers(0, 5, (trials, ns)).astype(np.float64)            # ties
        if kind == 2:
            x[:, ns - 5:] = np.inf                                              # pads
        cut = rng.integers(0, ns + 1, trials)
        for r in range(trials):
            up = np.sort(x[r, :cut[r]])
            dn = np.sort(x[r, cut[r]:])[::-1]
            seq = np.concatenate([up, dn])                                      # ascending, then descending
            rows.append(np.roll(seq, rng.integers(0, ns)) if r % 3 == 0 else (seq if r % 3 == 1 else seq[::-1]))
    x = np.array(rows, dtype=np.float32)
    y = simulate(ops, out, n_slots, x)
    ref = np.sort(x, axis=1)
    k = keep_ends if keep_ends else ns // 2
    assert np.array_equal(y[:, :k], ref[:, :k]) and np.array_equal(y[:, ns - k:], ref[:, ns - k:]), (ns, keep_ends)
    assert np.array_equal(np.sort(y, axis=1), ref), (ns, keep_ends, "set")


BITONIC = ((128, 0), (128, 16), (128, 32), (64, 0), (32, 0), (16, 0))       # (size, KEEP): the merges of the multi-lane kernels; the cascades of the selection front end


def variants():
    v = []
    for ns in SIZES:
        v.append((ns, (0, 0, 0, 0)))                                     # FullSort
        if ns >= 24:                                                     # zonal sigma (stack_fast.hip)
            kz, kp = (K_ZONE, K_PAD_MAX) if ns >= 48 else (4, 4)
            for pads in (kp, 0):                                         # 0: the TIGHT variant (exactly ns frames)
                zl, zh = kz, ns - kz - pads
                v.append((ns, (zl, zh // 2 - 1, zl + ns // 2 + 1, zh)))
        if ns >= 32:                                                     # median window
            v.append((ns, (0, (ns - K_MEDIAN_PAD) // 2 - 1, ns // 2 + 1, ns)))
    return v


def render():
    lines = ["// sort_tables.inc -- GENERATED by tools/gen_sort_tables.py, do not edit.",
             "// Sorting networks as 2-/3-input min, med, max operations on numbered slots.",
             "// op = {kind (0 min, 1 max, 2 min3, 3 med3, 4 max3), dst, a, b, c}", ""]
    summary = []
    for ns, e in variants():
        ops, out, n_slots, n_ce = build(ns, e)
        verify(ns, e, ops, out, n_slots)
        summary.append((ns, e, n_ce, len(ops), n_slots))
        lines.append("template <> struct FusedNet<%d, %d, %d, %d, %d> {" % ((ns,) + e))
        lines.append("    static constexpr int kCount = %d, kSlots = %d, kComparators = %d;" % (len(ops), n_slots, n_ce))
        lines.append("    static constexpr FusedOp kOps[%d] = {" % len(ops))
        row = []
        for op in ops:
            row.append("{%d,%d,%d,%d,%d}" % op)
            if len(row) == 8:
                lines.append("        " + ",".join(row) + ",")
                row = []
        if row:
            lines.append("        " + ",".join(row) + ",")
        lines.append("    };")
        lines.append("    static constexpr short kOut[%d] = {%s};" % (ns, ",".join(str(s) for s in out)))
        lines.append("};")
    lines.append("")
    lines.append("// half-cleaner cascades of the cross-lane bitonic merges (fast_ml_common.hpp): inputs are bitonic")
    lines.append("// sequences; KEEP > 0: only the KEEP lowest / highest positions end up ordered")
    lines.append("template <int NS, int KEEP> struct FusedBitonic;")
    for ns, keep_ends in BITONIC:
        ops, out, n_slots, n_ce = build_bitonic(ns, keep_ends)
        verify_bitonic(ns, keep_ends, ops, out, n_slots)
        summary.append((ns, ("bitonic", keep_ends), n_ce, len(ops), n_slots))
        lines.append("template <> struct FusedBitonic<%d, %d> {" % (ns, keep_ends))
        lines.append("    static constexpr int kCount = %d, kSlots = %d, kComparators = %d;" % (len(ops), n_slots, n_ce))
        lines.append("    static constexpr FusedOp kOps[%d] = {" % len(ops))
        row = []
        for op in ops:
            row.append("{%d,%d,%d,%d,%d}" % op)
            if len(row) == 8:
                lines.append("        " + ",".join(row) + ",")
                row = []
        if row:
            lines.append("        " + ",".join(row) + ",")
        lines.append("    };")
        lines.append("    static constexpr short kOut[%d] = {%s};" % (ns, ",".join(str(s) for s in out)))
        lines.append("};")
    return "\n".join(lines) + "\n", summary


def main():
    text, summary = render()
    if "--check" in sys.argv:
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        if cur != text:
            print("sort_tables.inc is stale: run tools/gen_sort_tables.py")
            return 1
        print("sort_tables.inc is current")
        return 0
    with open(OUT, "w") as f:
        f.write(text)
    for ns, e, n_ce, n_ops, n_slots in summary:
        print("NS %3d  E %-18s comparators %4d  instr %4d -> %4d (%.1f %%)  slots %d"
              % (ns, e, n_ce, 2 * n_ce, n_ops, 100.0 * n_ops / (2 * n_ce), n_slots))
    return 0


if __name__ == "__main__":
    sys.exit(main())
