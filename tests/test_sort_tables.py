"""The generated sorting networks (nightlight_amd/csrc/sort_tables.inc): the committed file is
what tools/gen_sort_tables.py produces, and every network -- simulated on the CPU -- sorts the
ranks its kernel reads and keeps the right set of samples everywhere else."""
import importlib.util
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gen():
    spec = importlib.util.spec_from_file_location("gen_sort_tables", os.path.join(ROOT, "tools", "gen_sort_tables.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _parse(text):
    """-> {(ns, e0, e1, e2, e3): (ops, out, n_slots, n_comparators)} from the committed header"""
    nets = {}
    for m in re.finditer(r"FusedNet<(\d+), (\d+), (\d+), (\d+), (\d+)> \{(.*?)\n\};", text, re.S):
        key = tuple(int(x) for x in m.groups()[:5])
        body = m.group(6)
        count, slots, ces = (int(x) for x in re.search(r"kCount = (\d+), kSlots = (\d+), kComparators = (\d+)", body).groups())
        ops_txt = re.search(r"kOps\[\d+\] = \{(.*?)\n    \};", body, re.S).group(1)
        ops = [tuple(int(x) for x in t.split(",")) for t in re.findall(r"\{([\d,]+)\}", ops_txt)]
        out = [int(x) for x in re.search(r"kOut\[\d+\] = \{([\d,]+)\}", body).group(1).split(",")]
        assert len(ops) == count and len(out) == key[0]
        nets[key] = (ops, out, slots, ces)
    return nets


def test_committed_tables_are_current():
    gen = _gen()
    text, _ = gen.render()
    assert open(gen.OUT).read() == text, "run tools/gen_sort_tables.py"


def test_every_network_sorts_what_its_kernel_reads():
    gen = _gen()
    nets = _parse(open(gen.OUT).read())
    assert set(nets) == {(ns,) + e for ns, e in gen.variants()}
    rng = np.random.default_rng(7)
    for (ns, *e), (ops, out, slots, ces) in nets.items():
        e = tuple(e)
        assert max(max(o[1:]) for o in ops) < slots and max(out) < slots
        assert len(ops) < 2 * ces                                  # fewer instructions than comparators * 2
        x = np.concatenate([rng.standard_normal((200, ns)) * 50 + 1000,
                            rng.integers(0, 3, (200, ns)).astype(np.float64)]).astype(np.float32)
        x[::5, ns - 5:] = np.inf                                   # missing samples sort last
        x[1::9, 0] = -np.inf
        y = gen.simulate(ops, out, slots, x)
        ref = np.sort(x, axis=1)
        exact, sets = gen.needed(ns, e)
        for a, b in exact:
            assert np.array_equal(y[:, a:b], ref[:, a:b]), (ns, e, a, b)
        for a, b in sets:
            assert np.array_equal(np.sort(y[:, a:b], axis=1), ref[:, a:b]), (ns, e, a, b)


@pytest.mark.parametrize("p", [1, 2, 4, 8])
def test_zero_one_exhaustive_small_merges(p):
    """All 2^(2p) zero-one inputs made of two sorted runs -- and, for the full sort of 2p <= 16
    elements, all 2^n zero-one inputs (the 0-1 principle covers min/med/max networks)."""
    gen = _gen()
    ns = 2 * p
    if ns < 8:
        pytest.skip("smallest generated network has 8 elements")
    ops, out, slots, _ = gen.build(ns, (0, 0, 0, 0))
    bits = ((np.arange(1 << ns)[:, None] >> np.arange(ns)[None, :]) & 1).astype(np.float32)
    y = gen.simulate(ops, out, slots, bits)
    assert np.array_equal(y, np.sort(bits, axis=1))


def _parse_bitonic(text):
    nets = {}
    for m in re.finditer(r"FusedBitonic<(\d+), (\d+)> \{(.*?)\n\};", text, re.S):
        ns, keep = int(m.group(1)), int(m.group(2))
        body = m.group(3)
        count, slots, ces = (int(x) for x in re.search(r"kCount = (\d+), kSlots = (\d+), kComparators = (\d+)", body).groups())
        ops_txt = re.search(r"kOps\[\d+\] = \{(.*?)\n    \};", body, re.S).group(1)
        ops = [tuple(int(x) for x in t.split(",")) for t in re.findall(r"\{([\d,]+)\}", ops_txt)]
        out = [int(x) for x in re.search(r"kOut\[\d+\] = \{([\d,]+)\}", body).group(1).split(",")]
        assert len(ops) == count and len(out) == ns
        nets[(ns, keep)] = (ops, out, slots, ces)
    return nets


def test_bitonic_merge_tables_sort_bitonic_columns():
    # the half-cleaner cascades of the cross-lane merges (fast_ml_common.hpp): inputs are what a
    # mirror / distance stage leaves in a lane -- ascending then descending, any rotation, reversed
    gen = _gen()
    nets = _parse_bitonic(open(gen.OUT).read())
    assert set(nets) == set(gen.BITONIC)
    rng = np.random.default_rng(11)
    for (ns, keep), (ops, out, slots, ces) in nets.items():
        assert len(ops) < 2 * ces and max(out) < slots
        rows = []
        for r in range(600):
            vals = rng.standard_normal(ns) * 40 + 900 if r % 2 else rng.integers(0, 4, ns).astype(np.float64)
            if r % 5 == 0:
                vals[:7] = np.inf
            cut = int(rng.integers(0, ns + 1))
            seq = np.concatenate([np.sort(vals[:cut]), np.sort(vals[cut:])[::-1]])
            seq = np.roll(seq, int(rng.integers(0, ns))) if r % 3 == 0 else (seq[::-1] if r % 3 == 1 else seq)
            rows.append(seq)
        x = np.array(rows, np.float32)
        y = gen.simulate(ops, out, slots, x)
        ref = np.sort(x, axis=1)
        k = keep if keep else ns // 2
        assert np.array_equal(y[:, :k], ref[:, :k]) and np.array_equal(y[:, ns - k:], ref[:, ns - k:]), (ns, keep)
        assert np.array_equal(np.sort(y, axis=1), ref), (ns, keep)
