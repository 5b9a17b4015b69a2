"""The bench-diff guard (tools/check_bench.py, profiles/expected_also.json) on committed bench lines: it must
flag the library round 5 shipped (pass tails of the C3 tile and of winsor 24 grown by the stream pool, goal-seek 61 instead of
52 ms) and pass the bench line of the library that is committed."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(name):
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_bench.py"), os.path.join(ROOT, "profiles", name)],
                          capture_output=True, text=True)


def test_guard_flags_the_regressed_library_of_round_5():
    p = _run("r05_bench_default_with_cpu_baseline.json")
    assert p.returncode == 1, p.stdout
    flagged = [l for l in p.stdout.splitlines() if "<--" in l]
    assert any(l.startswith("C3 tile") and "tail" in l and "goal-seek" in l for l in flagged), p.stdout
    assert any(l.startswith("winsor24") and "tail" in l for l in flagged), p.stdout
    # (C5 may be flagged as well: the committed library's nontemporal result stores made the median faster than round 5's)
    assert all(l.startswith(("C3 tile", "winsor24", "C5")) for l in flagged), p.stdout


def test_guard_passes_the_committed_library():
    p = _run("r06_bench_final.json")
    assert p.returncode == 0, p.stdout
    assert "ok: no workload slower than its band" in p.stdout
