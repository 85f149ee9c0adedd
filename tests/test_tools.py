"""tools/poison_bisect.py narrows "the library depends on what a fresh allocation holds" down to one allocation's ordinal.  No such
dependence exists in the library (the GPU runs found none), so the bisection itself never ran there: here it runs against a stand-in
command that honours the same environment switches as csrc/fpx_api.hip: dmalloc_raw and aborts at one chosen allocation."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FAKE = textwrap.dedent('''
    import os, sys
    lo = int(os.environ.get("FPX_POISON_ORD_LO", "0")); hi = int(os.environ.get("FPX_POISON_ORD_HI", str(2**62)))
    poison = os.environ.get("FPX_POISON") == "1"
    log = os.environ.get("FPX_ALLOC_LOG")
    bad = int(sys.argv[1])
    f = open(log, "w") if log else None
    for ordn in range(100):
        if f:
            f.write("%d %d libfpx.so+0x%x\\n" % (ordn, 64 * (ordn + 1), 0x1000 + ordn)); f.flush()
        if poison and ordn == bad and lo <= ordn <= hi:
            os.abort()
''')


@pytest.mark.parametrize("bad", [0, 37, 99])
def test_poison_bisect_names_the_allocation(tmp_path, bad):
    fake = tmp_path / "fake.py"
    fake.write_text(FAKE)
    out = tmp_path / "out"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "poison_bisect.py"), str(out), "--", sys.executable, str(fake), str(bad)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    rep = (out / "poison_bisect.txt").read_text()
    assert "clean run rc 0" in rep
    assert f"ordinal {bad} alone: rc" in rep and f"ordinal {bad} alone: rc 0" not in rep
    assert f"-> {bad} {64 * (bad + 1)} libfpx.so+0x{0x1000 + bad:x}" in rep


def test_poison_bisect_says_so_when_nothing_depends_on_fresh_contents(tmp_path):
    fake = tmp_path / "fake.py"
    fake.write_text(FAKE)
    out = tmp_path / "out"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "poison_bisect.py"), str(out), "--", sys.executable, str(fake), "1000"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0
    assert "no dependence on fresh contents found" in (out / "poison_bisect.txt").read_text()
