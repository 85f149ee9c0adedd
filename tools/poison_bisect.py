#!/usr/bin/env python3
"""Which allocation's fresh contents does the library depend on?

FPX_POISON=1 fills every fresh device allocation of libfpx with 0xCD (csrc/fpx_api.hip: dmalloc_raw).  If the command then fails, this
script bisects over the ORDINALS of the process's allocations (FPX_POISON_ORD_LO / _HI) down to one and prints its size and callers
(return addresses inside libfpx.so: resolve with `llvm-symbolizer -e libfpx.so <offset>`).

    python tools/poison_bisect.py OUT_DIR -- python -m pytest -x -q -m gpu tests/test_gpu_parity.py::test_zipf_caps_multi_segment
"""
import os
import subprocess
import sys


def run(cmd, out, tag, lo=None, hi=None, timeout=180):
    env = dict(os.environ, FPX_POISON="1", FPX_ALLOC_LOG=os.path.join(out, "alloc_%s.log" % tag))
    if lo is not None:
        env["FPX_POISON_ORD_LO"] = str(lo)
        env["FPX_POISON_ORD_HI"] = str(hi)
    with open(os.path.join(out, "run_%s.log" % tag), "w") as f:
        try:
            rc = subprocess.run(cmd, env=env, stdout=f, stderr=subprocess.STDOUT, timeout=timeout).returncode
        except subprocess.TimeoutExpired:
            rc = -999
    return rc


def main():
    out = sys.argv[1]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    os.makedirs(out, exist_ok=True)
    rep = open(os.path.join(out, "poison_bisect.txt"), "w")

    def say(*a):
        print(*a, file=rep, flush=True)
        print(*a, flush=True)

    env0 = dict(os.environ)
    env0.pop("FPX_POISON", None)
    with open(os.path.join(out, "run_clean.log"), "w") as f:
        rc = subprocess.run(cmd, env=env0, stdout=f, stderr=subprocess.STDOUT, timeout=300).returncode
    say("clean run rc", rc)
    rc = run(cmd, out, "all")
    lines = open(os.path.join(out, "alloc_all.log")).read().splitlines()
    say("everything poisoned: rc", rc, "allocations logged", len(lines))
    if rc == 0:
        say("no dependence on fresh contents found")
        return 0
    lo, hi = 0, len(lines) - 1
    step = 0
    while lo < hi:
        mid = (lo + hi) // 2
        rc = run(cmd, out, "s%d" % step, lo, mid)
        say("ordinals %d..%d poisoned: rc %d" % (lo, mid, rc))
        if rc != 0:
            hi = mid
        else:
            lo = mid + 1
        step += 1
    rc = run(cmd, out, "one", lo, lo)
    say("ordinal %d alone: rc %d" % (lo, rc))
    for ln in lines[max(0, lo - 3):lo + 2]:
        say(("-> " if ln.split()[0] == str(lo) else "   ") + ln)
    return 0


if __name__ == "__main__":
    sys.exit(main())
