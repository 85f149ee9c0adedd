"""The parity suites once more with the library's alternative paths forced (the switches are read once per process, so every
variant runs in a process of its own):

* FPX_LOCAL_SORT_MAX=0   batch-wide radix sort of the keys also for small batches (the default sorts batches of up to 2^20
                         pairs per query in LDS)
* FPX_FAST=0             the general path only (host round trips between the stages, rocPRIM partition)
* FPX_LEAN_HEAD=4        the whole-block instantiation of the lean probe kernel instead of the partial fetch
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUITES = ["tests/test_gpu_parity.py", "tests/test_gpu_fuzz.py", "tests/test_gpu_golden.py", "tests/test_gpu_sharded_abi.py"]


@pytest.mark.parametrize("env", [{"FPX_LOCAL_SORT_MAX": "0"}, {"FPX_FAST": "0"}, {"FPX_LEAN_HEAD": "4"},
                                 {"FPX_LOCAL_SORT_MAX": "0", "FPX_FAST": "0", "FPX_LEAN_HEAD": "4"}],
                         ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()))
def test_parity_suites_on_the_alternative_paths(env):
    if os.environ.get("FPX_VARIANT_CHILD") == "1":
        pytest.skip("already inside a variant run")
    e = dict(os.environ, FPX_VARIANT_CHILD="1", **env)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + SUITES,
                       cwd=ROOT, env=e, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
