"""The parity suites once more with the library's alternative paths forced (the switches are read once per process, so every
variant runs in a process of its own):

* FPX_LOCAL_SORT_MAX=0   batch-wide radix sort of the keys also for small batches (the default sorts batches of up to 2^20
                         pairs per query in LDS)
* FPX_FAST=0             the general path only (host round trips between the stages, rocPRIM partition)
* FPX_LEAN_HEAD=4        the whole-block instantiation of the lean probe kernel instead of the partial fetch
* FPX_DIRECT_MIN_ITEMS=0 EVERY file segment in its direct-addressed form (by default segments of >= 2^20 items): searches, counters, downloads and merges must be what the block form gives.
                         With FPX_FUSE_MIN=0: every segment on its own (k_probe_direct)
* FPX_FUSE_MIN=1         with it: every direct-addressed segment a column of a GROUP, even one alone (k_probe_group<8 | 16>, built
                         chunk by chunk from the blocks; by default groups of 2..16) -- the records binned in the probe kernel's
                         flush and scored a bin per workgroup (k_score_bin) on every batch after a workspace's first
* FPX_GROUP_PACKED=1     ... every group in the PACKED form (k_probe_pgroup: 128-byte lines of 4 / 8 hash values with their words inside;
                         by default only groups dense enough for it -- the full-size indexes of tests/test_gpu_fullsize.py)
* FPX_BINNED=0           ... with the two-level partition + k_score instead (the path of mixed snapshots)
* FPX_INLINE_DOUBLES=0   ... with every hash of several docs behind a list reference (no inline doubles)
* FPX_REC32=0            ... with 8-byte records in the bins (by default 4-byte ones where the doc ids leave room), bins of eight
                         queries whatever the batch, the keys of every batch ordered by our counting sort (FPX_ORDER_MIN_PAIRS=0)
* FPX_DIRECT=0           no segment direct-addressed: segments of >= 2^20 items (direct-addressed by default) are searched in
                         their blocks by the lean kernel (tests/test_gpu_fullsize.py compares the two forms at full size)
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUITES = ["tests/test_gpu_parity.py", "tests/test_gpu_fuzz.py", "tests/test_gpu_golden.py", "tests/test_gpu_sharded_abi.py"]
DIRECT_SUITES = SUITES + ["tests/test_gpu_builder.py", "tests/test_gpu_merge.py", "tests/test_gpu_api.py", "tests/test_gpu_frontend.py",
                          "tests/test_gpu_hashsplit.py", "tests/test_gpu_sharded.py"]
# (a group's directory is 8.6 GB whatever the segments' size, built for every snapshot: these variants run the suites with the
# fewest snapshots)
FUSED_SUITES = ["tests/test_gpu_golden.py", "tests/test_gpu_parity.py", "tests/test_gpu_api.py", "tests/test_gpu_merge.py", "tests/test_gpu_direct.py",
                "tests/test_gpu_hashshard.py", "tests/test_gpu_fuzz.py::test_fuzz_lean_sized_worlds"]


FUSED_SHORT = ["tests/test_gpu_golden.py", "tests/test_gpu_parity.py", "tests/test_gpu_direct.py", "tests/test_gpu_hashshard.py"]


VARIANTS = [{"FPX_DIRECT": "0"}, {"FPX_LOCAL_SORT_MAX": "0"}, {"FPX_FAST": "0"}, {"FPX_LEAN_HEAD": "4"},
            {"FPX_LOCAL_SORT_MAX": "0", "FPX_FAST": "0", "FPX_LEAN_HEAD": "4"},
            {"FPX_DIRECT_MIN_ITEMS": "0", "FPX_FUSE_MIN": "0"}, {"FPX_DIRECT_MIN_ITEMS": "0", "FPX_FUSE_MIN": "1"},
            {"FPX_DIRECT_MIN_ITEMS": "0", "FPX_FUSE_MIN": "1", "FPX_FAST": "0", "FPX_LOCAL_SORT_MAX": "0"},
            {"FPX_DIRECT_MIN_ITEMS": "0", "FPX_FUSE_MIN": "1", "FPX_GROUP_PACKED": "1"},
            {"FPX_DIRECT_MIN_ITEMS": "0", "FPX_FUSE_MIN": "1", "FPX_BINNED": "0"},
            {"FPX_DIRECT_MIN_ITEMS": "0", "FPX_FUSE_MIN": "1", "FPX_INLINE_DOUBLES": "0", "FPX_BIN_Q_LOG2": "2"},
            {"FPX_DIRECT_MIN_ITEMS": "0", "FPX_FUSE_MIN": "1", "FPX_REC32": "0", "FPX_BIN_Q_LOG2": "3", "FPX_ORDER_MIN_PAIRS": "0"}]


def _name(env):
    return ",".join(f"{k}={v}" for k, v in env.items())


def _run_variant(env):
    e = dict(os.environ, FPX_VARIANT_CHILD="1", **env)
    suites = FUSED_SUITES if env.get("FPX_FUSE_MIN") == "1" else DIRECT_SUITES if "FPX_DIRECT_MIN_ITEMS" in env else SUITES
    if env.get("FPX_FUSE_MIN") == "1" and len(env) > 2:          # the sub-variants of the grouped form: the suites that reach the switched code
        suites = FUSED_SHORT
    return subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + suites,
                          cwd=ROOT, env=e, capture_output=True, text=True, timeout=1500)


@pytest.fixture(scope="module")
def variant_runs():
    """every variant's child process, THREE at a time (each is a minute of small batches that leaves the GPU mostly idle: one
    after the other they were 11 of the suite's 14 minutes); a test waits for its own"""
    import concurrent.futures as cf
    if os.environ.get("FPX_VARIANT_CHILD") == "1":
        yield {}
        return
    with cf.ThreadPoolExecutor(int(os.environ.get("FPX_VARIANT_JOBS", "3"))) as pool:
        yield {_name(env): pool.submit(_run_variant, env) for env in VARIANTS}


@pytest.mark.parametrize("env", VARIANTS, ids=_name)
def test_parity_suites_on_the_alternative_paths(env, variant_runs):
    if os.environ.get("FPX_VARIANT_CHILD") == "1":
        pytest.skip("already inside a variant run")
    r = variant_runs[_name(env)].result()
    if r.returncode != 0:
        # Three children share the one GPU (up to 137 GB of group lines each in the packed variant, process groups, dozens of contexts): a
        # child that fails next to the others is run once more ON ITS OWN -- after all the others have finished -- and that run decides.
        # Seen once in round 4: the packed variant failed in the full suite and passed 76 / 76 alone and next to two neighbours.
        first = r.stdout[-1500:] + r.stderr[-500:]
        for f in variant_runs.values():
            f.result()
        r = _run_variant(env)
        sys.stderr.write(f"\nvariant {_name(env)} failed next to the other children and was run again alone (rc {r.returncode}); first failure:\n{first}\n")
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
