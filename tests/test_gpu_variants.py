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
* FPX_REC32=0            ... with 8-byte records in the bins (by default 4-byte ones where the doc ids leave room), the keys of every
                         batch ordered by our counting sort (FPX_ORDER_MIN_PAIRS=0), and in the same child
  FPX_POISON=1           every fresh device allocation of the library filled with 0xCD before it is handed out (csrc/fpx_api.hip:
                         dmalloc_raw): nothing may depend on what fresh device memory happens to hold -- usually zeros, which is how such a
                         dependence hides (tools/poison_bisect.py narrows a failure down to one allocation)
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




# (the three switches of the block form's paths run TOGETHER -- batch-wide sort, general path, whole-block lean kernel --: each on its
# own was another child of four suites, a fifth of the GPU suite's time between them; the batch-wide sort with the device-sized path
# and the partial-fetch lean kernel is what tests/test_gpu_fullsize.py's block-form runs take)
# (seven children.  Round 5 ran ten: the grouped form on the general path alone -- every workspace's first batch takes it in the other
# children --, the builder without inline doubles and forced bin sizes -- options folded into constants since --, and the poisoned
# allocations in a child of their own)
VARIANTS = [{"FPX_DIRECT": "0"},
            {"FPX_LOCAL_SORT_MAX": "0", "FPX_FAST": "0", "FPX_LEAN_HEAD": "4"},
            {"FPX_DIRECT_MIN_ITEMS": "0", "FPX_FUSE_MIN": "0"}, {"FPX_DIRECT_MIN_ITEMS": "0", "FPX_FUSE_MIN": "1"},
            {"FPX_DIRECT_MIN_ITEMS": "0", "FPX_FUSE_MIN": "1", "FPX_GROUP_PACKED": "1"},
            {"FPX_DIRECT_MIN_ITEMS": "0", "FPX_FUSE_MIN": "1", "FPX_BINNED": "0"},
            {"FPX_DIRECT_MIN_ITEMS": "0", "FPX_FUSE_MIN": "1", "FPX_REC32": "0", "FPX_ORDER_MIN_PAIRS": "0", "FPX_POISON": "1"}]


def _name(env):
    return ",".join(f"{k}={v}" for k, v in env.items())


def _suites(env):
    if env.get("FPX_FUSE_MIN") == "1":
        # the sub-variants of the grouped form: the suites that reach the switched code
        if "FPX_BINNED" in env:
            return ["tests/test_gpu_parity.py", "tests/test_gpu_direct.py"]
        if "FPX_POISON" in env:         # (the builders, the group's arenas, the searches' workspaces, merges and downloads; the bins' records, the windows' cells)
            return ["tests/test_gpu_parity.py", "tests/test_gpu_direct.py", "tests/test_gpu_merge.py", "tests/test_gpu_hashshard.py"]
        if "FPX_GROUP_PACKED" in env:
            # (every group of this child costs 64 / 137 GB of lines and seconds of mapping them, however small its segments: the suites
            # that reach the packed form's own code -- its probe kernel, its builder, its downloads, its window slices.  The golden
            # scenarios -- a snapshot, so a group, per step: 53 s of this child's 324 -- run on groups in the child above; what the packed form
            # adds is its lines, which the suites here, tests/test_gpu_query_wg.py and the full-size tests search)
            return [s for s in FUSED_SUITES if "test_gpu_api" not in s and "test_gpu_fuzz" not in s and "test_gpu_golden" not in s]
        return FUSED_SUITES
    return DIRECT_SUITES if "FPX_DIRECT_MIN_ITEMS" in env else SUITES


def _need_gb(env):
    """HBM a variant's child may hold at its peak.  A packed group's lines cost memory by the HASH SPACE, not by the items: 69 GB for
    up to eight columns, 137 GB for sixteen -- and tests/test_gpu_hashshard.py holds the unsharded group AND the ranks' windows of it
    (another 69 GB between them).  The directory + words form: 8.6 / 17 GB per group.  Everything else: small indexes in blocks."""
    # (measured the hard way: with 150 / 30 the packed child ran out of HBM next to four neighbours -- a download out of a packed group
    # takes 6 GB of line counts and bases on top of two groups)
    if env.get("FPX_GROUP_PACKED") == "1":
        return 170
    if env.get("FPX_FUSE_MIN") == "1":
        return 45
    return 15


def _free_gb():
    import torch
    free_b, _ = torch.cuda.mem_get_info()
    return free_b / 2**30


class _HbmScheduler:
    """Starts a variant's child when the device has room for what it may take (round 4 ran three children at a time whatever they
    needed: next to two neighbours the packed variant's groups did not fit once, its segments stayed in their blocks, and the test
    that asserts the grouped layout failed -- HBM exhaustion, not a race: alone, and with the room reserved, it passes)."""

    def __init__(self, jobs):
        import threading
        self.cv = threading.Condition()
        self.budget = _free_gb() - 15.0      # what the device has free now, before any child runs
        self.reserved = 0.0                  # GB promised to the running children (they take it gradually: the free HBM of the moment would over-admit)
        self.running = 0
        self.jobs = jobs
        import time
        self.t_start = time.time()

    def run(self, env):
        need = _need_gb(env)
        with self.cv:
            while self.running and (self.running >= self.jobs or self.reserved + need > self.budget):
                self.cv.wait()
            self.running += 1
            self.reserved += need
        import time
        t0 = time.time()
        try:
            e = dict(os.environ, FPX_VARIANT_CHILD="1", **env)
            r = None
            r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", "--durations=25"] + _suites(env),
                               cwd=ROOT, env=e, capture_output=True, text=True, timeout=1500)
            return r
        finally:
            out = os.path.join(ROOT, "gpurun_out")
            if os.path.isdir(out):               # (a GPU box run by tools/*.sh: how long every child took, and its slowest tests, for the suite's time budget)
                with open(os.path.join(out, "variant_times.txt"), "a") as f:
                    f.write(f"{_name(env)}: started {t0 - self.t_start:.0f} s into the module, took {time.time() - t0:.0f} s\n")
                    if r is not None:
                        f.write("".join("    " + ln + "\n" for ln in r.stdout.splitlines() if " call " in ln or " setup " in ln))
            with self.cv:
                self.running -= 1
                self.reserved -= need
                self.cv.notify_all()


@pytest.fixture(scope="module")
def variant_runs():
    """every variant's child process, as many at a time as the device's free HBM allows (each is a minute of small batches that leaves
    the GPU mostly idle: one after the other they were 11 of the suite's 14 minutes); a test waits for its own"""
    import concurrent.futures as cf
    if os.environ.get("FPX_VARIANT_CHILD") == "1":
        yield {}
        return
    from fpx_testlib import fpx
    c = fpx.Context(0)
    c.trim()                                 # (a line buffer this process may keep for its next group is not the children's to wait for)
    c.close()
    sched = _HbmScheduler(int(os.environ.get("FPX_VARIANT_JOBS", "6")))
    # the largest first: the packed variant starts on an empty device, the small ones fill in around it
    order = sorted(VARIANTS, key=_need_gb, reverse=True)
    with cf.ThreadPoolExecutor(len(VARIANTS)) as pool:
        yield {_name(env): pool.submit(sched.run, env) for env in order}


@pytest.mark.parametrize("env", VARIANTS, ids=_name)
def test_parity_suites_on_the_alternative_paths(env, variant_runs):
    if os.environ.get("FPX_VARIANT_CHILD") == "1":
        pytest.skip("already inside a variant run")
    r = variant_runs[_name(env)].result()
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-1000:]
