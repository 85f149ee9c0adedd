"""GPU parity tests: the HIP path through the C ABI vs the CPU oracle, bit-exact {id, score} lists,
plus the reference's scanned-blocks / scanned-docs totals."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
# every file segment direct-addressed (tests/test_gpu_variants.py): results and counters must not change, but the assertions
# about which block kernel carried a batch do not apply
DIRECT_FORCED = os.environ.get("FPX_DIRECT_MIN_ITEMS") == "0"


@pytest.fixture(scope="module")
def env():
    from fpx_testlib import fpx, oracle, Pair
    ctx = fpx.Context(0)
    yield fpx, oracle, Pair, ctx


def _queries(fpx, seed, nq, ndocs, H, qlen, dist=0, first_doc=1):
    flat, off, targets = fpx.synth.make_queries(seed, 1234, nq, ndocs, H, query_len=qlen, dist=dist, first_doc=first_doc)
    return [flat[int(off[i]):int(off[i + 1])] for i in range(nq)], targets


def test_single_segment_uniform(env):
    fpx, oracle, Pair, ctx = env
    seed, ndocs, H = 11, 20000, 64
    p = Pair(ctx)
    p.add_file(fpx.synth.synth_items(seed, 1, ndocs, H), 1, ndocs, 1, np.arange(1, ndocs + 1))
    p.finish()
    qs, targets = _queries(fpx, seed, 64, ndocs, H, 200)
    got, st = p.check(qs, fpx.http_options())
    assert all(g and g[0][0] == int(t) for g, t in zip(got, targets))
    assert st.algorithmic_bytes == st.scanned_blocks * 512


def test_golden_api_vectors(env):
    """tests/test_fingerprint_api.py:5-52,102-260 through the GPU path (memory segments)."""
    fpx, oracle, Pair, ctx = env
    p = Pair(ctx)
    p.add_memory_changes([("insert", 1, [101, 201, 301]), ("insert", 2, [102, 202, 302])], 1)
    p.finish()
    got, _ = p.check([[101, 201, 301], [101, 201, 301, 102, 202, 302]], fpx.http_options())
    assert got == [[(1, 3)], [(1, 3), (2, 3)]]
    p = Pair(ctx)
    p.add_memory_changes([("insert", 1, [100, 200, 300])], 1)
    p.add_memory_changes([("insert", 1, [100, 200, 999])], 2)
    p.finish()
    got, _ = p.check([[100, 200, 300], [100, 200, 999]], fpx.http_options())
    assert got == [[(1, 2)], [(1, 3)]]
    p = Pair(ctx)
    p.add_memory_changes([("insert", 1, [101, 201, 301]), ("insert", 2, [102, 202, 302])], 1)
    p.add_memory_changes([("delete", 1), ("delete", 2)], 2)
    p.finish()
    got, _ = p.check([[101, 201, 301, 102, 202, 302]], fpx.http_options())
    assert got == [[]]


def test_zipf_caps_multi_segment(env):
    """Hot hashes spanning > 4 blocks and > 1000 docs exercise MAX_BLOCKS_PER_HASH / MAX_DOCS_PER_HASH
    (src/FileSegment.zig:25-26,171-174); 3 segments, contiguous id ranges."""
    fpx, oracle, Pair, ctx = env
    seed, H, per = 5, 96, 30000
    p = Pair(ctx)
    for s in range(3):
        lo = s * per + 1
        p.add_file(fpx.synth.synth_items(seed, lo, per, H, dist=1), lo, lo + per - 1, s + 1, np.arange(lo, lo + per))
    p.finish()
    qs, _ = _queries(fpx, seed, 96, 3 * per, H, 300, dist=1)
    got, st = p.check(qs, fpx.http_options())
    # the hottest pool value alone: every segment stops at the 4-block cap / after > 1000 docs
    hot0 = int(fpx.synth.mix64(np.uint64(seed) ^ np.uint64(0x5bd1e9955bd1e995)) >> np.uint64(32))
    _, st1 = p.check([[hot0]], fpx.SearchOptions(10, 1, 10))
    assert st1.probes == 3 and st1.scanned_blocks > 3 and st1.scanned_docs > 1000
    # legacy options (src/legacy.zig:192-198): limit 500, min_score 1 -> thousands of candidates per query
    p.check(qs[:24], fpx.SearchOptions(500, 1, 10))
    p.check(qs[:8], fpx.SearchOptions(100000, 1, 0))


@pytest.mark.parametrize("block_size", [64, 100, 128, 256, 1024, 4096])
def test_block_sizes(env, block_size):
    """block_size comes from the segment header, 64..4096 (src/filefmt.zig:236); small id ranges make
    blocks with > 128 items (multi-chunk decode), 4096-B blocks hold ~1000 items."""
    fpx, oracle, Pair, ctx = env
    seed, ndocs, H = 21 + block_size, 3000, 48
    p = Pair(ctx)
    p.add_file(fpx.synth.synth_items(seed, 1, ndocs, H, dist=1), 1, ndocs, 1, np.arange(1, ndocs + 1), block_size=block_size)
    p.finish()
    qs, _ = _queries(fpx, seed, 32, ndocs, H, 120, dist=1)
    got, st = p.check(qs, fpx.http_options())
    # SURVEY 8(d)'s numerator: visited blocks x the SEGMENT's block size (src/filefmt.zig:29-31,71) -- in every storage form (the
    # variants of tests/test_gpu_variants.py run this with the segment direct-addressed, alone and as a group's column)
    assert st.scanned_blocks > 0 and st.algorithmic_bytes == st.scanned_blocks * block_size, (st.algorithmic_bytes, st.scanned_blocks, block_size)
    # ... also with a second segment of ANOTHER block size in the snapshot (the two never share a group)
    p2 = Pair(ctx)
    p2.add_file(fpx.synth.synth_items(seed, 1, ndocs, H, dist=1), 1, ndocs, 1, np.arange(1, ndocs + 1), block_size=block_size)
    p2.add_file(fpx.synth.synth_items(seed + 1, ndocs + 1, ndocs, H, dist=1), ndocs + 1, 2 * ndocs, 2, np.arange(ndocs + 1, 2 * ndocs + 1), block_size=512)
    p2.finish()
    singles = [oracle.Snapshot([sg], []) for sg in p2.orc_file]
    want_bytes = 0
    for q in qs[:12]:
        for one, bs in zip(singles, (block_size, 512)):
            want_bytes += one.search(q, with_stats=True)[1].scanned_blocks * bs
    got2, st2 = p2.check(qs[:12], fpx.http_options())
    assert st2.algorithmic_bytes == want_bytes, (st2.algorithmic_bytes, want_bytes)
    p.check(qs[:8], fpx.SearchOptions(500, 1, 10))


def test_supersession_and_tombstones(env):
    """Re-inserted docs and tombstones in newer file/memory segments hide older postings
    (src/common.zig:121-129,158; src/Index.zig:133-149), combined with top-k truncation."""
    fpx, oracle, Pair, ctx = env
    seed, H, n = 77, 32, 6000
    rng = np.random.default_rng(3)
    p = Pair(ctx)
    p.add_file(fpx.synth.synth_items(seed, 1, n, H), 1, n, 1, np.arange(1, n + 1))
    # segment 2: 300 docs of segment 1 re-inserted with fresh hashes + 150 tombstones + 1000 new docs
    re_ids = np.sort(rng.choice(np.arange(1, n + 1), 300, replace=False)).astype(np.uint64)
    tomb = np.sort(rng.choice(np.setdiff1d(np.arange(1, n + 1), re_ids), 150, replace=False)).astype(np.uint64)
    new_ids = np.arange(n + 1, n + 1001, dtype=np.uint64)
    ids2 = np.concatenate([re_ids, new_ids])
    h2 = fpx.synth.synth_hashes(seed + 1, ids2, H).astype(np.uint64)
    items2 = np.sort(((h2 << np.uint64(32)) | ids2[:, None]).ravel())
    docs2 = np.concatenate([ids2, tomb]).astype(np.uint32)
    alive2 = np.concatenate([np.ones(len(ids2), np.uint8), np.zeros(len(tomb), np.uint8)])
    p.add_file(items2, int(docs2.min()), int(docs2.max()), 2, docs2, alive2)
    # memory segments: an overwrite of a doc from segment 2 and a delete of a doc from segment 1
    victim = int(re_ids[0])
    p.add_memory_changes([("insert", victim, fpx.synth.synth_hashes(seed + 2, [victim], H)[0].tolist())], 3)
    p.add_memory_changes([("delete", int(np.setdiff1d(np.arange(1, n + 1), np.concatenate([re_ids, tomb]))[0]))], 4)
    p.finish()
    # queries: old hashes of re-inserted docs (must NOT be found), their new hashes (found), tombstoned docs, plain docs
    def q_of(seed_, doc):
        return fpx.synth.synth_hashes(seed_, [doc], H)[0]
    qs = [q_of(seed, int(d)) for d in re_ids[:10]] + [q_of(seed + 1, int(d)) for d in re_ids[:10]]
    qs += [q_of(seed, int(d)) for d in tomb[:10]] + [q_of(seed, d) for d in range(1, 11)]
    qs += [q_of(seed + 2, victim), q_of(seed + 1, victim)]
    got, _ = p.check(qs, fpx.SearchOptions(10, 1, 10))
    assert all(not g or g[0][0] != int(d) for g, d in zip(got[:10], re_ids[:10]))   # old versions are hidden
    assert got[10] == [] and got[11][0] == (int(re_ids[1]), H)                      # victim overwritten again
    assert all(not g or g[0][0] != int(d) for g, d in zip(got[20:30], tomb[:10]))   # tombstones hide
    assert got[-2][0] == (victim, H)


def test_query_edge_cases(env):
    """empty query, duplicate hashes (src/Index.zig:1056-1096), hash 0 / 0xFFFFFFFF, hashes beyond every
    block, min_score defaults from the RAW length (src/MultiIndex.zig:304), ties broken by id."""
    fpx, oracle, Pair, ctx = env
    p = Pair(ctx)
    items = oracle.pack_items([(0, 5), (0, 6), (7, 5), (0xFFFFFFFF, 5), (0xFFFFFFFF, 9), (100, 9), (100, 5), (100, 6)])
    items = np.sort(items)
    p.add_file(items, 5, 9, 1, [5, 6, 9])
    p.finish()
    qs = [[], [100, 100, 100], [0, 0xFFFFFFFF, 7, 100], [0xFFFFFFFF], [1, 2, 3], [100] * 45, [0xFFFFFFFE, 5, 8]]
    got, _ = p.check(qs, fpx.http_options())
    assert got[0] == [] and got[1] == [(5, 1), (6, 1), (9, 1)] and got[2][0] == (5, 4)
    assert got[5] == []     # 45 raw hashes -> min_score 3 even though only one unique hash
    p.check(qs, fpx.SearchOptions(2, 1, 100))
    p.check(qs, fpx.SearchOptions(1, 2, 50))


def test_single_search_entry_point(env):
    """fpx_search (one query, IndexReader.search signature) agrees with the batched entry point."""
    fpx, oracle, Pair, ctx = env
    seed, ndocs, H = 3, 5000, 40
    p = Pair(ctx)
    p.add_file(fpx.synth.synth_items(seed, 1, ndocs, H), 1, ndocs, 1, np.arange(1, ndocs + 1))
    p.finish()
    qs, targets = _queries(fpx, seed, 4, ndocs, H, 80)
    for q, t in zip(qs, targets):
        r = fpx.SearchResults(fpx.http_options())
        p.reader.search(q, r)
        assert r.getResults() == p.osnap.search(q) and r.getResults()[0][0] == int(t)


def test_insert_many_50k_golden(env):
    """tests/test_fingerprint_api.py:67-99 / test_parallel_loading.py:11-67 on the GPU: exactly [(100, 100)]."""
    import random
    fpx, oracle, Pair, ctx = env
    n, max_hash, nseg = 50000, 2 ** 18, 3
    per = (n + nseg - 1) // nseg
    p = Pair(ctx)
    for s in range(nseg):
        lo, hi = s * per + 1, min(n, (s + 1) * per)
        hs = np.empty((hi - lo + 1, 100), np.uint64)
        for i in range(lo, hi + 1):
            rng = random.Random(i)
            hs[i - lo] = [rng.randint(0, max_hash) for _ in range(100)]
        ids = np.arange(lo, hi + 1, dtype=np.uint64)
        p.add_file(np.sort(((hs << np.uint64(32)) | ids[:, None]).ravel()), lo, hi, s + 1, ids.astype(np.uint32))
    p.finish()
    rng = random.Random(100)
    q = [rng.randint(0, max_hash) for _ in range(100)]
    got, _ = p.check([q], fpx.http_options())
    assert got == [[(100, 100)]]


def test_duplicate_postings_score_above_query_length(env):
    """A doc that holds the same hash several times scores once per posting, so a score can exceed the
    number of query hashes (segments keep duplicate items, src/MemorySegment.zig:139)."""
    fpx, oracle, Pair, ctx = env
    p = Pair(ctx)
    items = np.sort(oracle.pack_items([(50, 7)] * 9 + [(50, 8)] * 2 + [(60, 7)]))
    p.add_file(items, 7, 8, 1, [7, 8])
    p.add_memory(np.sort(oracle.pack_items([(50, 9)] * 5)), 9, 9, 2, [9])
    p.finish()
    got, _ = p.check([[50], [50, 60], [60]] * 40, fpx.SearchOptions(10, 1, 10))
    assert got[0] == [(7, 9), (9, 5), (8, 2)] and got[1][0] == (7, 10)


def test_cpp_host_mirror_example(env):
    """acoustid-index_amd/host/fpx.hpp: the reference's round-trip and duplicate-hash tests through the C++ mirror."""
    import os
    import subprocess
    from fpx_testlib import ROOT
    host = os.path.join(ROOT, "acoustid-index_amd", "host")
    subprocess.check_call(["bash", os.path.join(host, "build_host.sh")], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(host, "example_search")], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr


@pytest.mark.parametrize("presence", [False, True])
@pytest.mark.parametrize("dist", [0, 1])
def test_lean_kernel_and_deferred_pass(env, dist, presence, monkeypatch):
    """Batches with >= 2^20 probes on 512-B segments run k_probe_lean8 (two-level match) and finish the rows it
    cannot handle (wide deltas, runs crossing quads/blocks, > 32 quads) with the deferred generic pass.
    dist = 1 (hot pool) makes a large share of the probes take the deferred pass.
    presence: with / without the segments' presence bitmaps (default: on for segments of >= 2^20 items) -- probes of
    absent hashes are answered (and counted as the reference counts them) without their block being fetched."""
    fpx, oracle, Pair, ctx = env
    if not DIRECT_FORCED:
        monkeypatch.setenv("FPX_DIRECT", "0")           # (segments of >= 2^20 items are direct-addressed by default: this test is the block kernels')
    monkeypatch.setenv("FPX_PRESENCE_MIN_ITEMS", "1" if presence else str(1 << 62))
    seed, H, per, S = 91 + dist, 128, 9000, 3          # 1.15 M items per segment: 2-byte hash deltas, lean-eligible
    p = Pair(ctx)
    for s in range(S):
        lo = s * per + 1 + (70000 if s else 0)          # ids >= 65536 give 3-byte docids (<= 32 quads per block)
        p.add_file(fpx.synth.synth_items(seed, lo, per, H, dist=dist), lo, lo + per - 1, s + 1, np.arange(lo, lo + per))
    p.finish()
    nq = 360
    flat, off, targets = fpx.synth.make_queries(seed, 77, nq, per, H, query_len=1000, dist=dist, first_doc=1)
    qs = [flat[int(off[i]):int(off[i + 1])] for i in range(nq)]
    got, st = p.check(qs, fpx.http_options())
    assert st.probes >= (1 << 20) - 4096
    if DIRECT_FORCED:
        assert all(g and g[0][0] == int(t) for g, t in zip(got, targets))
        return
    assert 0 < st.probe_kernel_bytes <= st.algorithmic_bytes                            # the lean kernel worked ...
    if presence:
        assert st.probe_kernel_fetched_bytes < st.probe_kernel_bytes // 2               # 1.15 M items: most query hashes are absent
    else:
        assert st.probe_kernel_fetched_bytes >= st.probe_kernel_bytes                   # every visited block was read (+ gaps)
    if dist == 1:
        assert st.probe_kernel_bytes < st.algorithmic_bytes                             # ... and so did the deferred pass
    if dist == 0:
        assert st.generic_iters < st.probes // 8        # the lean path carried the bulk
    assert all(g and g[0][0] == int(t) for g, t in zip(got, targets))


def test_score_table_multi_pass_and_many_candidates(env):
    """k_score sizes its LDS table for the AVERAGE hits per query; a query far above the average is counted in
    several passes over disjoint doc classes.  One heavy query (hot hashes, min_score 1, limit 500) rides in a
    batch of light ones."""
    fpx, oracle, Pair, ctx = env
    seed, H, per = 401, 96, 40000
    p = Pair(ctx)
    p.add_file(fpx.synth.synth_items(seed, 1, per, H, dist=1), 1, per, 1, np.arange(1, per + 1))
    p.add_file(fpx.synth.synth_items(seed, per + 1, per, H, dist=1), per + 1, 2 * per, 2, np.arange(per + 1, 2 * per + 1))
    p.finish()
    hot = [int(fpx.synth.mix64(np.uint64(seed) ^ np.uint64(0x5bd1e9955bd1e995) ^ (np.uint64(k) << np.uint64(32))) >> np.uint64(32))
           for k in range(64)]
    heavy = np.array(hot, dtype=np.uint32)                     # 64 hot values x up to ~2000 docs each
    light = [fpx.synth.synth_hashes(seed, [d], H, 1)[0][:8] for d in range(1, 120)]
    opts = [fpx.SearchOptions(500, 1, 0)] + [fpx.SearchOptions(5, 1, 10)] * len(light)
    got, st = p.check([heavy] + light, opts)
    assert len(got[0]) == 500 and st.hits > 20000 and st.candidates > 10000


@pytest.mark.parametrize("name,ndocs,H,S,batch,qlen,limit", [
    ("configs[1] / 1000: one segment, batch 1024", 10000, 256, 1, 1024, 1000, 40),
    ("configs[2] / 1000: 16 segments", 100000, 256, 16, 256, 1000, 40),
    ("configs[4] / 1000: 120 hashes, limit 100", 128000, 120, 8, 512, 1000, 100),
])
def test_baseline_config_mini_twins(env, name, ndocs, H, S, batch, qlen, limit):
    """The /1000 twins of the BASELINE configs: same code paths, sizes the oracle finishes in seconds."""
    fpx, oracle, Pair, ctx = env
    seed, per = 20260928, ndocs // S
    p = Pair(ctx)
    for s in range(S):
        lo = s * per + 1
        g = fpx.FileSegment.synth(ctx, seed, lo, per, H, 0, 512, s + 1)           # built on the GPU ...
        blocks, index = g.download()
        p.gpu_segs.append(g)
        p.orc_file.append(oracle.file_segment(blocks, 512, index, lo, lo + per - 1, s + 1, np.arange(lo, lo + per)))
    p.finish()
    flat, off, targets = fpx.synth.make_queries(seed, 4242, batch, per * S, H, query_len=qlen)
    qs = [flat[int(off[i]):int(off[i + 1])] for i in range(batch)]
    got, st = p.check(qs, fpx.http_options(limit=limit))
    assert all(g and g[0][0] == int(t) for g, t in zip(got, targets))
    assert st.probes >= batch * 900 * S


def test_cpp_request_coalescer(env):
    """acoustid-index_amd/host/fpx_coalescer.hpp: 64 threads of single searches served as device batches."""
    import os
    import subprocess
    from fpx_testlib import ROOT
    host = os.path.join(ROOT, "acoustid-index_amd", "host")
    subprocess.check_call(["bash", os.path.join(host, "build_host.sh")], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(host, "test_coalescer")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("ok"), out.stdout + out.stderr


@pytest.mark.parametrize("presence", [False, True])
@pytest.mark.parametrize("dist", [0, 1])
def test_lean_kernel_with_supersession_tombstones_and_wide_docids(env, dist, presence, monkeypatch):
    """The lean path on segments whose docs are partly superseded (per-posting `dead` filter), with a segment of sparse
    ids (docid deltas of 3 and 4 bytes, ids above 2^24), tombstones and inserts in memory segments on top."""
    fpx, oracle, Pair, ctx = env
    monkeypatch.setenv("FPX_PRESENCE_MIN_ITEMS", "1" if presence else str(1 << 62))
    seed, H, per = 733, 128, 9000                      # 1.15 M items per file segment: lean-eligible
    p = Pair(ctx)
    # dist = 1: hot hashes with ~1000 docs per probe overflow the workgroups' hit staging, so superseded docs are dropped both
    # at the staged flush and on the direct-append path
    a = fpx.synth.synth_items(seed, 1, per, H, dist=dist)
    p.add_file(a, 1, per, 1, np.arange(1, per + 1))
    # segment B re-inserts ids 5001..9000 with fresh hashes (A's copies become dead) and adds 9001..14000
    b = fpx.synth.synth_items(seed + 1, 5001, per, H, dist=dist)
    p.add_file(b, 5001, 5000 + per, 2, np.arange(5001, 5001 + per))
    # segment C: sparse ids 20000 + 3001 * i (up to 27 M: 4-byte docid values, 2- and 3-byte deltas inside runs)
    c = fpx.synth.synth_items(seed + 2, 1, per, H, dist=dist)
    ids_c = (20000 + 3001 * np.arange(per)).astype(np.uint64)
    c = np.sort((c & ~np.uint64(0xFFFFFFFF)) | ids_c[(c & np.uint64(0xFFFFFFFF)).astype(np.int64) - 1])
    p.add_file(c, int(ids_c[0]), int(ids_c[-1]), 3, ids_c.astype(np.uint32))
    # memory segments: tombstones over all three, one overwrite, one fresh doc
    h_a = ((a[(a & np.uint64(0xFFFFFFFF)) == 77][:5]) >> np.uint64(32)).astype(np.uint32).tolist()
    p.add_memory_changes([("delete", 10), ("delete", 6000), ("delete", int(ids_c[5])), ("insert", 77, h_a + [1, 2, 3])], 4)
    p.add_memory_changes([("insert", 30000000, [9, 8, 7]), ("delete", 12000)], 5)
    p.finish()
    rng = np.random.default_rng(3)
    nq = 360
    qs = []
    for i in range(nq):
        src = (a, b, c)[i % 3]
        doc = np.uint64(src[rng.integers(0, len(src))] & np.uint64(0xFFFFFFFF))
        hs = (src[(src & np.uint64(0xFFFFFFFF)) == doc] >> np.uint64(32)).astype(np.uint32)
        noise = rng.integers(0, 1 << 32, 1000 - len(hs), dtype=np.uint64).astype(np.uint32)
        qs.append(np.concatenate([hs, noise]))
    # some queries aimed at the superseded / deleted docs and at the memory segments
    for doc, src in ((10, a), (6000, a), (6000, b), (12000, b), (77, a), (int(ids_c[5]), c)):
        hs = (src[(src & np.uint64(0xFFFFFFFF)) == np.uint64(doc)] >> np.uint64(32)).astype(np.uint32)
        qs.append(np.concatenate([hs, np.array([1, 2, 3, 9, 8, 7], np.uint32), rng.integers(0, 1 << 32, 1000 - len(hs) - 6, dtype=np.uint64).astype(np.uint32)]))
    got, st = p.check(qs, fpx.SearchOptions(max_results=20, min_score=1, min_score_pct=0))
    assert st.probes >= (1 << 20)
    assert 0 < st.probe_kernel_bytes                                                   # the lean kernel ran
    top = {i: (g[0] if g else None) for i, g in enumerate(got)}
    assert top[nq + 0] is None or top[nq + 0][0] != 10                                # deleted
    assert all(r[0] != 6000 for r in got[nq + 1])                                     # A's copy of 6000 is superseded, B's deleted
    assert got[nq + 4][0][0] == 77 and got[nq + 4][0][1] >= 8                         # overwritten in a memory segment


def test_deferred_list_overflow_falls_back_to_generic_pass(env, monkeypatch):
    """A lean-eligible segment (512-B blocks, > 2^20 items) in which EVERY block holds a 4-byte hash delta: the lean
    kernel defers nearly every probe, its deferred lists overflow, and the batch is rerun on the generic kernel."""
    fpx, oracle, Pair, ctx = env
    if not DIRECT_FORCED:
        monkeypatch.setenv("FPX_DIRECT", "0")           # (the block kernels' test)
    rng = np.random.default_rng(77)
    n_clusters, per_cluster = 1 << 14, 80                       # clusters 2^18 apart: a 4-byte delta every 80 items
    base = (np.arange(n_clusters, dtype=np.uint64) << np.uint64(18))
    h = (base[:, None] + rng.integers(0, 4000, (n_clusters, per_cluster), dtype=np.uint64)).ravel()
    ids = rng.integers(1, 5001, len(h), dtype=np.uint64)
    items = np.sort((h << np.uint64(32)) | ids)
    p = Pair(ctx)
    p.add_file(items, 1, 5000, 1, np.arange(1, 5001))
    p.finish()
    assert len(items) >= (1 << 20)
    nq = 80
    qs = []
    for i in range(nq):
        real = (items[rng.integers(0, len(items), 300)] >> np.uint64(32)).astype(np.uint32)
        noise = (base[rng.integers(0, n_clusters, 700)] + rng.integers(0, 1 << 18, 700, dtype=np.uint64)).astype(np.uint32)
        qs.append(np.concatenate([real, noise]))
    got, st = p.check(qs, fpx.SearchOptions(max_results=20, min_score=1, min_score_pct=0))
    assert st.probes >= (1 << 16)
    assert DIRECT_FORCED or st.generic_iters > st.probes // 4   # the generic per-value decode carried the batch


def test_hit_buffer_overflow_regrows(env):
    """2000 hashes with 600 docs each: one query yields 1.2 M hit records, more than a fresh workspace's hit buffer
    (2^20 records).  The single-query fast path notices at its final check and reruns the general path, which regrows
    the buffer; a batch does the same in its retry loop."""
    fpx, oracle, Pair, _ = env
    ctx = fpx.Context(0)                                      # fresh context = fresh workspace pool with default capacities
    rng = np.random.default_rng(5)
    hot = rng.choice(1 << 32, 2000, replace=False).astype(np.uint64)
    n_docs = 6000
    h = np.stack([rng.choice(hot, 200, replace=False) for _ in range(n_docs)])
    ids = np.arange(1, n_docs + 1, dtype=np.uint64)
    items = np.sort(((h << np.uint64(32)) | ids[:, None]).ravel())
    p = Pair(ctx)
    p.add_file(items, 1, n_docs, 1, ids.astype(np.uint32))
    p.finish()
    q = hot.astype(np.uint32)
    opt = fpx.SearchOptions(max_results=30, min_score=1, min_score_pct=0)
    res = fpx.SearchResults(opt)
    p.reader.search(q, res)                                   # single-query entry point
    want = p.osnap.search(q, 30, 1, 0)
    assert res.getResults() == want and res.stats.hits > (1 << 20)
    ctx2 = fpx.Context(0)
    p2 = Pair(ctx2)
    p2.add_file(items, 1, n_docs, 1, ids.astype(np.uint32))
    p2.finish()
    got, st = p2.check([q, q[:1000], q[500:]], opt)            # batch entry point, fresh pool again
    assert st.hits > (1 << 20)


@pytest.mark.parametrize("pct", [0, 10, 60, 100])
def test_low_floor_multi_pass_with_relative_cutoff(env, pct):
    """The legacy protocol's options (min_score 1, limit up to 500, top_score_percent) on queries with thousands of
    hit docs each: the exact count needs several passes (or the larger table), the best score is found in a count-only
    round and the floor is raised to top * pct / 100 before candidates are emitted -- results must not change."""
    fpx, oracle, Pair, ctx = env
    seed, H, per = 733, 64, 30000
    p = Pair(ctx)
    for s in range(3):
        lo = s * per + 1
        p.add_file(fpx.synth.synth_items(seed, lo, per, H, dist=1), lo, lo + per - 1, s + 1, np.arange(lo, lo + per))
    p.finish()
    hot = [int(fpx.synth.mix64(np.uint64(seed) ^ np.uint64(0x5bd1e9955bd1e995) ^ (np.uint64(k) << np.uint64(32))) >> np.uint64(32))
           for k in range(48)]
    qs = []
    for i in range(40):                                      # every query: a real doc + a varying set of hot values
        doc = 1 + 977 * i
        qs.append(np.concatenate([fpx.synth.synth_hashes(seed, [doc], H, 1)[0], np.array(hot[i % 7::3], np.uint32)]))
    for limit, floor in ((500, 1), (40, 1), (100, 2), (10, 3)):
        got, st = p.check(qs, fpx.SearchOptions(limit, floor, pct))
        assert st.hits > 40 * 3000                            # thousands of hit records per query: several count passes


@pytest.mark.parametrize("block_size", [64, 512, 4096])
def test_small_segments_in_big_batches(env, block_size):
    """Fresh checkpoints: file segments of 10^4 .. 10^5 items next to a big one.  In a big batch they are probed block by
    block from their decoded items (k_probe_small): hot hashes whose runs span up to 4 blocks (the 1000-docs stop), gaps
    between blocks, hashes above every block, re-inserted and deleted docs, duplicate query hashes."""
    fpx, oracle, Pair, ctx = env
    rng = np.random.default_rng(block_size)
    p = Pair(ctx)
    big = fpx.synth.synth_items(900, 1, 9000, 128, dist=1)                       # 1.15 M items: the lean kernel's
    p.add_file(big, 1, 9000, 1, np.arange(1, 9001))
    hot = rng.integers(0, 1 << 32, 6, dtype=np.uint64)
    all_small = []
    for s in range(4):
        n_docs = int(rng.integers(200, 3000))
        ids = np.sort(rng.choice(np.arange(1, 20000), n_docs, replace=False)).astype(np.uint64)     # overlaps the big segment's ids
        h = rng.integers(0, 1 << 31, (n_docs, 32), dtype=np.uint64)              # nothing above 2^31: queries reach past the last block
        m = rng.random(n_docs) < 0.7
        h[m, 0] = hot[rng.integers(0, 6, int(m.sum()))]
        h[:, 1] = h[:, 0]
        alive = (rng.random(n_docs) > 0.05).astype(np.uint8)
        items = np.sort(((h[alive == 1] << np.uint64(32)) | ids[alive == 1][:, None]).ravel())
        p.add_file(items, int(ids.min()), int(ids.max()), 2 + s, ids.astype(np.uint32), alive, block_size=block_size)
        all_small.append(items)
    p.add_memory_changes([("delete", 15), ("insert", 25000, [int(hot[0]), 7])], 9)
    p.finish()
    small_h = np.concatenate([(x >> np.uint64(32)).astype(np.uint32) for x in all_small])
    qs = []
    for i in range(90):
        q = np.concatenate([rng.choice(small_h, 300), hot[: 1 + i % 6].astype(np.uint32),
                            rng.integers(0, 1 << 32, 700, dtype=np.uint64).astype(np.uint32)])
        if i % 4 == 0:
            q = np.concatenate([q, q[:40]])
        qs.append(q)
    got, st = p.check(qs, [fpx.SearchOptions(int(rng.choice([5, 40, 500])), int(rng.choice([1, 2, 10])), int(rng.choice([0, 10, 100]))) for _ in qs])
    assert st.probes >= (1 << 16) and st.scanned_docs > 100000                    # the walks over the hot runs really happened


@pytest.mark.parametrize("nlight", [40, 600])
def test_heavy_queries_counted_in_doc_class_rounds(env, nlight):
    """A query with 100x the batch's average hit records would saturate k_score's counting filter (sized for the average):
    it is counted in rounds over disjoint doc classes instead.  Heavy queries (hot hashes, floors 4..50, with and without
    the relative cut-off) ride in a batch of light ones; results must equal the oracle's."""
    fpx, oracle, Pair, ctx = env
    seed, H, per = 977, 64, 40000
    p = Pair(ctx)
    for s in range(4):
        lo = s * per + 1
        p.add_file(fpx.synth.synth_items(seed, lo, per, H, dist=1), lo, lo + per - 1, s + 1, np.arange(lo, lo + per))
    p.finish()
    hot = np.array([int(fpx.synth.mix64(np.uint64(seed) ^ np.uint64(0x5bd1e9955bd1e995) ^ (np.uint64(k) << np.uint64(32))) >> np.uint64(32))
                    for k in range(96)], dtype=np.uint32)
    qs, opts = [], []
    for i, (floor, pct, limit) in enumerate([(4, 0, 500), (8, 10, 100), (20, 50, 40), (50, 0, 10), (5, 100, 10), (4, 0, 3)]):
        doc = 1 + 4001 * i
        qs.append(np.concatenate([fpx.synth.synth_hashes(seed, [doc], H, 1)[0], hot[i % 3:]]))
        opts.append(fpx.SearchOptions(limit, floor, pct))
    for d in range(1, nlight + 1):
        qs.append(fpx.synth.synth_hashes(seed, [17 * d], H, 1)[0][:12])
        opts.append(fpx.SearchOptions(5, 4, 10))
    got, st = p.check(qs, opts)
    assert st.hits > 400000                               # ~90 k records per heavy query, a few dozen per light one
    assert all(len(g) >= 1 for g in got[:6])
