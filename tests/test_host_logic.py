"""Host-side logic that needs no GPU: synthetic generators (numpy vs the oracle's C), option derivation,
query flattening, segment assignment."""
import numpy as np

from fpx_testlib import fpx, oracle


def test_generators_agree_numpy_vs_c():
    for dist in (0, 1):
        docs = np.array([1, 2, 3, 1000, 123456789, 0xFFFFFFFF], dtype=np.uint64)
        h = fpx.synth.synth_hashes(42, docs, 37, dist)
        for di, d in enumerate(docs):
            for j in (0, 1, 17, 36):
                assert int(h[di, j]) == oracle.synth_hash(42, int(d), j, dist)
        assert np.array_equal(fpx.synth.synth_items(9, 5, 300, 33, dist), oracle.synth_items(9, 5, 300, 33, dist))


def test_hot_pool_exercises_caps():
    items = fpx.synth.synth_items(5, 1, 30000, 96, 1)
    hashes = (items >> np.uint64(32)).astype(np.uint32)
    _, counts = np.unique(hashes, return_counts=True)
    assert counts.max() > 1000            # MAX_DOCS_PER_HASH is reachable
    blocks, index = oracle.build_blocks(items, 1, 512)
    assert (np.diff(index.astype(np.int64)) == 0).sum() >= 4      # one hash spans > 4 blocks


def test_make_queries_shape_and_targets():
    flat, off, targets = fpx.synth.make_queries(7, 99, 16, 5000, 64, query_len=200)
    assert flat.dtype == np.uint32 and len(flat) == 16 * 200 and off[-1] == 16 * 200
    for q in range(16):
        want = fpx.synth.synth_hashes(7, [int(targets[q])], 64)[0]
        got = flat[q * 200:q * 200 + 64]
        same = (got == want).sum()
        assert 40 <= same <= 64           # ~10 % of the hashes carry a flipped bit
        assert all(bin(int(a) ^ int(b)).count("1") <= 1 for a, b in zip(got, want))


def test_option_derivation_matches_reference_defaults():
    # src/api.zig:7-22, src/server.zig:192-193, src/MultiIndex.zig:302-306
    o = fpx.http_options()
    assert (o.max_results, o.min_score, o.min_score_pct) == (40, None, 10)
    assert fpx.http_options(limit=1000).max_results == 100 and fpx.http_options(limit=0).max_results == 1
    c = o.to_c()
    assert (c.max_results, c.has_min_score, c.min_score_pct) == (40, 0, 10)
    for n in (0, 1, 19, 20, 21, 1000):
        assert oracle.lib().orc_default_min_score(n) == (n + 19) // 20


def test_flatten_masks_to_u32_and_keeps_order():
    from acoustid_index_amd.index import _flatten
    flat, off = _flatten([[1, 2, 3], [], [2 ** 32 + 5, 7]])
    assert off.tolist() == [0, 3, 3, 5] and flat.tolist() == [1, 2, 3, 5, 7]


def test_assign_segments_balances():
    own = fpx.sharding.assign_segments([10] * 16, 8)
    assert sorted(np.bincount(own, minlength=8).tolist()) == [2] * 8
    own = fpx.sharding.assign_segments([100, 1, 1, 1, 1, 1, 1, 1], 2)
    load = [sum(w for w, o in zip([100, 1, 1, 1, 1, 1, 1, 1], own) if o == r) for r in range(2)]
    assert max(load) == 100 and min(load) == 7


def test_baseline_config_1_cpu_plumbing():
    """BASELINE.json configs[0]: 10 k fingerprints x 256 u32 hashes, one /_search query on the CPU path
    (here: the oracle, HTTP default options).  The target ranks first with nearly all of its hashes."""
    seed, ndocs, H = 2026, 10000, 256
    items = oracle.synth_items(seed, 1, ndocs, H)
    blocks, index = oracle.build_blocks(items, 1, 512)
    seg = oracle.file_segment(blocks, 512, index, 1, ndocs, 1, np.arange(1, ndocs + 1, dtype=np.uint32))
    snap = oracle.Snapshot([seg], [])
    flat, off, targets = fpx.synth.make_queries(seed, 1, 4, ndocs, H, query_len=1000)
    for q in range(4):
        res, st = snap.search(flat[int(off[q]):int(off[q + 1])], 40, None, 10, with_stats=True)
        assert res and res[0][0] == int(targets[q]) and res[0][1] >= 200
        assert all(s >= 50 for _, s in res)                  # min_score = (1000 + 19) // 20
        assert st.probes >= 990 and st.scanned_blocks <= st.probes + 8
