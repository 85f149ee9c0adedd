"""The behaviours NO reference test pins: the MAX_BLOCKS_PER_HASH / MAX_DOCS_PER_HASH truncation, hashes
spanning several blocks, block sizes other than 512, partial last quads, the relative cut-off with ties, tombstone
supersession combined with top-k truncation.  The C oracle is cross-checked here against an independent brute force
written from the spec in SURVEY.md appendix B, working on the RAW postings (sorted item arrays + where the blocks start),
never on encoded bytes -- so the two share nothing but the block boundaries."""
import numpy as np
import pytest

from fpx_testlib import fpx, oracle

MAX_BLOCKS_PER_HASH, MAX_DOCS_PER_HASH = 4, 1000        # src/FileSegment.zig:25-26


class RawSegment:
    def __init__(self, items, block_starts, commit_id, doc_ids, alive, min_doc, max_doc, is_file=True):
        self.h = (items >> np.uint64(32)).astype(np.int64)
        self.d = (items & np.uint64(0xFFFFFFFF)).astype(np.int64)
        self.starts = list(block_starts) + [len(items)]
        self.commit, self.is_file = commit_id, is_file
        self.docs = {int(i): bool(a) for i, a in zip(doc_ids, alive)}
        self.min_doc, self.max_doc = min_doc, max_doc

    def scan(self, h, incr):
        """-> (visited blocks, scanned docs) for one hash"""
        if not self.is_file:
            lo, hi = np.searchsorted(self.h, h, "left"), np.searchsorted(self.h, h, "right")
            for d in self.d[lo:hi]:
                incr(int(d), self.commit)
            return 0, 0
        nblocks = len(self.starts) - 1
        b = 0
        while b < nblocks and self.h[self.starts[b + 1] - 1] < h:       # first block whose max hash >= h
            b += 1
        nb = nd = 0
        while b < nblocks:
            s, e = self.starts[b], self.starts[b + 1]
            if self.h[s] > h:
                break
            m = [int(self.d[i]) for i in range(s, e) if self.h[i] == h]
            for d in m:
                incr(d, self.commit)
            nb += 1
            nd += len(m)
            if nb >= MAX_BLOCKS_PER_HASH or nd > MAX_DOCS_PER_HASH:
                break
            b += 1
        return nb, nd


def brute_search(segments, query, max_results, min_score, pct):
    hits = {}

    def incr(doc, commit):
        c, s = hits.get(doc, (-1, 0))
        if c < commit:
            hits[doc] = (commit, 1)
        elif c == commit:
            hits[doc] = (c, s + 1)

    blocks = docs = 0
    uniq = sorted(set(int(x) for x in query))
    for seg in segments:
        for h in uniq:
            nb, nd = seg.scan(h, incr)
            blocks += nb
            docs += nd
    if min_score is None:
        min_score = (len(query) + 19) // 20
    cand = sorted(((s, d, c) for d, (c, s) in hits.items() if s >= min_score), key=lambda t: (-t[0], t[1]))
    out = []
    for s, d, c in cand:
        if len(out) == max_results:
            break
        if any(g.commit > c and g.min_doc <= d <= g.max_doc and d in g.docs for g in segments):
            continue
        if s < min_score:
            break
        if not out:
            min_score = max(min_score, min(s * pct // 100, 0xFFFFFFFF))
        out.append((d, s))
    return out, blocks, docs


def block_starts_of(blocks, block_size, nblocks):
    n = [int(blocks[b * block_size + 4]) | (int(blocks[b * block_size + 5]) << 8) for b in range(nblocks)]
    return np.concatenate([[0], np.cumsum(n)[:-1]]).astype(np.int64).tolist() if nblocks else []


def make_world(block_size, seed):
    rng = np.random.default_rng(seed)
    raws, ofile, omem = [], [], []
    H = 24
    hot = rng.integers(0, 1 << 32, 6, dtype=np.uint64)              # hot hashes: > 1000 docs, > 4 blocks
    commit = 0
    for s in range(3):
        commit += 1
        lo = s * 2500 + 1
        ids = np.arange(lo, lo + 2500 + (3 if s == 1 else 0), dtype=np.uint64)   # s == 1: item count not a multiple of 4
        if s:
            ids = np.sort(np.concatenate([ids, rng.choice(np.arange(1, lo), 120, replace=False).astype(np.uint64)]))
        h = rng.integers(0, 1 << 32, (len(ids), H), dtype=np.uint64)
        h[:, 0] = np.where(rng.random(len(ids)) < 0.6, hot[rng.integers(0, 6, len(ids))], h[:, 0])
        h[:, 1] = h[:, 0]                                            # duplicate postings inside a doc
        items = np.sort(((h << np.uint64(32)) | ids[:, None]).ravel())
        if s == 1:
            items = items[:-1]
        alive = np.ones(len(ids), np.uint8)
        blocks, index = oracle.build_blocks(items, int(ids.min()), block_size)
        ofile.append(oracle.file_segment(blocks, block_size, index, int(ids.min()), int(ids.max()), commit, ids.astype(np.uint32), alive))
        raws.append(RawSegment(items, block_starts_of(blocks, block_size, len(index)), commit, ids, alive, int(ids.min()), int(ids.max())))
    # a memory segment with tombstones for popular docs and an overwrite
    commit += 1
    changes = [("delete", 7), ("delete", 2600), ("insert", 11, [int(hot[0]), int(hot[1]), 5]), ("delete", 5100)]
    m = oracle.memory_segment_from_changes(changes, commit)
    omem.append(m)
    mids, malive = m.docs()
    raws.append(RawSegment(m.items(), [], commit, mids, malive, m.min_doc_id, m.max_doc_id, is_file=False))
    return raws, oracle.Snapshot(ofile, omem), hot, rng


@pytest.mark.parametrize("block_size", [64, 512, 4096])
def test_oracle_equals_brute_force_on_unpinned_behaviour(block_size):
    raws, snap, hot, rng = make_world(block_size, 1000 + block_size)
    all_h = np.concatenate([r.h for r in raws[:3]])
    queries = []
    for i in range(40):
        q = rng.choice(all_h, 60).tolist() + [int(hot[i % 6]), int(hot[(i + 1) % 6])] + rng.integers(0, 1 << 32, 10).tolist()
        if i % 5 == 0:
            q += q[:7]                                               # duplicate query hashes
        queries.append(q)
    hit_caps = False
    for q in queries:
        for (mr, ms, pct) in ((10, 1, 0), (40, None, 10), (3, 2, 100), (500, 1, 10), (5, 1, 50), (10, 1, 150), (10, 1, 0xFFFFFFFF)):
            want, wb, wd = brute_search(raws, q, mr, ms, pct)
            got, st = snap.search(q, mr, ms, pct, with_stats=True)
            assert got == want
            assert (st.scanned_blocks, st.scanned_docs) == (wb, wd)
        hit_caps = hit_caps or wd > MAX_DOCS_PER_HASH
    assert hit_caps                                                  # the truncation rules were really exercised


def test_brute_force_reproduces_reference_vectors():
    """sanity of the brute force itself on vectors the reference DOES pin (tests/golden/search_kat.json)"""
    import json
    import os
    sc = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "search_kat.json")))["scenarios"]
    for s in sc:
        raws = []
        for sd in s["segments"]:
            m = oracle.memory_segment_from_changes([tuple(c) for c in sd["changes"]], sd["commit_id"])
            ids, alive = m.docs()
            items = m.items()
            if sd["kind"] == "file":
                blocks, index = oracle.build_blocks(items, m.min_doc_id, sd["block_size"])
                raws.append(RawSegment(items, block_starts_of(blocks, sd["block_size"], len(index)), sd["commit_id"], ids, alive,
                                       m.min_doc_id, m.max_doc_id))
            else:
                raws.append(RawSegment(items, [], sd["commit_id"], ids, alive, m.min_doc_id, m.max_doc_id, is_file=False))
        for chk in s["checks"]:
            mr, ms, pct = (40, None, 10) if chk.get("http") else (chk["max_results"], chk["min_score"], chk["min_score_pct"])
            got, _, _ = brute_search(raws, chk["query"], mr, ms, pct)
            assert [list(r) for r in got] == chk["expect"], s["name"]


def test_search_many_equals_single_searches(orc):
    """orc_search_many (pthread workers with recycled collectors: bench.py's CPU baseline) returns exactly what
    orc_search returns query by query, for any thread count, with and without the default floor."""
    import numpy as np
    seed, ndocs, H = 31, 3000, 48
    items = orc.synth_items(seed, 1, ndocs, H, dist=1)
    half = ndocs // 2
    a = items[(items & 0xFFFFFFFF) <= half]
    b = items[(items & 0xFFFFFFFF) > half]
    segs = []
    for part, lo, hi, cid in ((a, 1, half, 1), (b, half + 1, ndocs, 2)):
        blocks, index = orc.build_blocks(part, lo, 512)
        segs.append(orc.file_segment(blocks, 512, index, lo, hi, cid, np.arange(lo, hi + 1, dtype=np.uint32)))
    snap = orc.Snapshot(segs, [])
    rng = np.random.default_rng(5)
    queries = []
    for i in range(40):
        d = int(rng.integers(1, ndocs + 1))
        own = np.array([orc.synth_hash(seed, d, j, 1) for j in range(H)], np.uint32)
        queries.append(np.concatenate([own, rng.integers(0, 2**32, int(rng.integers(0, 60)), dtype=np.uint64).astype(np.uint32)]))
    queries.append(np.zeros(0, np.uint32))
    lens = np.array([len(q) for q in queries], np.uint64)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    flat = np.concatenate(queries).astype(np.uint32)
    for (mr, ms, pct) in ((40, None, 10), (5, 1, 50), (500, 1, 0)):
        want = [snap.search(q, mr, ms, pct) for q in queries]
        for nthreads, secs in ((1, 0.0), (7, 0.0), (16, 0.05)):
            out, out_n, rep = snap.search_many(flat, offsets, mr, ms, pct, nthreads=nthreads, min_seconds=secs)
            got = [[(int(out[q, i, 0]), int(out[q, i, 1])) for i in range(int(out_n[q]))] for q in range(len(queries))]
            assert got == want
            assert rep["queries_done"] >= len(queries) and len(rep["latency_ms"]) >= len(queries)
            assert rep["wall_s"] >= secs
