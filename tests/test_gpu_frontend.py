"""The reference's HTTP-level vectors driven through frontend.py + hostindex.Index on the GPU:
tests/test_fingerprint_api.py:5-260, tests/test_content_negotiation.py:6-161, tests/test_legacy.py:61-91 (transcribed as
data), plus the properties of the host mirror: snapshot isolation, read-your-writes, checkpoint/merge invariance,
the request coalescer, and the stdlib HTTP wrapper."""
import json
import os
import threading

import msgpack
import numpy as np
import pytest

import http_wrapper
from legacy_session import LegacySession

pytestmark = pytest.mark.gpu

J = {"Content-Type": "application/json"}


@pytest.fixture(scope="module")
def env():
    from fpx_testlib import fpx, oracle
    return fpx, oracle, fpx.Context(0)


@pytest.fixture()
def mi(env):
    fpx, _, ctx = env
    m = fpx.MultiIndex(ctx)
    m.create_index("main")
    return m


def search(fpx, mi, query, headers=J, **extra):
    st, ct, body = fpx.frontend.handle_search(mi, "main", headers, json.dumps({"query": query, **extra}).encode())
    assert st == 200, body
    return json.loads(body)["results"]


def update(fpx, mi, changes):
    st, ct, body = fpx.frontend.handle_update(mi, "main", J, json.dumps({"changes": changes}).encode())
    assert st == 200, body
    return json.loads(body)["version"]


def test_insert_single_and_multi(env, mi):
    fpx = env[0]
    fe = fpx.frontend
    st, ct, body = fe.handle_put_fingerprint(mi, "main", 1, J, json.dumps({"hashes": [101, 201, 301]}).encode())
    assert (st, json.loads(body)) == (200, {})
    assert search(fpx, mi, [101, 201, 301]) == [{"id": 1, "score": 3}]
    v = update(fpx, mi, [{"insert": {"id": 1, "hashes": [101, 201, 301]}}, {"insert": {"id": 2, "hashes": [102, 202, 302]}}])
    assert v > 0
    assert search(fpx, mi, [101, 201, 301, 102, 202, 302]) == [{"id": 1, "score": 3}, {"id": 2, "score": 3}]


def test_update_full_and_partial(env, mi):
    fpx = env[0]
    update(fpx, mi, [{"insert": {"id": 1, "hashes": [100, 200, 300]}}])
    update(fpx, mi, [{"insert": {"id": 1, "hashes": [1000, 2000, 3000]}}])
    assert search(fpx, mi, [100, 200, 300]) == []
    assert search(fpx, mi, [1000, 2000, 3000]) == [{"id": 1, "score": 3}]
    update(fpx, mi, [{"insert": {"id": 1, "hashes": [100, 200, 999]}}])
    assert search(fpx, mi, [100, 200, 300]) == [{"id": 1, "score": 2}]
    assert search(fpx, mi, [100, 200, 999]) == [{"id": 1, "score": 3}]


def test_deletes(env, mi):
    fpx = env[0]
    update(fpx, mi, [{"insert": {"id": 1, "hashes": [101, 201, 301]}}, {"insert": {"id": 2, "hashes": [102, 202, 302]}}])
    update(fpx, mi, [{"delete": {"id": 1}}, {"delete": {"id": 2}}])
    assert search(fpx, mi, [101, 201, 301, 102, 202, 302]) == []
    update(fpx, mi, [{"insert": {"id": 3, "hashes": [5, 6]}}])
    st, _, _ = fpx.frontend.handle_delete_fingerprint(mi, "main", 3, J)
    assert st == 200 and search(fpx, mi, [5, 6], min_score=1) == []
    st, _, body = fpx.frontend.handle_update(mi, "main", J, json.dumps({"changes": [{"insert": {"id": 0, "hashes": [1]}}]}).encode())
    assert st == 400 and json.loads(body) == {"error": "InvalidFingerprintId"}


def test_content_negotiation_end_to_end(env, mi):
    fpx = env[0]
    fe = fpx.frontend
    st, ct, body = fe.handle_put_fingerprint(mi, "main", 1, {}, msgpack.packb({"hashes": [101, 201, 301]}))
    assert (st, ct, msgpack.loads(body)) == (200, "application/vnd.msgpack", {})
    st, ct, body = fe.handle_search(mi, "main", {}, msgpack.packb({"query": [101, 201, 301]}))
    assert (st, ct) == (200, "application/vnd.msgpack") and msgpack.loads(body) == {"r": [{"i": 1, "s": 3}]}
    st, ct, body = fe.handle_search(mi, "main", {"Content-Type": "application/json", "Accept": "application/vnd.msgpack"},
                                    json.dumps({"query": [101, 201, 301]}).encode())
    assert msgpack.loads(body) == {"r": [{"i": 1, "s": 3}]}
    st, ct, body = fe.handle_search(mi, "missing", J, json.dumps({"query": [1]}).encode())
    assert st == 404


def test_legacy_ordering_and_limit(env, mi):
    fpx = env[0]
    update(fpx, mi, [{"insert": {"id": 1001, "hashes": [11000, 12000, 13000]}}, {"insert": {"id": 1002, "hashes": [11000, 12000, 19000]}}])
    assert search(fpx, mi, [11000, 12000, 13000], min_score=1) == [{"id": 1001, "score": 3}, {"id": 1002, "score": 2}]
    assert search(fpx, mi, [11000, 12000, 19000], min_score=1) == [{"id": 1002, "score": 3}, {"id": 1001, "score": 2}]
    assert search(fpx, mi, [11000, 12000, 19000], min_score=1, limit=1) == [{"id": 1002, "score": 3}]
    assert search(fpx, mi, [11000, 12000, 19000], min_score=1, score_pct=100) == [{"id": 1002, "score": 3}]


def test_snapshot_isolation_and_read_your_writes(env):
    fpx, _, ctx = env
    ix = fpx.Index(ctx, auto_checkpoint=False)
    ix.update([("insert", 1, [100, 1])])
    old = ix.acquire_reader()
    for i in range(2, 31):
        ix.update([("insert", i, [100, i])])
    opt = fpx.SearchOptions(max_results=100, min_score=1, min_score_pct=10)
    r_old, r_new = fpx.SearchResults(opt), fpx.SearchResults(opt)
    old.search([100], r_old)
    ix.acquire_reader().search([100], r_new)
    assert len(r_old.getResults()) == 1 and len(r_new.getResults()) == 30          # src/Index.zig:1403-1444
    assert ix.version == 30 and len(ix.memory) == 30


def test_checkpoint_and_merge_do_not_change_results(env):
    """the same stream of commits searched (a) as memory segments, (b) after GPU checkpoints, (c) after a GPU file merge,
    and by the oracle over the memory segments: identical results"""
    fpx, oracle, ctx = env
    rng = np.random.default_rng(11)
    ix = fpx.Index(ctx, auto_checkpoint=False)
    orc_mems = []
    commits = []
    for c in range(1, 13):
        changes = []
        for _ in range(40):
            doc = int(rng.integers(1, 300))
            if rng.random() < 0.15:
                changes.append(("delete", doc))
            else:
                changes.append(("insert", doc, rng.integers(0, 1 << 12, 30).tolist()))
        commits.append(changes)
        ix.update(changes)
        orc_mems.append(oracle.memory_segment_from_changes(changes, c))
    osnap = oracle.Snapshot([], orc_mems)
    queries = [rng.integers(0, 1 << 12, 60).tolist() for _ in range(50)]
    opt = fpx.SearchOptions(max_results=20, min_score=1, min_score_pct=0)
    want = [osnap.search(q, 20, 1, 0) for q in queries]

    def got():
        return ix.acquire_reader().search_batch(queries, opt)[0]
    assert got() == want
    # checkpoint the first six commits' worth, then the rest
    with ix._write:
        first = ix._snapshot.merge(ix.memory[:6], 512)
        ix._publish([first], ix.memory[6:])
    assert got() == want
    ix.checkpoint()
    assert len(ix.files) == 2 and not ix.memory and got() == want
    ix.merge_files(0, 2)
    assert len(ix.files) == 1 and got() == want
    assert ix.files[0].commit_id == 1


def test_auto_checkpoint(env):
    fpx, _, ctx = env
    ix = fpx.Index(ctx, max_memory_segments=4)
    for i in range(1, 10):
        ix.update([("insert", i, [1000, i])])
    assert len(ix.files) == 2 and len(ix.memory) == 1
    res = fpx.SearchResults(fpx.SearchOptions(max_results=100, min_score=1, min_score_pct=0))
    ix.acquire_reader().search([1000], res)
    assert [r[0] for r in res.getResults()] == list(range(1, 10))


def test_coalescer_batches_concurrent_searches(env, mi):
    fpx = env[0]
    update(fpx, mi, [{"insert": {"id": i, "hashes": [i * 10 + k for k in range(5)]}} for i in range(1, 65)])
    co = fpx.coalescer.SearchCoalescer(max_batch=64, max_wait_ms=20.0)
    out = {}

    def worker(i):
        st, ct, body = fpx.frontend.handle_search(mi, "main", J, json.dumps({"query": [i * 10 + k for k in range(5)], "min_score": 1}).encode(),
                                                  searcher=co)
        out[i] = (st, json.loads(body))
    ts = [threading.Thread(target=worker, args=(i,)) for i in range(1, 65)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    co.close()
    assert all(out[i] == (200, {"results": [{"id": i, "score": 5}]}) for i in range(1, 65))
    assert co.requests == 64 and co.batches < 64          # searches shared device batches


def test_coalescer_timeout_maps_to_503(env, mi):
    fpx = env[0]
    update(fpx, mi, [{"insert": {"id": 1, "hashes": [1, 2, 3]}}])

    class Slow(fpx.coalescer.SearchCoalescer):
        def search(self, reader, hashes, options, timeout_ms=0):
            raise fpx.SearchTimeout("deadline passed")
    co = Slow()
    st, ct, body = fpx.frontend.handle_search(mi, "main", J, json.dumps({"query": [1, 2, 3], "timeout": 1}).encode(), searcher=co)
    co.close()
    assert st == 503 and json.loads(body) == {"error": "SearchTimeout"}


def test_http_wrapper_round_trip(env):
    fpx, _, ctx = env
    mi = fpx.MultiIndex(ctx)
    srv = http_wrapper.serve(mi, "127.0.0.1", 0)
    port = srv.server_address[1]
    t = threading.Thread(target=srv.serve_forever, daemon=True)
    t.start()
    try:
        def call(method, path, data=None, headers=None):
            import http.client                  # (urllib would add a Content-Type of its own to a bare body)
            c = http.client.HTTPConnection("127.0.0.1", port, timeout=30)
            c.request(method, path, body=data, headers=headers or {})
            r = c.getresponse()
            out = r.status, r.getheader("Content-Type"), r.read()
            c.close()
            return out
        assert call("GET", "/_health")[0] == 200
        assert call("PUT", "/idx1")[0] == 200
        st, ct, body = call("PUT", "/idx1/1", json.dumps({"hashes": [101, 201, 301]}).encode(), J)
        assert (st, json.loads(body)) == (200, {})
        st, ct, body = call("POST", "/idx1/_search", json.dumps({"query": [101, 201, 301]}).encode(), J)
        assert st == 200 and json.loads(body) == {"results": [{"id": 1, "score": 3}]}
        st, ct, body = call("POST", "/idx1/_search", msgpack.packb({"query": [101, 201, 301]}))
        assert (st, ct) == (200, "application/vnd.msgpack") and msgpack.loads(body) == {"r": [{"i": 1, "s": 3}]}
        assert call("PUT", "/idx1/2", b'{"hashes": [1]}', {"Content-Type": "invalid/type"})[0] == 415
        assert call("POST", "/nope/_search", json.dumps({"query": [1]}).encode(), J)[0] == 404
        assert call("DELETE", "/idx1/1", None, J)[0] == 200
        st, ct, body = call("POST", "/idx1/_search", json.dumps({"query": [101, 201, 301]}).encode(), J)
        assert json.loads(body) == {"results": []}
    finally:
        srv.shutdown()


def test_persist_open_and_snapshot_round_trip(env, tmp_path):
    """commits -> GPU checkpoint + file merge -> persist in the reference's segment-file format -> reopen in a new
    Index -> export as a snapshot stream -> restore elsewhere -> reopen: the same answers everywhere (and the oracle's)"""
    import io
    fpx, oracle, ctx = env
    rng = np.random.default_rng(21)
    ix = fpx.Index(ctx, auto_checkpoint=False)
    orc_mems = []
    for c in range(1, 9):
        changes = [("insert", int(rng.integers(1, 200)), rng.integers(0, 1 << 12, 25).tolist()) for _ in range(30)]
        changes += [("delete", int(rng.integers(1, 200))) for _ in range(3)]
        ix.update(changes)
        orc_mems.append(oracle.memory_segment_from_changes(changes, c))
        if c == 3 or c == 6:
            ix.checkpoint()
    osnap = oracle.Snapshot([], orc_mems)
    queries = [rng.integers(0, 1 << 12, 50).tolist() for _ in range(30)]
    opt = fpx.SearchOptions(max_results=20, min_score=1, min_score_pct=0)
    want = [osnap.search(q, 20, 1, 0) for q in queries]
    assert ix.acquire_reader().search_batch(queries, opt)[0] == want
    d1 = str(tmp_path / "data")
    infos = ix.persist(d1)                                     # checkpoints commits 7..8 too
    assert [i[:2] for i in infos] == [(1, 2), (4, 2), (7, 1)]   # commit intervals [1,3] [4,6] [7,8] (src/segment.zig:38-51)
    assert sorted(os.listdir(d1)) == sorted(["manifest"] + [fpx.segfile.segment_file_name(i[0], i[1]) for i in infos])
    ix2 = fpx.Index.open(ctx, d1)
    assert ix2.version == 8 and len(ix2.files) == 3
    assert ix2.acquire_reader().search_batch(queries, opt)[0] == want
    ix2.merge_files(0, 3)
    assert ix2.files[0].merges == 7 and ix2.acquire_reader().search_batch(queries, opt)[0] == want
    buf = io.BytesIO()
    ix2.export_snapshot(buf, str(tmp_path / "data2"), generation=5)
    d3 = str(tmp_path / "restored")
    fpx.segfile.restore_snapshot(d3, buf.getvalue(), 5)
    ix3 = fpx.Index.open(ctx, d3)
    assert ix3.version == 8 and len(ix3.files) == 1
    assert ix3.acquire_reader().search_batch(queries, opt)[0] == want
    ix3.update([("insert", 5000, [1, 2, 3])])                  # the reopened index keeps committing where it left off
    assert ix3.version == 9


def test_legacy_protocol_search(env):
    """tests/test_legacy.py:61-91 through the GPU path"""
    fpx, _, ctx = env
    mi = fpx.MultiIndex(ctx)
    s = LegacySession(mi)
    assert s.cmd("begin") == "OK "
    assert s.cmd("insert 1001 11000,12000,13000") == "OK "
    assert s.cmd("insert 1002 11000,12000,19000") == "OK "
    assert s.cmd("commit") == "OK "
    assert s.cmd("search 11000,12000,13000") == "OK 1001:3 1002:2"
    assert s.cmd("search 11000,12000,19000") == "OK 1002:3 1001:2"
    assert s.cmd("begin") == "OK " and s.cmd("insert 6001 61000,62000,63000") == "OK " and s.cmd("rollback") == "OK "
    assert s.cmd("search 61000,62000,63000") == "OK "
    s.cmd("begin"); s.cmd("insert 2001 21000,22000"); s.cmd("insert 2002 21000,22000"); s.cmd("commit")
    both = s.cmd("search 21000,22000")
    assert both.startswith("OK ") and len(both[3:].split()) == 2
    assert s.cmd("set max_results 1") == "OK " and len(s.cmd("search 21000,22000")[3:].split()) == 1
    s2 = LegacySession(mi)                                       # another connection sees the commits
    assert s2.cmd("search -4294956296,12000") == "OK 1001:2 1002:2"          # 11000 sent in its signed-wrapped form
    s.cmd("begin"); s.cmd("set attribute foo bar"); s.cmd("commit")
    assert s2.cmd("get attribute foo") == "OK bar"


def test_readers_survive_concurrent_publishes_and_merges(env):
    """The reference's lifetime contract (src/Index.zig:1-6): a reader keeps the snapshot it acquired -- and through it
    every segment -- alive while a writer publishes new snapshots, checkpoints and merges (which drop the last owning
    reference to retired segments).  Readers hammer the index during 150 commits, 12 GPU checkpoints and 3 file merges;
    every doc, once visible to a thread, stays visible with its full score (read-your-writes is monotonic)."""
    fpx, _, ctx = env
    ix = fpx.Index(ctx, auto_checkpoint=False)
    n_commits = 150
    stop = threading.Event()
    errors = []

    def hashes(k):
        return [k * 16 + j for j in range(5)]

    def reader(tid):
        seen = set()
        rng = np.random.default_rng(tid)
        opt = fpx.SearchOptions(max_results=5, min_score=1, min_score_pct=0)
        try:
            while not stop.is_set():
                k = int(rng.integers(1, n_commits + 1))
                rd = ix.acquire_reader()
                res = fpx.SearchResults(opt)
                rd.search(hashes(k), res)
                got = res.getResults()
                if got:
                    assert got == [(k, 5)], got
                    seen.add(k)
                else:
                    assert k not in seen, f"doc {k} disappeared"
                if rng.random() < 0.2:                       # batched entry point on the same held snapshot
                    ks = [int(x) for x in rng.integers(1, n_commits + 1, 8)]
                    out, _ = rd.search_batch([hashes(x) for x in ks], opt)
                    for x, r in zip(ks, out):
                        assert r in ([], [(x, 5)])
                        if x in seen:
                            assert r == [(x, 5)]
        except Exception as e:                               # surfaces in the main thread
            errors.append(repr(e))

    threads = [threading.Thread(target=reader, args=(t,)) for t in range(6)]
    [t.start() for t in threads]
    try:
        for k in range(1, n_commits + 1):
            ix.update([("insert", k, hashes(k))])
            if k % 12 == 0:
                ix.checkpoint()
            if k % 48 == 0 and len(ix.files) >= 2:
                ix.merge_files(0, len(ix.files))
    finally:
        stop.set()
        [t.join() for t in threads]
    assert not errors, errors[:3]
    assert len(ix.files) <= 3 and ix.version == n_commits
    res = fpx.SearchResults(fpx.SearchOptions(max_results=5, min_score=1, min_score_pct=0))
    ix.acquire_reader().search(hashes(n_commits), res)
    assert res.getResults() == [(n_commits, 5)]
