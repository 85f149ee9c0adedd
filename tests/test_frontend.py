"""/_search front end (frontend.py) without a GPU: content negotiation, request decoding, clamps, status mapping and
response framing, against the behaviour the reference pins in src/server.zig:84-196, src/api.zig:7-72 and
tests/test_content_negotiation.py (vectors transcribed as data)."""
import json

import msgpack
import pytest

from fpx_testlib import fpx
from legacy_session import LegacySession

fe = fpx.frontend


class FakeIndex:
    loading = False


class FakeMulti:
    def __init__(self):
        self.idx = FakeIndex()

    def get_index(self, name):
        if name != "main":
            raise fpx.hostindex.IndexNotFound(name)
        return self.idx


class Recorder:
    """stands in for the device: records what the handler asks for"""

    def __init__(self, results=((1, 3),), exc=None):
        self.results, self.exc, self.calls = list(results), exc, []

    def __call__(self, index, hashes, options, timeout_ms):
        self.calls.append((list(hashes), options, timeout_ms))
        if self.exc:
            raise self.exc
        return self.results


def test_request_type_rules():
    # src/server.zig:84-95
    assert fe.request_type({"content-type": "application/json"}, b"{}") == fe.JSON
    assert fe.request_type({"content-type": "application/vnd.msgpack"}, b"\x80") == fe.MSGPACK
    assert fe.request_type({"content-type": "application/json; charset=utf-8"}, b"{}") == fe.JSON
    assert fe.request_type({}, b"\x80") == fe.MSGPACK          # no header + body -> msgpack
    assert fe.request_type({}, b"") == fe.JSON                 # no header, no body -> JSON
    with pytest.raises(fe.UnsupportedMediaType):
        fe.request_type({"content-type": "invalid/type"}, b"x")


def test_response_type_rules():
    # src/server.zig:97-105
    assert fe.response_type({"accept": fe.MSGPACK, "content-type": fe.JSON}, b"{}") == fe.MSGPACK
    assert fe.response_type({"accept": "text/html", "content-type": fe.JSON}, b"{}") == fe.JSON
    assert fe.response_type({"accept": "*/*"}, b"\x80") == fe.MSGPACK
    assert fe.response_type({"content-type": "invalid/type"}, b"x") == fe.JSON


def test_msgpack_default_no_headers():
    # tests/test_content_negotiation.py:6-34: full names posted as msgpack, one-letter keys come back
    rec = Recorder([(1, 3)])
    st, ct, body = fe.handle_search(FakeMulti(), "main", {}, msgpack.packb({"query": [101, 201, 301]}), rec)
    assert (st, ct) == (200, "application/vnd.msgpack")
    assert msgpack.loads(body) == {"r": [{"i": 1, "s": 3}]}
    assert rec.calls[0][0] == [101, 201, 301]
    # the one-letter request keys of src/api.zig:24-26
    st, ct, body = fe.handle_search(FakeMulti(), "main", {}, msgpack.packb({"q": [7], "l": 2}), rec)
    assert st == 200 and rec.calls[1][0] == [7] and rec.calls[1][1].max_results == 2


def test_json_and_mixed_formats():
    # tests/test_content_negotiation.py:37-61, :79-112
    rec = Recorder([(1, 3)])
    body = json.dumps({"query": [101, 201, 301]}).encode()
    st, ct, out = fe.handle_search(FakeMulti(), "main", {"Content-Type": "application/json"}, body, rec)
    assert (st, ct) == (200, "application/json") and json.loads(out) == {"results": [{"id": 1, "score": 3}]}
    st, ct, out = fe.handle_search(FakeMulti(), "main", {"Content-Type": "application/json", "Accept": "application/vnd.msgpack"}, body, rec)
    assert (st, ct) == (200, "application/vnd.msgpack") and msgpack.loads(out) == {"r": [{"i": 1, "s": 3}]}
    st, ct, out = fe.handle_search(FakeMulti(), "main", {"Content-Type": "application/vnd.msgpack", "Accept": "application/json"},
                                   msgpack.packb({"q": [5]}), rec)
    assert (st, ct) == (200, "application/json") and json.loads(out) == {"results": [{"id": 1, "score": 3}]}


def test_defaults_and_clamps():
    # src/api.zig:7-22, src/server.zig:189-193, src/MultiIndex.zig:302-306
    rec = Recorder([])
    H = {"Content-Type": "application/json"}
    fe.handle_search(FakeMulti(), "main", H, json.dumps({"query": [1, 2, 3]}).encode(), rec)
    q, o, t = rec.calls[-1]
    assert (o.max_results, o.min_score, o.min_score_pct, t) == (40, None, 10, 500)
    fe.handle_search(FakeMulti(), "main", H, json.dumps({"query": [1], "limit": 0, "timeout": 999999}).encode(), rec)
    q, o, t = rec.calls[-1]
    assert (o.max_results, t) == (1, 10000)
    fe.handle_search(FakeMulti(), "main", H, json.dumps({"query": [1], "limit": 5000, "timeout": 0, "min_score": 7, "score_pct": 50}).encode(), rec)
    q, o, t = rec.calls[-1]
    assert (o.max_results, o.min_score, o.min_score_pct, t) == (100, 7, 50, 0)
    # msgpack short keys
    fe.handle_search(FakeMulti(), "main", {}, msgpack.packb({"q": [9], "l": 3, "t": 20, "m": 2, "s": 0}), rec)
    q, o, t = rec.calls[-1]
    assert (q, o.max_results, o.min_score, o.min_score_pct, t) == ([9], 3, 2, 0, 20)


@pytest.mark.parametrize("body", [
    b"invalid json data", b"", b"[]", b'{"limit": 3}', b'{"query": "x"}', b'{"query": [-1]}', b'{"query": [4294967296]}',
    b'{"query": [1], "bogus": 1}', b'{"query": [1.5]}', b'{"query": [1], "limit": "3"}',
])
def test_bad_request_json(body):
    rec = Recorder()
    st, ct, out = fe.handle_search(FakeMulti(), "main", {"Content-Type": "application/json"}, body, rec)
    assert st == 400 and ct == "application/json" and json.loads(out) == {"error": "BadRequest"}
    assert not rec.calls


def test_error_formats_and_statuses():
    # tests/test_content_negotiation.py:115-161, src/server.zig:110-125
    rec = Recorder()
    st, ct, out = fe.handle_search(FakeMulti(), "main", {"Content-Type": "application/vnd.msgpack"}, b"invalid msgpack data", rec)
    assert (st, ct) == (400, "application/vnd.msgpack") and msgpack.loads(out) == {"e": "BadRequest"}
    st, ct, out = fe.handle_search(FakeMulti(), "main", {}, b"invalid data", rec)
    assert (st, ct) == (400, "application/vnd.msgpack")
    st, ct, out = fe.handle_search(FakeMulti(), "main", {"Content-Type": "invalid/type"}, b'{"query": [1]}', rec)
    assert st == 415 and json.loads(out) == {"error": "UnsupportedMediaType"}
    st, ct, out = fe.handle_search(FakeMulti(), "nope", {"Content-Type": "application/json"}, b'{"query": [1]}', rec)
    assert st == 404 and json.loads(out) == {"error": "IndexNotFound"}
    st, ct, out = fe.handle_search(FakeMulti(), "main", {"Content-Type": "application/json"}, b'{"query": [1]}',
                                   Recorder(exc=fpx.SearchTimeout("t")))
    assert st == 503 and json.loads(out) == {"error": "SearchTimeout"}
    m = FakeMulti()
    m.idx.loading = True
    st, ct, out = fe.handle_search(m, "main", {"Content-Type": "application/json"}, b'{"query": [1]}', rec)
    assert st == 503 and json.loads(out) == {"error": "IndexNotReady"}
    st, ct, out = fe.handle_search(FakeMulti(), "main", {"Content-Type": "application/json"}, b'{"query": [1]}',
                                   Recorder(exc=RuntimeError("boom")))
    assert st == 500


def test_parse_changes():
    ch, ev = fe.parse_changes({"changes": [{"insert": {"id": 1, "hashes": [1, 2]}}, {"delete": {"id": 2}}], "expected_version": 4}, False)
    assert ch == [("insert", 1, [1, 2]), ("delete", 2)] and ev == 4
    ch, ev = fe.parse_changes({"c": [{"i": {"i": 7, "h": [9]}}, {"d": {"i": 8}}]}, True)
    assert ch == [("insert", 7, [9]), ("delete", 8)] and ev is None
    ch, ev = fe.parse_changes({"changes": [{"insert": {"id": 7, "hashes": [9]}}]}, True)     # msgpack with full names
    assert ch == [("insert", 7, [9])]
    with pytest.raises(fe.BadRequest):
        fe.parse_changes({"changes": [{"upsert": {}}]}, False)


# ---- legacy line protocol (src/legacy.zig), vectors of tests/test_legacy.py that need no device -------------------------
class _NoIndex:
    legacy_attrs = {}

    def get_index(self, name):
        raise fpx.hostindex.IndexNotFound(name)

    def create_index(self, name):
        raise AssertionError("no commit expected")


def test_legacy_protocol_without_device():
    s = LegacySession(_NoIndex())
    assert s.cmd("echo hello world") == "OK hello world"                    # tests/test_legacy.py:30-37
    assert s.cmd("") == "OK "
    assert s.cmd("frobnicate x").startswith("ERR ")
    assert s.cmd("search notanumber") == "ERR invalid fingerprint"
    assert s.cmd("search") == "ERR expected one argument"
    assert s.cmd("insert 1 1,2,3") == "ERR not in transaction"
    assert s.cmd("begin") == "OK " and s.cmd("begin") == "ERR already in transaction"
    assert s.cmd("insert x 1,2") == "ERR invalid document id"
    assert s.cmd("insert 5 1,,2") == "ERR invalid fingerprint"
    assert s.cmd("rollback") == "OK " and s.cmd("rollback") == "ERR not in transaction"
    assert s.cmd("get max_results") == "OK 500" and s.cmd("set max_results 1") == "OK " and s.cmd("get max_results") == "OK 1"
    assert s.cmd("set max_results x") == "ERR invalid value"
    assert s.cmd("set attribute foo bar") == "ERR not in transaction"
    assert s.cmd("get attribute foo") == "OK "
    assert s.cmd("optimize") == "ERR not in transaction"
    assert LegacySession(_NoIndex(), read_only=True).cmd("begin") == "ERR read-only replica"
    # signed decimals are reinterpreted as u32 (src/legacy.zig:318-330)
    assert LegacySession.parse_fingerprint("-1,2147483648,-2147483648") == [0xFFFFFFFF, 0x80000000, 0x80000000]
    assert s.cmd("search 1,2,3") == "OK "                                    # nothing committed
