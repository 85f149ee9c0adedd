"""`POST /:index/_search` front end: request decoding, content negotiation, option clamps, status
mapping and response encoding of the reference, above the C ABI.

    src/server.zig:84-142   requestType / responseType / decodeAs / sendError / respond
    src/server.zig:189-196  handleSearch (limit and timeout clamps)
    src/api.zig:7-27        SearchRequest defaults and msgpack keys q,t,l,m,s
    src/api.zig:57-72       SearchResult / SearchResponse (msgpack keys i,s / r)
    src/MultiIndex.zig:287-330  search(): option derivation, timeout -> SearchTimeout

Wire formats: JSON uses the full field names.  MessagePack maps are keyed by the FIRST LETTER of each field
(`field_name_prefix = 1`): responses are written with one-letter keys, and a request key is matched on its first letter,
which is how the reference's own test can post {'query': ...} as msgpack and read back {'r': [{'i':..,'s':..}]}
(tests/test_content_negotiation.py:6-34).  The msgpack library the reference links (msgpack.zig@bef6671) is not vendored,
so only what that test file pins is claimed for the msgpack framing; the media type is `application/vnd.msgpack`.
Networking is the host's business: `handle_search` maps (headers, body) to (status, content type, body); tests/http_wrapper.py
wraps it in a stdlib HTTP server for demonstrations and for replaying the reference's HTTP tests.
"""
import json
from dataclasses import dataclass, field
from typing import List, Optional

import msgpack

from ._lib import SearchTimeout
from . import index as _ix
from .hostindex import IndexNotFound, InvalidFingerprintId, VersionMismatch

JSON = "application/json"
MSGPACK = "application/vnd.msgpack"

# src/api.zig:7-11
DEFAULT_SEARCH_TIMEOUT = 500
MAX_SEARCH_TIMEOUT = 10000
DEFAULT_SEARCH_LIMIT = 40
MIN_SEARCH_LIMIT = 1
MAX_SEARCH_LIMIT = 100

U32_MAX = 0xFFFFFFFF


class BadRequest(Exception):
    pass


class UnsupportedMediaType(Exception):
    pass


class IndexNotReady(Exception):
    pass


# error name -> HTTP status (src/server.zig:110-125)
STATUS = {
    "BadRequest": 400, "InvalidIndexName": 400, "InvalidFingerprintId": 400,
    "IndexNotFound": 404, "FingerprintNotFound": 404,
    "IndexNotReady": 503, "SearchTimeout": 503,
    "VersionMismatch": 409,
    "UnsupportedMediaType": 415,
    "NotImplemented": 501,
}


def error_name(exc):
    if isinstance(exc, SearchTimeout):
        return "SearchTimeout"
    if isinstance(exc, (BadRequest, UnsupportedMediaType, IndexNotReady, IndexNotFound, InvalidFingerprintId, VersionMismatch)):
        return type(exc).__name__
    if isinstance(exc, MemoryError):
        return "OutOfMemory"
    return "Unexpected"


def media_type(value):
    """http.ContentType.fromContentType: the media type without parameters, lower-cased"""
    if value is None:
        return None
    return value.split(";")[0].strip().lower()


def request_type(headers, body):
    """src/server.zig:84-95: an explicit Content-Type wins; explicit but unsupported -> 415; no header -> msgpack when
    there is a body, JSON otherwise."""
    ct = media_type(headers.get("content-type"))
    if ct is not None:
        if ct in (JSON, MSGPACK):
            return ct
        raise UnsupportedMediaType(ct)
    return MSGPACK if body else JSON


def response_type(headers, body):
    """src/server.zig:97-105: an explicit, supported Accept wins, else mirror the request type (JSON if that fails)."""
    acc = media_type(headers.get("accept"))
    if acc in (JSON, MSGPACK):
        return acc
    try:
        return request_type(headers, body)
    except UnsupportedMediaType:
        return JSON


def _fields(obj, fields, short_keys):
    """field name -> value.  JSON: exact names, unknown fields are errors (std.json default).  msgpack: a key selects
    the field that shares its first letter."""
    names = {(f[0] if short_keys else f): f for f in fields}
    out = {}
    for k, v in obj.items():
        if isinstance(k, bytes):
            k = k.decode("utf-8", "replace")
        if not isinstance(k, str):
            raise BadRequest("map keys must be strings")
        kk = k[:1] if short_keys else k
        if kk not in names:
            raise BadRequest(f"unknown field {k!r}")
        out[names[kk]] = v
    return out


def _u32(v, what):
    if isinstance(v, bool) or not isinstance(v, int) or v < 0 or v > U32_MAX:
        raise BadRequest(f"{what} must be an unsigned 32-bit integer")
    return v


@dataclass
class SearchRequest:            # src/api.zig:14-27
    query: List[int]
    timeout: int = DEFAULT_SEARCH_TIMEOUT
    limit: int = DEFAULT_SEARCH_LIMIT
    min_score: Optional[int] = None
    score_pct: int = 10

    FIELDS = ("query", "timeout", "limit", "min_score", "score_pct")

    @classmethod
    def from_obj(cls, obj, short_keys):
        if not isinstance(obj, dict):
            raise BadRequest("request body must be a map")
        kw = _fields(obj, cls.FIELDS, short_keys)
        if "query" not in kw:
            raise BadRequest("missing field 'query'")
        q = kw["query"]
        if not isinstance(q, (list, tuple)):
            raise BadRequest("'query' must be an array")
        kw["query"] = [_u32(x, "query hash") for x in q]
        for f in ("timeout", "limit", "score_pct"):
            if f in kw:
                kw[f] = _u32(kw[f], f)
        if kw.get("min_score") is not None:
            kw["min_score"] = _u32(kw["min_score"], "min_score")
        return cls(**kw)

    def sanitize(self):
        """handleSearch (src/server.zig:189-193): clamp the untrusted limit and timeout"""
        self.limit = max(min(self.limit, MAX_SEARCH_LIMIT), MIN_SEARCH_LIMIT)
        self.timeout = min(self.timeout, MAX_SEARCH_TIMEOUT)
        return self

    def options(self):
        """src/MultiIndex.zig:302-306"""
        return _ix.SearchOptions(max_results=self.limit, min_score=self.min_score, min_score_pct=self.score_pct)


def decode_body(ctype, body):
    """decodeAs (src/server.zig:107-113); any decode failure is a BadRequest (requireBody :150-154)"""
    if not body:
        raise BadRequest("missing body")
    try:
        if ctype == JSON:
            return json.loads(body), False
        return msgpack.unpackb(body, raw=False, strict_map_key=False), True
    except Exception as e:              # malformed JSON / msgpack
        raise BadRequest(str(e)) from None


def encode(ctype, obj_json, obj_msgpack):
    if ctype == JSON:
        return json.dumps(obj_json, separators=(",", ":")).encode()
    return msgpack.packb(obj_msgpack)


def error_response(headers, body, exc):
    name = error_name(exc)
    status = STATUS.get(name, 500)
    ctype = response_type(headers, body)
    return status, ctype, encode(ctype, {"error": name}, {"e": name})


def _lower(headers):
    return {str(k).lower(): v for k, v in (headers or {}).items()}


def handle_search(multi_index, index_name, headers, body, searcher=None):
    """POST /:index/_search -> (status, content_type, body bytes).

    `searcher(index, hashes, options, timeout_ms) -> [(id, score)]` lets a host route the search through a request
    coalescer (coalescer.SearchCoalescer); the default runs one fpx_search on the index's current snapshot."""
    headers = _lower(headers)
    try:
        ctype = request_type(headers, body)
        obj, short = decode_body(ctype, body)
        req = SearchRequest.from_obj(obj, short).sanitize()
        index = multi_index.get_index(index_name)
        if getattr(index, "loading", False):
            raise IndexNotReady(index_name)
        if searcher is not None:
            results = searcher(index, req.query, req.options(), req.timeout)
        else:
            res = _ix.SearchResults(req.options())
            index.acquire_reader().search(req.query, res, timeout_ms=req.timeout)
            results = res.getResults()
        rtype = response_type(headers, body)
        return 200, rtype, encode(rtype,
                                  {"results": [{"id": int(i), "score": int(s)} for i, s in results]},
                                  {"r": [{"i": int(i), "s": int(s)} for i, s in results]})
    except Exception as e:
        return error_response(headers, body, e)


# ---- the minimum of `_update` needed to drive the search path from the reference's HTTP tests --------------------
def parse_changes(obj, short):
    """UpdateRequest (src/api.zig:29-37) and Change (src/change.zig): changes = [{insert:{id,hashes}} | {delete:{id}} |
    {set_metadata:{entries}}], optional metadata, optional expected_version."""
    if not isinstance(obj, dict):
        raise BadRequest("request body must be a map")
    req = _fields(obj, ("changes", "metadata", "expected_version"), short)
    changes = req.get("changes")
    if not isinstance(changes, (list, tuple)):
        raise BadRequest("missing field 'changes'")
    out = []
    for ch in changes:
        if not isinstance(ch, dict) or len(ch) != 1:
            raise BadRequest("a change is a map with exactly one of insert / delete / set_metadata")
        (op, arg), = _fields(ch, ("insert", "delete", "set_metadata"), short).items()
        if not isinstance(arg, dict):
            raise BadRequest("a change's argument is a map")
        if op == "insert":
            a = _fields(arg, ("id", "hashes"), short)
            if "id" not in a or not isinstance(a.get("hashes"), (list, tuple)):
                raise BadRequest("insert needs id and hashes")
            out.append(("insert", _u32(a["id"], "id"), [_u32(h, "hash") for h in a["hashes"]]))
        elif op == "delete":
            a = _fields(arg, ("id",), short)
            if "id" not in a:
                raise BadRequest("delete needs id")
            out.append(("delete", _u32(a["id"], "id")))
        else:
            out.append(("set_metadata", arg))
    ev = req.get("expected_version")
    if ev is not None and (isinstance(ev, bool) or not isinstance(ev, int) or ev < 0):
        raise BadRequest("expected_version must be an unsigned integer")
    return out, ev


def handle_update(multi_index, index_name, headers, body):
    """POST /:index/_update -> (status, content_type, body): each request is one commit = one memory segment."""
    headers = _lower(headers)
    try:
        ctype = request_type(headers, body)
        obj, short = decode_body(ctype, body)
        changes, ev = parse_changes(obj, short)
        version = multi_index.get_index(index_name).update(changes, expected_version=ev)
        rtype = response_type(headers, body)
        return 200, rtype, encode(rtype, {"version": version}, {"v": version})
    except Exception as e:
        return error_response(headers, body, e)


def handle_put_fingerprint(multi_index, index_name, fp_id, headers, body):
    """PUT /:index/:id {hashes} (src/server.zig:226-234)"""
    headers = _lower(headers)
    try:
        ctype = request_type(headers, body)
        obj, short = decode_body(ctype, body)
        if not isinstance(obj, dict):
            raise BadRequest("request body must be a map")
        a = _fields(obj, ("hashes",), short)
        if not isinstance(a.get("hashes"), (list, tuple)):
            raise BadRequest("missing field 'hashes'")
        hashes = [_u32(h, "hash") for h in a["hashes"]]
        multi_index.get_index(index_name).update([("insert", _u32(fp_id, "id"), hashes)])
        rtype = response_type(headers, body)
        return 200, rtype, encode(rtype, {}, {})
    except Exception as e:
        return error_response(headers, body, e)


def handle_delete_fingerprint(multi_index, index_name, fp_id, headers, body=b""):
    """DELETE /:index/:id (src/server.zig:236-243)"""
    headers = _lower(headers)
    try:
        multi_index.get_index(index_name).update([("delete", _u32(fp_id, "id"))])
        rtype = response_type(headers, body)
        return 200, rtype, encode(rtype, {}, {})
    except Exception as e:
        return error_response(headers, body, e)
