"""TEST HELPER (not part of the product package: src/legacy.zig is out of scope).
Replays the reference's tests/test_legacy.py vectors through the GPU search path.
The legacy line protocol's path into the search (src/legacy.zig): `search h1,h2,...` with the
session's options (limit 500, min_score 1, top_score_percent 10, not clamped like the HTTP front end), plus the small
transaction vocabulary the reference's tests drive it with (begin / insert / commit / rollback, get / set of session
attributes).  A Session maps one text line to one reply line ("OK ..." / "ERR ..."); sockets are the host's business.
Index attributes (metadata) are kept per MultiIndex in memory only."""
from fpx_testlib import fpx as _fpx

_ix = _fpx.index
SearchTimeout = _fpx.SearchTimeout
IndexNotFound = _fpx.hostindex.IndexNotFound

INDEX_NAME = "main"                       # src/legacy.zig: the legacy protocol serves one fixed index


class LegacySession:
    def __init__(self, multi_index, read_only=False, searcher=None):
        self.mi, self.read_only, self.searcher = multi_index, read_only, searcher
        self.attr = {"max_results": 500, "top_score_percent": 10, "timeout": 0, "idle_timeout": 60000}   # :44-47
        self.in_txn = False
        self.changes, self.pending_attrs = [], {}
        if not hasattr(multi_index, "legacy_attrs"):
            multi_index.legacy_attrs = {}

    # ---- one line in, one line out (src/legacy.zig:144-182, reply :132-137)
    def cmd(self, line):
        kind, payload = self._dispatch(line.rstrip("\r\n"))
        return ("OK " if kind else "ERR ") + payload

    @staticmethod
    def parse_fingerprint(s):
        """comma-separated SIGNED decimals reinterpreted as u32 (src/legacy.zig:318-330)"""
        if not s:
            raise ValueError("empty fingerprint")
        out = []
        for tok in s.split(","):
            try:
                v = int(tok, 10)
            except ValueError:
                raise ValueError("invalid fingerprint") from None
            if not (tok.lstrip("+-").isdigit() and -(1 << 63) <= v < (1 << 63)):
                raise ValueError("invalid fingerprint")
            out.append(v & 0xFFFFFFFF)
        return out

    def _dispatch(self, line):
        toks = [t for t in line.split(" ") if t]
        if not toks:
            return True, ""
        cmd, a = toks[0], toks[1:]
        if cmd == "echo":
            return True, " ".join(a)
        if cmd == "search":
            return self._search(a)
        if cmd == "insert":
            if not self.in_txn:
                return False, "not in transaction"
            if len(a) != 2:
                return False, "expected two arguments"
            if not (a[0].isdigit() and int(a[0]) <= 0xFFFFFFFF):
                return False, "invalid document id"
            try:
                hashes = self.parse_fingerprint(a[1])
            except ValueError as e:
                return False, str(e)
            self.changes.append(("insert", int(a[0]), hashes))
            return True, ""
        if cmd == "begin":
            if self.read_only:
                return False, "read-only replica"
            if self.in_txn:
                return False, "already in transaction"
            self.changes, self.pending_attrs, self.in_txn = [], {}, True
            return True, ""
        if cmd == "commit":
            if not self.in_txn:
                return False, "not in transaction"
            if self.changes or self.pending_attrs:
                try:
                    self.mi.create_index(INDEX_NAME).update(self.changes)
                    self.mi.legacy_attrs.update(self.pending_attrs)
                except Exception:
                    return False, "commit failed"
            self.in_txn, self.changes, self.pending_attrs = False, [], {}
            return True, ""
        if cmd == "rollback":
            if not self.in_txn:
                return False, "not in transaction"
            self.in_txn, self.changes, self.pending_attrs = False, [], {}
            return True, ""
        if cmd in ("optimize", "cleanup"):
            return (True, "") if self.in_txn else (False, "not in transaction")
        if cmd == "get":
            name = a[0] if len(a) == 1 else (a[1] if len(a) == 2 and a[0] == "attribute" else None)
            if name is None:
                return False, "expected one argument"
            if name in self.attr:
                return True, str(self.attr[name])
            return True, self.mi.legacy_attrs.get(name, "")
        if cmd == "set":
            if len(a) == 2:
                name, value = a
            elif len(a) == 3 and a[0] == "attribute":
                name, value = a[1], a[2]
            else:
                return False, "expected two arguments"
            if name in self.attr:
                if not (value.isdigit() and int(value) <= 0xFFFFFFFF):
                    return False, "invalid value"
                self.attr[name] = int(value)
                return True, ""
            if not self.in_txn:
                return False, "not in transaction"
            self.pending_attrs[name] = value
            return True, ""
        return False, "unknown command"

    def _search(self, a):
        if len(a) != 1:
            return False, "expected one argument"
        try:
            hashes = self.parse_fingerprint(a[0])
        except ValueError as e:
            return False, str(e)
        opts = _ix.SearchOptions(max_results=self.attr["max_results"], min_score=1,
                                 min_score_pct=self.attr["top_score_percent"])          # :192-198
        try:
            index = self.mi.get_index(INDEX_NAME)
            if self.searcher is not None:
                results = self.searcher(index, hashes, opts, self.attr["timeout"])
            else:
                res = _ix.SearchResults(opts)
                index.acquire_reader().search(hashes, res, timeout_ms=self.attr["timeout"])
                results = res.getResults()
        except SearchTimeout:
            return False, "timeout exceeded"
        except IndexNotFound:
            results = []                     # nothing committed yet: the reference's index exists but is empty
        except Exception:
            return False, "search failed"
        return True, " ".join(f"{i}:{s}" for i, s in results)
