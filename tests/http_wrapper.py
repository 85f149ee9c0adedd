"""TEST HELPER: a stdlib HTTP server around acoustid-index_amd/frontend.py's handlers, used to replay the reference's
HTTP-level vectors.  Networking is out of scope for the product (the host stays Zig)."""
from fpx_testlib import fpx as _fpx

_fe = _fpx.frontend
handle_search, handle_update = _fe.handle_search, _fe.handle_update
handle_put_fingerprint, handle_delete_fingerprint = _fe.handle_put_fingerprint, _fe.handle_delete_fingerprint
response_type, encode, _lower, JSON = _fe.response_type, _fe.encode, _fe._lower, _fe.JSON


def serve(multi_index, host="127.0.0.1", port=6081, searcher=None):
    """A stdlib HTTP server around the handlers above (demonstration / replaying the reference's HTTP tests)."""
    from http.server import BaseHTTPRequestHandler, ThreadingHTTPServer

    class H(BaseHTTPRequestHandler):
        protocol_version = "HTTP/1.1"

        def _body(self):
            n = int(self.headers.get("Content-Length") or 0)
            return self.rfile.read(n) if n else b""

        def _send(self, status, ctype, body):
            self.send_response(status)
            self.send_header("Content-Type", ctype)
            self.send_header("Content-Length", str(len(body)))
            self.end_headers()
            self.wfile.write(body)

        def _route(self, method):
            parts = [p for p in self.path.split("?")[0].split("/") if p]
            hdrs, body = dict(self.headers.items()), self._body()
            if method == "GET" and parts == ["_health"]:
                return self._send(200, "text/plain", b"OK\n")
            if len(parts) == 1 and method == "PUT":
                multi_index.create_index(parts[0])
                rt = response_type(_lower(hdrs), body)
                return self._send(200, rt, encode(rt, {}, {}))
            if len(parts) == 1 and method == "DELETE":
                multi_index.delete_index(parts[0])
                rt = response_type(_lower(hdrs), body)
                return self._send(200, rt, encode(rt, {}, {}))
            if len(parts) == 2 and method == "POST" and parts[1] == "_search":
                return self._send(*handle_search(multi_index, parts[0], hdrs, body, searcher))
            if len(parts) == 2 and method == "POST" and parts[1] == "_update":
                return self._send(*handle_update(multi_index, parts[0], hdrs, body))
            if len(parts) == 2 and parts[1].isdigit() and method == "PUT":
                return self._send(*handle_put_fingerprint(multi_index, parts[0], int(parts[1]), hdrs, body))
            if len(parts) == 2 and parts[1].isdigit() and method == "DELETE":
                return self._send(*handle_delete_fingerprint(multi_index, parts[0], int(parts[1]), hdrs, body))
            return self._send(404, JSON, b'{"error":"NotFound"}')

        def do_GET(self): self._route("GET")          # noqa: E704
        def do_PUT(self): self._route("PUT")          # noqa: E704
        def do_POST(self): self._route("POST")        # noqa: E704
        def do_DELETE(self): self._route("DELETE")    # noqa: E704

        def log_message(self, *a):
            pass

    return ThreadingHTTPServer((host, port), H)
