"""Request coalescer (SURVEY.md 8(b) threading note, 8(f)-2): concurrent single searches from many host threads are
collected into one device batch (fpx_search_batch) so that launches are amortised, under each request's own timeout.
The C++ twin for compiled hosts is host/fpx_coalescer.hpp.

A request waits at most `max_wait_ms` for company (or until `max_batch` requests are queued); the dispatcher thread
groups the queued requests by the snapshot they captured at submit time -- a search always runs on the snapshot that
was current when it arrived (src/Index.zig:430-434) -- and issues one batched search per snapshot.
"""
import threading
import time
from concurrent.futures import Future
from concurrent.futures import TimeoutError as FutureTimeout

from ._lib import SearchTimeout


class SearchCoalescer:
    def __init__(self, max_batch=1024, max_wait_ms=1.0):
        self.max_batch, self.max_wait = max_batch, max_wait_ms / 1e3
        self._q = []
        self._cv = threading.Condition()
        self._stop = False
        self.batches = 0
        self.requests = 0
        self._t = threading.Thread(target=self._run, name="fpx-coalescer", daemon=True)
        self._t.start()

    def close(self):
        with self._cv:
            self._stop = True
            self._cv.notify_all()
        self._t.join()

    # searcher signature of frontend.handle_search
    def __call__(self, index, hashes, options, timeout_ms):
        return self.search(index.acquire_reader(), hashes, options, timeout_ms)

    def search(self, reader, hashes, options, timeout_ms=0):
        fut = Future()
        deadline = time.monotonic() + timeout_ms / 1e3 if timeout_ms else None
        with self._cv:
            self._q.append((reader, list(hashes), options, deadline, fut))
            self._cv.notify_all()
        try:
            return fut.result(timeout=None if deadline is None else max(0.0, deadline - time.monotonic()))
        except FutureTimeout:
            raise SearchTimeout("search timed out waiting for its batch") from None

    def _run(self):
        while True:
            with self._cv:
                while not self._q and not self._stop:
                    self._cv.wait()
                if self._stop and not self._q:
                    return
                t_first = time.monotonic()
                while len(self._q) < self.max_batch and not self._stop:
                    left = self.max_wait - (time.monotonic() - t_first)
                    if left <= 0:
                        break
                    self._cv.wait(left)
                batch, self._q = self._q[:self.max_batch], self._q[self.max_batch:]
            groups = {}
            for item in batch:
                groups.setdefault(id(item[0].snapshot), []).append(item)
            for items in groups.values():
                now = time.monotonic()
                live = []
                for it in items:
                    if it[3] is not None and it[3] <= now:
                        it[4].set_exception(SearchTimeout("deadline passed before dispatch"))
                    else:
                        live.append(it)
                if not live:
                    continue
                # A device batch carries ONE deadline (the kernels poll one cancel word): requests are ordered by theirs and
                # the batch is cut wherever the next deadline is more than 4x further away, the most urgent class first --
                # an impatient request neither waits behind, nor is cancelled together with, a patient one.  Inside a class the
                # batch may run as long as its most patient member allows; every caller still enforces its own deadline on
                # the future.
                live.sort(key=lambda it: float("inf") if it[3] is None else it[3])
                classes, cur = [], [live[0]]
                for it in live[1:]:
                    a = cur[0][3]
                    far = (it[3] is None) != (a is None) or (a is not None and (it[3] - now) > 4.0 * max(a - now, 1e-3))
                    if far:
                        classes.append(cur)
                        cur = [it]
                    else:
                        cur.append(it)
                classes.append(cur)
                for grp in classes:
                    now2 = time.monotonic()
                    timed = all(it[3] is not None for it in grp)
                    tmo = max(1, int((max(it[3] for it in grp) - now2) * 1e3)) if timed else 0
                    try:
                        res, _ = grp[0][0].search_batch([it[1] for it in grp], [it[2] for it in grp], timeout_ms=tmo)
                        for it, r in zip(grp, res):
                            it[4].set_result(r)
                    except Exception as e:             # the whole class shares the failure (e.g. FPX_E_TIMEOUT)
                        for it in grp:
                            if not it[4].done():
                                it[4].set_exception(e)
                    self.batches += 1
                    self.requests += len(grp)
