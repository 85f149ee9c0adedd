"""GPU parity tests: the HIP path through the C ABI vs the CPU oracle, bit-exact {id, score} lists,
plus the reference's scanned-blocks / scanned-docs totals."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


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
