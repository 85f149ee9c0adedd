"""The GPU segment builder must emit byte-for-byte what the reference writer would
(oracle.build_blocks = src/filefmt.zig:94-138 + src/block.zig:438-567) for the same sorted items."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from fpx_testlib import fpx, oracle
    return fpx, oracle, fpx.Context(0)


@pytest.mark.parametrize("first_doc,num_docs,H,dist,block_size", [
    (1, 5000, 64, 0, 512),
    (1, 5000, 63, 1, 512),        # odd item count -> partial last quad
    (1000001, 3001, 33, 1, 512),  # large doc ids, n % 4 == 1
    (1, 2000, 50, 1, 64),
    (1, 2000, 50, 0, 100),
    (1, 20000, 40, 1, 4096),
    (7, 1, 3, 0, 512),            # a single partial quad
    (1, 300000, 16, 0, 512),      # many chunks: exercises the fixpoint across chunk boundaries
])
def test_builder_bytes_match_reference_writer(env, first_doc, num_docs, H, dist, block_size):
    fpx, oracle, ctx = env
    seed = 1234 + num_docs
    seg = fpx.FileSegment.synth(ctx, seed, first_doc, num_docs, H, dist, block_size, 1)
    blocks, index = seg.download()
    items = fpx.synth.synth_items(seed, first_doc, num_docs, H, dist)
    # the three generators agree (numpy here, C in the oracle, HIP in the builder)
    assert np.array_equal(items, oracle.synth_items(seed, first_doc, num_docs, H, dist))
    want_blocks, want_index = oracle.build_blocks(items, first_doc, block_size)
    assert seg.num_blocks == len(want_index)
    assert np.array_equal(index, want_index)
    assert blocks.size == want_blocks.size and np.array_equal(blocks, want_blocks)
    assert seg.getSize() == num_docs * H


def test_builder_degenerate_equal_costs(env):
    """All items cost the same -> chains started at different quads never merge: the fixpoint needs
    one round per chunk and must still be exact."""
    fpx, oracle, ctx = env
    seg = fpx.FileSegment.synth(ctx, 5, 1, 40000, 1, 0, 512, 1)
    blocks, index = seg.download()
    items = fpx.synth.synth_items(5, 1, 40000, 1, 0)
    wb, wi = oracle.build_blocks(items, 1, 512)
    assert np.array_equal(index, wi) and np.array_equal(blocks, wb)


def test_search_on_gpu_built_segments(env):
    fpx, oracle, ctx = env
    seed, H, per = 9, 64, 20000
    segs, osegs = [], []
    for s in range(3):
        lo = s * per + 1
        g = fpx.FileSegment.synth(ctx, seed, lo, per, H, 0, 512, s + 1)
        blocks, index = g.download()
        segs.append(g)
        osegs.append(oracle.file_segment(blocks, 512, index, lo, lo + per - 1, s + 1, np.arange(lo, lo + per)))
    reader = fpx.IndexReader(fpx.Segments(ctx, segs))
    osnap = oracle.Snapshot(osegs, [])
    flat, off, targets = fpx.synth.make_queries(seed, 5, 48, 3 * per, H, query_len=400)
    qs = [flat[int(off[i]):int(off[i + 1])] for i in range(48)]
    got, st = reader.search_batch(qs, fpx.http_options())
    for q, g, t in zip(qs, got, targets):
        assert g == osnap.search(q) and g[0][0] == int(t)
