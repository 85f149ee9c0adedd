"""Shared helpers for the parity tests: build the SAME index for the GPU path and the CPU oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from __graft_entry__ import load_package  # noqa: E402
from oracle import oracle  # noqa: E402

fpx = load_package()


class Pair:
    """One logical index held twice: resident on the GPU (fpx) and in host RAM (oracle)."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.gpu_segs, self.orc_file, self.orc_mem = [], [], []

    def add_file(self, items, min_doc_id, max_doc_id, commit_id, doc_ids, doc_alive=None, block_size=512):
        blocks, index = oracle.build_blocks(items, min_doc_id, block_size)
        self.gpu_segs.append(fpx.FileSegment(self.ctx, blocks, block_size, index, min_doc_id, max_doc_id, commit_id,
                                             doc_ids, doc_alive))
        self.orc_file.append(oracle.file_segment(blocks, block_size, index, min_doc_id, max_doc_id, commit_id,
                                                 doc_ids, doc_alive))
        return blocks, index

    def add_memory(self, items, min_doc_id, max_doc_id, commit_id, doc_ids, doc_alive=None):
        self.gpu_segs.append(fpx.MemorySegment(self.ctx, items, min_doc_id, max_doc_id, commit_id, doc_ids, doc_alive))
        self.orc_mem.append(oracle.memory_segment(items, min_doc_id, max_doc_id, commit_id, doc_ids, doc_alive))

    def add_memory_changes(self, changes, commit_id):
        m = oracle.memory_segment_from_changes(changes, commit_id)
        ids, alive = m.docs()
        self.gpu_segs.append(fpx.MemorySegment(self.ctx, m.items(), m.min_doc_id, m.max_doc_id, commit_id, ids, alive))
        self.orc_mem.append(m)

    def finish(self):
        self.reader = fpx.IndexReader(fpx.Segments(self.ctx, self.gpu_segs))
        self.osnap = oracle.Snapshot(self.orc_file, self.orc_mem)
        return self

    def check(self, queries, options, with_stats=True):
        got, st = self.reader.search_batch(queries, options)
        # the same batch again: a workspace's first batch takes the general path (it measures the record count the
        # device-sized path sizes its bins with), the repeat takes the device-sized one -- same results, same counters
        got2, st2 = self.reader.search_batch(queries, options)
        assert got2 == got, "device-sized path differs from the general path"
        assert (st2.scanned_blocks, st2.scanned_docs, st2.probes, st2.hits) == (st.scanned_blocks, st.scanned_docs, st.probes, st.hits), \
            ((st2.scanned_blocks, st2.scanned_docs, st2.probes, st2.hits), (st.scanned_blocks, st.scanned_docs, st.probes, st.hits))
        # (which path a run took is in st.path_flags; a batch the device-sized path hands back -- a full bin, scores too wide
        # for the candidate key -- is redone on the general one and the next few batches of that workspace skip the attempt)
        opts = options if isinstance(options, list) else [options] * len(queries)
        # ... and once more with per-query scan statistics (fpx_search_batch_stats): the reference's num_blocks / num_docs per hash
        # (src/FileSegment.zig:177-178), summed per query, must be the oracle's for EVERY query
        got3, st3, qblocks, qdocs = self.reader.search_batch_stats(queries, options)
        assert got3 == got and (st3.scanned_blocks, st3.scanned_docs) == (st.scanned_blocks, st.scanned_docs)
        blocks = docs = 0
        for i, (q, o) in enumerate(zip(queries, opts)):
            want, ost = self.osnap.search(q, o.max_results, o.min_score, o.min_score_pct, with_stats=True)
            assert got[i] == want, f"query {i}: gpu {got[i][:8]} != oracle {want[:8]}"
            if with_stats:
                assert (int(qblocks[i]), int(qdocs[i])) == (ost.scanned_blocks, ost.scanned_docs), \
                    f"query {i}: scanned blocks / docs {int(qblocks[i])} / {int(qdocs[i])}, oracle {ost.scanned_blocks} / {ost.scanned_docs}"
            blocks += ost.scanned_blocks
            docs += ost.scanned_docs
        if with_stats:
            assert st.scanned_blocks == blocks, (st.scanned_blocks, blocks)
            assert st.scanned_docs == docs, (st.scanned_docs, docs)
        return got, st
