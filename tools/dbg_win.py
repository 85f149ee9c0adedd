import os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from fpx_testlib import fpx, oracle, Pair
world = 2
ctxs = [fpx.Context(0) for k in range(world)]
for c in ctxs:
    c.set_option("direct", 1); c.set_option("direct_min_items", 0); c.set_option("fuse_min", 1)
seed, S, per, H = 62, 2, 5000, 48
full = Pair(ctxs[0])
slices = [[] for _ in range(world)]
for s in range(S):
    lo = s * per + 1
    ids = np.arange(lo, lo + per, dtype=np.uint64)
    h = fpx.synth.synth_hashes(seed + s, ids, H, 1).astype(np.uint64)
    items = np.sort(((h << np.uint64(32)) | ids[:, None]).ravel())
    blocks, index = full.add_file(items, lo, lo + per - 1, s + 1, ids.astype(np.uint32))
    for k, sl in enumerate(fpx.file_segment_windows(ctxs, blocks, 512, index, lo, lo + per - 1, s + 1, ids.astype(np.uint32))):
        slices[k].append(sl)
B = 16
flat, off, _ = fpx.synth.make_queries(seed, 3, B, S * per, H, query_len=160, dist=1)
queries = [flat[int(off[i]):int(off[i + 1])] for i in range(B)]
q0 = [int(h) for h in queries[0][:60]]
for changes in ([("insert", 900001, q0)], [("insert", 900001, q0[:30])]):
    commit = len(full.gpu_segs) + 1
    full.add_memory_changes(changes, commit)
    m = full.orc_mem[-1]
    ids, alive = m.docs()
    print("mem items", len(m.items()), "docs", ids, alive, "commit", commit)
    for k in range(world):
        slices[k].append(fpx.MemorySegment(ctxs[k], m.items(), m.min_doc_id, m.max_doc_id, commit, ids, alive))
full.finish()
opts = fpx.http_options()
want, _ = full.reader.search_batch(queries, opts)
print("unsharded q0", want[0][:3], "oracle", full.osnap.search(queries[0])[:3])
for k in range(world):
    snap = fpx.Segments(ctxs[k], slices[k])
    print("rank", k, "info", snap.info())
    got, st = fpx.IndexReader(snap).search_batch(queries, fpx.SearchOptions(40, 1, 0))
    print("rank", k, "alone q0", [x for x in got[0] if x[0] == 900001], "hits", st.hits)
sh = fpx.ShardedIndexReader(fpx.WindowShardedSegments(ctxs, slices))
got, st = sh.search_batch(queries, opts)
print("sharded q0", got[0][:3], "hits", st.hits)
got, st = sh.search_batch(queries, fpx.SearchOptions(40, 1, 0))
print("sharded legacy q0", got[0][:3])
