import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from fpx_testlib import fpx, oracle, Pair
ctx = fpx.Context(0)
seed, H, n = 77, 32, 6000
def q_of(seed_, doc): return fpx.synth.synth_hashes(seed_, [doc], H)[0]
for variant in range(3):
    p = Pair(ctx)
    p.add_file(fpx.synth.synth_items(seed, 1, n, H), 1, n, 1, np.arange(1, n + 1))
    if variant >= 1:
        p.add_memory_changes([("delete", 1)], 4)
    if variant >= 2:
        p.add_memory_changes([("insert", 9, [1,2,3])], 5)
    p.finish()
    qs = [q_of(seed, d) for d in range(1, 8)]
    got, st = p.reader.search_batch(qs, fpx.SearchOptions(10, 1, 10))
    want = [p.osnap.search(q, 10, 1, 10) for q in qs]
    print("variant", variant)
    for g, w in zip(got, want): print("  ", g[:3], w[:3])
