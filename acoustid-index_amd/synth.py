"""Seeded synthetic fingerprints and queries, numpy implementation.

The same definition exists in oracle/fpx_oracle.c (orc_synth_hash) and in csrc/fpx_build.hip;
tests check the three against each other."""
import numpy as np

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_G = np.uint64(0x9E3779B97F4A7C15)
_D = np.uint64(0xD1B54A32D192ED03)
_HOT = np.uint64(0x5bd1e9955bd1e995)


def mix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x + _G
        x = (x ^ (x >> np.uint64(30))) * _M1
        x = (x ^ (x >> np.uint64(27))) * _M2
        return x ^ (x >> np.uint64(31))


def synth_hashes(seed, docs, H, dist=0):
    """hashes[d, j] for doc ids `docs` (array) and j in [0, H) -> uint32 [len(docs), H]"""
    docs = np.asarray(docs, dtype=np.uint64)
    with np.errstate(over="ignore"):
        a = mix64(np.uint64(seed) + docs * _D)
        r = mix64(a[:, None] ^ np.arange(H, dtype=np.uint64)[None, :])
    out = (r >> np.uint64(32)).astype(np.uint32)
    if dist == 1:
        hot = (r & np.uint64(0xFFFF)) < np.uint64(1311)
        e = ((r >> np.uint64(16)) & np.uint64(0xFF)) % np.uint64(12)
        k = (np.uint64(1) << e) + ((r >> np.uint64(24)) & ((np.uint64(1) << e) - np.uint64(1))) - np.uint64(1)
        hv = (mix64(np.uint64(seed) ^ _HOT ^ (k << np.uint64(32))) >> np.uint64(32)).astype(np.uint32)
        out = np.where(hot, hv, out)
    return out


def synth_items(seed, first_doc, num_docs, H, dist=0):
    """sorted u64 items (hash << 32 | id) of docs [first_doc, first_doc + num_docs)"""
    docs = np.arange(first_doc, first_doc + num_docs, dtype=np.uint64)
    h = synth_hashes(seed, docs, H, dist).astype(np.uint64)
    items = ((h << np.uint64(32)) | docs[:, None]).ravel()
    items.sort()
    return items


def make_queries(seed, qseed, num_queries, total_docs, H, query_len=1000, dist=0, first_doc=1, flip_frac=0.10, first_query=0):
    """query q = all H hashes of a uniformly chosen target doc, one random bit flipped in
    ~10 % of them, padded with uniform noise hashes up to `query_len`.  Returns (flat u32, offsets u64, targets).
    first_query: the queries [first_query, first_query + num_queries) of the batch (a rank's share of a sharded batch)."""
    qs = np.arange(first_query, first_query + num_queries, dtype=np.uint64)
    r = mix64(np.uint64(qseed) ^ (qs * _D))
    targets = (np.uint64(first_doc) + r % np.uint64(total_docs)).astype(np.uint64)
    hs = synth_hashes(seed, targets, H, dist)                      # [Q, H]
    jr = mix64(mix64(np.uint64(qseed) + qs * _G)[:, None] ^ np.arange(max(H, query_len), dtype=np.uint64)[None, :])
    flip = (jr[:, :H] & np.uint64(0xFFFF)) < np.uint64(int(flip_frac * 65536))
    bit = ((jr[:, :H] >> np.uint64(16)) % np.uint64(32)).astype(np.uint32)
    hs = np.where(flip, hs ^ (np.uint32(1) << bit), hs)
    n = max(query_len, H)
    out = np.empty((num_queries, n), np.uint32)
    out[:, :H] = hs
    if n > H:
        out[:, H:] = (mix64(jr[:, H:] ^ np.uint64(0xABCDEF)) >> np.uint64(32)).astype(np.uint32)
    offsets = (np.arange(num_queries + 1, dtype=np.uint64) * np.uint64(n))
    return np.ascontiguousarray(out.ravel()), offsets, targets.astype(np.uint32)
