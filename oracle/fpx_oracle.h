/*
 * fpx_oracle.h -- CPU oracle for the fpindex /_search hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's
 * CPU algorithm (acoustid/acoustid-index, Zig) used as the checker in tests/,
 * __graft_entry__.smoke() and the cpu_baseline leg of bench.py.  Nothing on the
 * product path (libfpx.so, the acoustid-index_amd package) may link or call it.
 *
 * Parity status: the reference is Zig 0.16 with four un-vendored dependencies; no
 * Zig toolchain exists in the build image, so the reference itself cannot be run
 * here (oracle/_ref is therefore absent -- see DESIGN.md).  The restatement is
 * pinned instead against every known-answer test the reference's own test blocks
 * hold for this path (tests/test_oracle_kat.py transcribes them as data:
 * src/streamvbyte.zig:518-908, src/block.zig:317-719, src/segment.zig:112-143,
 * src/filefmt.zig:293-338, src/Index.zig:1056-1479, tests/test_fingerprint_api.py).
 *
 * Every function cites the reference file:line it follows.
 */
#ifndef FPX_ORACLE_H
#define FPX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- StreamVByte (src/streamvbyte.zig) ---------------------------------- */
enum { ORC_V0124 = 0, ORC_V0124_MINUS1 = 1, ORC_V1234 = 2 }; /* :63-67 */

uint8_t orc_svb_length(int variant, uint8_t control);                     /* :178-211 */
size_t  orc_svb_decode_quad(int variant, uint8_t control, const uint8_t *in,
                            uint32_t out[4]);                              /* :216-258 */
size_t  orc_svb_decode_quad_delta(int variant, uint8_t control, const uint8_t *in,
                                  uint32_t out[4], uint32_t carry);        /* :264-283 */
void    orc_svb_delta_decode_in_place(uint32_t *data, size_t n, uint32_t first); /* :287-339 */
void    orc_svb_decode_values(size_t total_items, size_t start_item, size_t end_item,
                              const uint8_t *in, uint32_t *out, int variant,
                              int delta, uint32_t first_value);            /* :341-412 */
size_t  orc_svb_encode_quad_0124(const uint32_t in[4], uint8_t *out_data, uint8_t *out_control); /* :461 */
size_t  orc_svb_encode_quad_1234(const uint32_t in[4], uint8_t *out_data, uint8_t *out_control); /* :472 */
size_t  orc_svb_encode_quad_size_0124(const uint32_t in[4]);               /* :483 */
size_t  orc_svb_encode_quad_size_1234(const uint32_t in[4]);               /* :501 */
/* select the decode implementation: 0 = scalar, 1 = SSSE3 pshufb (as the reference, :32-60) */
void    orc_set_simd(int on);
int     orc_get_simd(void);

/* ---- Block codec (src/block.zig) ---------------------------------------- */
#define ORC_MIN_BLOCK_SIZE 64
#define ORC_MAX_BLOCK_SIZE 4096
#define ORC_MAX_ITEMS_PER_BLOCK (ORC_MAX_BLOCK_SIZE / 2)
#define ORC_BLOCK_HEADER_SIZE 8

typedef struct { uint32_t min_hash; uint16_t num_items; uint16_t docids_offset; } orc_block_header; /* :46-50 */

/* items are u64 = hash<<32 | id (src/segment.zig:87-106) */
size_t orc_block_encode(const uint64_t *items, size_t n, uint32_t min_doc_id,
                        uint8_t *out, size_t block_size);                  /* :438-567 */
void   orc_block_header_decode(const uint8_t *data, orc_block_header *h);  /* :53-56 */
/* full decode of one block into hashes[]/docids[] (BlockReader.getItems, :137-203,:282-305) */
size_t orc_block_decode_items(const uint8_t *block, size_t block_len, uint32_t min_doc_id,
                              uint32_t *hashes, uint32_t *docids);
/* BlockReader.findHash (:217-231): writes [start,end) */
void   orc_block_find_hash(const uint8_t *block, size_t block_len, uint32_t hash,
                           uint32_t *start, uint32_t *end);
/* BlockReader.searchHash (:268-271): returns number of docids written to out */
size_t orc_block_search_hash(const uint8_t *block, size_t block_len, uint32_t min_doc_id,
                             uint32_t hash, uint32_t *out);

/* ---- Segments / snapshot / search --------------------------------------- */
typedef struct orc_segment orc_segment;
typedef struct orc_snapshot orc_snapshot;
typedef struct { uint32_t id; uint32_t score; } orc_result;               /* src/common.zig:45-48 */
typedef struct {
    uint64_t scanned_blocks;   /* sum of FileSegment.search's num_blocks (src/FileSegment.zig:154,171) */
    uint64_t scanned_docs;     /* sum of num_docs (:153,172) */
    uint64_t hits_unique;      /* hit-map entries before finish */
    uint64_t probes;           /* (unique hash, file segment) pairs */
} orc_stats;

/* Build the blocks + block_index of a file segment from sorted items exactly as
 * filefmt.writeBlocks does (src/filefmt.zig:94-138): greedy fill, 2048-item window,
 * index entry = last consumed hash, one all-zero terminator block.
 * Returns malloc'd buffers (free with orc_free). blocks_len includes the terminator. */
int  orc_build_blocks(const uint64_t *sorted_items, size_t n, uint32_t min_doc_id,
                      uint32_t block_size, uint8_t **blocks, size_t *blocks_len,
                      uint32_t **block_index, uint32_t *num_blocks);
void orc_free(void *p);

/* A resident file segment (fields of src/FileSegment.zig:33-48). Buffers are copied. */
orc_segment *orc_segment_create_file(const uint8_t *blocks, size_t blocks_len, uint32_t block_size,
                                     const uint32_t *block_index, uint32_t num_blocks,
                                     uint32_t min_doc_id, uint32_t max_doc_id, uint64_t commit_id,
                                     const uint32_t *doc_ids, const uint8_t *doc_alive, uint32_t num_docs);
/* Same, but borrows blocks/block_index (caller keeps them alive) -- for multi-GB bench samples */
orc_segment *orc_segment_create_file_borrowed(const uint8_t *blocks, size_t blocks_len, uint32_t block_size,
                                     const uint32_t *block_index, uint32_t num_blocks,
                                     uint32_t min_doc_id, uint32_t max_doc_id, uint64_t commit_id,
                                     const uint32_t *doc_ids, const uint8_t *doc_alive, uint32_t num_docs);
/* A memory segment from already sorted items (src/MemorySegment.zig:27-28) */
orc_segment *orc_segment_create_memory(const uint64_t *items, size_t n,
                                       uint32_t min_doc_id, uint32_t max_doc_id, uint64_t commit_id,
                                       const uint32_t *doc_ids, const uint8_t *doc_alive, uint32_t num_docs);
/* MemorySegment.build (src/MemorySegment.zig:81-148) from a change batch:
 * kind[i] 0 = insert, 1 = delete; hashes of change i are hashes[hash_off[i]..hash_off[i+1]) */
orc_segment *orc_segment_build_memory(const uint8_t *kind, const uint32_t *ids,
                                      const uint32_t *hashes, const uint64_t *hash_off,
                                      size_t num_changes, uint64_t commit_id);
/* test aid for hash-range sharding (include/fpx.h: fpx_segment_create_file_slice): scan only hashes in
 * (lo_excl, hi_incl] in this segment -- the segment then stands for one slice */
void orc_segment_set_window(orc_segment *s, int has_lo, uint32_t lo_excl, int has_hi, uint32_t hi_incl);
void orc_segment_free(orc_segment *s);
/* introspection for tests */
size_t orc_segment_num_items(const orc_segment *s);
const uint64_t *orc_segment_items(const orc_segment *s);   /* memory segments only */
uint32_t orc_segment_min_doc_id(const orc_segment *s);
uint32_t orc_segment_max_doc_id(const orc_segment *s);
uint32_t orc_segment_num_docs(const orc_segment *s);
const uint32_t *orc_segment_doc_ids(const orc_segment *s);
const uint8_t *orc_segment_doc_alive(const orc_segment *s);

/* Segments snapshot (src/Index.zig:36-150): file[] then memory[], oldest -> newest. Borrows segments. */
orc_snapshot *orc_snapshot_create(orc_segment *const *file, uint32_t n_file,
                                  orc_segment *const *memory, uint32_t n_memory);
void orc_snapshot_free(orc_snapshot *s);
int  orc_snapshot_has_newer_commit(const orc_snapshot *s, uint32_t id, uint64_t commit_id); /* :133-149 */

/* SearchOptions derivation (src/MultiIndex.zig:302-306): has_min_score==0 -> (raw_len+19)/20 */
uint32_t orc_default_min_score(uint32_t raw_query_len);

/* IndexReader.search + SearchResults.finish + getResults
 * (src/Index.zig:170-177, src/common.zig:121-173). `hashes` is raw (unsorted, dups allowed)
 * and is NOT modified.  Returns number of results written (<= out_cap), or -1 on OOM. */
int  orc_search(const orc_snapshot *snap, const uint32_t *hashes, uint32_t n,
                uint32_t max_results, uint32_t min_score, uint32_t min_score_pct,
                orc_result *out, uint32_t out_cap, orc_stats *stats);

/* Many searches at once -- the CPU baseline of bench.py.  `nthreads` persistent workers, one search per thread at a time
 * on the shared snapshot with recycled collectors, as the reference's executors do (src/main.zig:272-276,
 * src/common.zig:186-300).  Query q is hashes[offsets[q] .. offsets[q+1]); has_min_score == 0 -> (raw_len + 19) / 20.
 * The workers cycle over the query set until min_seconds have passed (every query at least once); the results of the
 * FIRST pass go to out[q * out_cap ..] / out_n[q]; latency_ms[i] is the duration of the i-th search handed out
 * (i < latency_cap).  wall_seconds / queries_done give the throughput.  Returns 0, -1 on failure. */
int  orc_search_many(const orc_snapshot *snap, const uint32_t *hashes, const uint64_t *offsets, uint32_t nq,
                     uint32_t max_results, int has_min_score, uint32_t min_score, uint32_t min_score_pct,
                     uint32_t nthreads, double min_seconds,
                     orc_result *out, uint32_t out_cap, uint32_t *out_n,
                     float *latency_ms, uint64_t latency_cap,
                     double *wall_seconds, uint64_t *queries_done);

/* Effective parallelism of the box for `nthreads` compute-bound threads (a container may cap CPU time below the visible
 * core count): aggregate spin rate of nthreads threads over the rate of one thread alone, each measured for `seconds`. */
double orc_cpu_parallelism(uint32_t nthreads, double seconds);

/* The hit map after the segment scans and before finish (id, commit_id, score), for tests
 * that pin `results.hits.get(id).score` (src/filefmt.zig:336-337, src/Index.zig:1079). */
int  orc_search_hits(const orc_snapshot *snap, const uint32_t *hashes, uint32_t n,
                     uint32_t *ids, uint64_t *commit_ids, uint32_t *scores, uint32_t cap);

/* ---- SegmentMerger (src/segment_merger.zig:85-151): prepare() + the read()/advance() k-way merge ----
 * `collection` answers hasNewerCommit; `sources` are segments of it, oldest to newest.  A doc of a source that has a
 * newer commit in the collection is skipped (its items too); the others keep their alive/tombstone status in the merged
 * docs map.  min/max_doc_id cover the kept docs (0/0 when none); commit_id is the smallest of the sources
 * (SegmentInfo.merge, src/segment.zig:38-51).  Outputs are malloc'd (orc_free); doc ids ascending. Returns 0, -1 on OOM,
 * -2 when there are no sources (error.NoSources). */
int orc_merge_segments(const orc_snapshot *collection, orc_segment *const *sources, uint32_t n_sources,
                       uint64_t **items, size_t *num_items,
                       uint32_t **doc_ids, uint8_t **doc_alive, uint32_t *num_docs,
                       uint32_t *min_doc_id, uint32_t *max_doc_id, uint64_t *commit_id);

/* ---- seeded synthetic fingerprints (shared definition with the GPU builder) -- */
uint64_t orc_mix64(uint64_t x);
/* hash j of document `doc` under `seed`; dist 0 = uniform u32, 1 = 2 % hot pool */
uint32_t orc_synth_hash(uint64_t seed, uint32_t doc, uint32_t j, int dist);
/* fill items[(doc-first_doc)*H + j] = hash<<32|doc for docs [first_doc, first_doc+num_docs), then sort */
/* the same items SORTED by (hash, doc), on `nthreads` host threads; out, tmp: num_docs * H items each */
int orc_synth_items_sorted_mt(uint64_t seed, uint32_t first_doc, uint32_t num_docs, uint32_t H, int dist, uint64_t *out, uint64_t *tmp, uint32_t nthreads);
void orc_synth_items(uint64_t seed, uint32_t first_doc, uint32_t num_docs, uint32_t H, int dist,
                     uint64_t *items);
void orc_sort_u64(uint64_t *v, size_t n);

#ifdef __cplusplus
}
#endif
#endif
