/*
 * fpx.h -- C ABI of libfpx, the MI355X-native search path for the AcoustID
 * fingerprint inverted index ("fpindex", acoustid/acoustid-index).
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.
 * The reference has no FFI for this path today (it is ordinary Zig method calls),
 * so each entry point cites the reference call site it replaces; the Zig-side
 * `extern fn` declarations a maintainer would add are in INTEGRATION.md.
 *
 * Threading: every function is re-entrant.  Searches may be issued concurrently
 * from many OS threads on one snapshot (the reference runs one search per zio
 * executor thread on an immutable snapshot, src/main.zig:272-276, src/Index.zig:1-6);
 * each call takes a pooled device workspace + HIP stream (the analogue of
 * SearchResultsPool, src/common.zig:186-300) and holds no lock while the GPU works.
 *
 * Ownership: inputs are borrowed for the duration of the call and copied to HBM;
 * outputs are written to caller-provided memory; handles are reference counted and
 * a segment stays resident while any snapshot references it (src/Index.zig:53-63,
 * src/FileSegment.zig:62-73, src/shared_ptr.zig:68-78).
 *
 * Errors: int status, 0 = success.  No exceptions or longjmp cross this boundary.
 */
#ifndef FPX_H
#define FPX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FPX_OK          0
#define FPX_E_NOMEM    -1   /* error.OutOfMemory (host or HBM) */
#define FPX_E_TIMEOUT  -2   /* error.SearchTimeout (src/MultiIndex.zig:319-322): no partial results */
#define FPX_E_DEVICE   -3   /* HIP runtime / kernel failure */
#define FPX_E_INVAL    -4   /* malformed argument */
#define FPX_E_NODEVICE -5   /* no gfx950 device visible: the HIP path is the only path */
#define FPX_E_AGAIN    -6   /* a caller-provided device buffer was too small; the call reports the size it needs: retry */

typedef struct fpx_ctx fpx_ctx;
typedef struct fpx_segment fpx_segment;
typedef struct fpx_snapshot fpx_snapshot;

/* SearchResult (src/common.zig:45-48, src/api.zig:57-60) */
typedef struct { uint32_t id; uint32_t score; } fpx_result;

/* Per-query SearchOptions (src/common.zig:50-54) as derived by MultiIndex.search
 * (src/MultiIndex.zig:302-306): has_min_score == 0 means "null" -> (raw query length + 19) / 20.
 * max_results is used as given (the HTTP front end clamps to [1,100], src/server.zig:192;
 * the legacy front end passes 500, src/legacy.zig:194).  min_score_pct is any u32, as upstream (src/server.zig:189-193
 * clamps only limit and timeout): top * pct / 100 is computed in 64 bits and saturates at u32 max. */
typedef struct {
    uint32_t max_results;
    uint32_t min_score;
    uint32_t has_min_score;
    uint32_t min_score_pct;
} fpx_opts;

/* Totals the reference exports as fpindex_scanned_blocks_per_hash / _docs_per_hash
 * (src/FileSegment.zig:177-178, src/metrics.zig:93-101) plus device timing. */
typedef struct {
    uint64_t probes;            /* (unique query hash, file segment) pairs */
    uint64_t scanned_blocks;    /* sum of the reference's num_blocks counter (src/FileSegment.zig:154,171) */
    uint64_t scanned_docs;      /* sum of num_docs (:153,172), before supersession filtering */
    uint64_t hits;              /* postings accumulated after supersession filtering */
    uint64_t algorithmic_bytes; /* sum over visited blocks of that segment's block_size */
    uint64_t candidates;        /* (query, doc) pairs with score >= min_score */
    float    probe_kernel_ms;   /* HIP-event time of the MAIN probe kernel(s) of the call: k_probe_group (+ k_probe_direct) on direct-addressed
                                   segments, else k_probe_lean8 when the batch is large and the segments are dense 512-B ones, else k_probe */
    float    total_gpu_ms;      /* first launch -> last kernel of this call, on the call's stream */
    uint32_t probe_launches;    /* launches of the main probe kernel */
    uint32_t generic_iters;     /* block visits that needed the generic per-value decode path */
    uint64_t probe_kernel_bytes;/* algorithmic bytes of the blocks the main probe kernel visited itself */
    float    probe_aux_ms;      /* deferred / generic auxiliary probe passes (their blocks are in algorithmic_bytes) */
    uint32_t path_flags;        /* bit 0: the batch ran on the device-sized path (one host round trip; hit records binned by
                                   query as they are produced), bit 1: ... and needed the second round trip for queries with
                                   more candidates than slots.  Neither: the general path (first batch of a workspace,
                                   exchanges, anything the device-sized path handed back).  bit 2: direct-addressed segments
                                   were probed as a GROUP (k_probe_group), bit 3: ... whose records went straight into bins of a
                                   few queries, scored a bin per workgroup (k_score_bin), bit 4: the batch ran a step of a sharded protocol,
                                   bit 5: ... and hot hashes' lists reached the score kernel by reference ("hot_refs"),
                                   bit 6: the batch was searched a query per workgroup (k_search_query: no keys, no bins),
                                   bit 7: the snapshot was searched in TWO PARTS -- its one packed group (+ the memory segments) a query
                                   per workgroup, the file segments next to it by the pipeline, the two tables merged (a live index
                                   between merges: fpx_snapshot_create) */
    uint64_t probe_kernel_fetched_bytes; /* block bytes the main probe kernel really fetched, in 128-byte lines: a probe
                                   whose hash the segment's presence bits know to be absent counts as a visited block (as in
                                   the reference) without the block being read, and a block that is read costs two lines up
                                   front + the line of the matching docids, so this is <= probe_kernel_bytes.  Direct-addressed
                                   segments: the directory lines, word pieces and list heads read (64-byte units, halved) */
} fpx_stats;

/* ---- context ----------------------------------------------------------- */
/* One context per GPU / process.  device = HIP ordinal, -1 = current device. */
int  fpx_ctx_create(int device, fpx_ctx **out);
void fpx_ctx_destroy(fpx_ctx *ctx);
int  fpx_ctx_device(const fpx_ctx *ctx);     /* the HIP ordinal the context lives on */
/* The options of a context: everything that steers the library's behaviour.  Each falls back to the environment variable
 * FPX_<NAME> (the tests' and the A/B tools' way in), then to the default; a value below the option's smallest one (-1 for most,
 * -2 for "group_packed", where -1 is the explicit setting "decide by the data") = back to that fallback.
 * Storage forms -- read when a segment is created / a snapshot first holds it; fpx_segment_layout(), fpx_segment_layout_reason()
 * and fpx_snapshot_info() say what came of it:
 *   "direct"             1 | 0   dense segments trade their blocks for a direct-addressed form (default 1)
 *   "direct_min_items"   items from which a segment counts as dense (default 2^20)
 *   "fuse_min"           direct-addressed segments of one hash window that form a GROUP together: this many or more (default 2; 0: never)
 *   "group_packed"       1 | 0 | -1   a group's form: PACKED lines (one HBM line per query hash; dense groups) | directory + words |
 *                        by the group's density (default)
 *   "presence_min_items" items from which a segment in blocks gets presence bits and probe records (the lean kernel; default 2^20)
 *   "lean_head"          4: the lean kernel always fetches whole blocks (default 0: two lines where a block's head fits them)
 * Search paths -- read per batch:
 *   "query_wg"           1 | 0   a snapshot that is ONE packed group and nothing else (the resident index between merges) is searched a
 *                        QUERY PER WORKGROUP -- dedup, probe, count and floor in one kernel, the hit records never leaving the CU
 *                        (csrc/fpx_qsearch.hpp; default 1) | by the pipeline below like every other snapshot
 *   "fast"               1 | 0   the device-sized path (one host round trip per batch; default 1)
 *   "binned"             1 | 0   groups drop their records into bins of a few queries, scored a bin per workgroup (default 1)
 *   "rec32"              1 | 0   4-byte records in the bins where the doc ids leave room (default 1)
 *   "local_sort_max", "order_min_pairs"   pair counts that choose how a batch's keys are ordered
 *   "hot_refs"           1 | 0 | -1   the lists of HOT hashes (64+ docs) reach the score kernel by reference instead of a copy per query |
 *                        are copied into the bins | by the records the workspace's last batch brought (default)
 *   "lean_min"           probes from which block-form segments take the lean kernel (default 2^16)
 *   "sharded_workers"    worker threads per device of a sharded snapshot (1..16, default 3)
 * (FPX_SHARDED_RCCL alone stays with the process: whether librccl is loaded at all.) */
int  fpx_ctx_set_option(fpx_ctx *ctx, const char *name, int64_t value);
int  fpx_ctx_get_option(const fpx_ctx *ctx, const char *name, int64_t *value);   /* the value in force */
/* Device memory the library keeps for reuse on the context's GPU: the LINE buffer of the last group that was released (a group's
 * lines cost memory by the hash space -- 8.6 .. 137 GB -- and the runtime takes seconds to map and unmap that much; the next group
 * of the same size, e.g. the one fpx_segments_regroup builds after the next merge, takes the buffer over).  One buffer per device
 * at most; the library frees it by itself before any of its own allocations fails for lack of memory.  fpx_ctx_trim frees it
 * now (another process, or another library in this one, is to have the memory) and returns the bytes that went. */
uint64_t fpx_ctx_trim(fpx_ctx *ctx);
const char *fpx_strerror(int status);
/* last error text of the calling thread (valid until its next fpx call) */
const char *fpx_last_error(void);
int  fpx_version(void);

/* ---- segments ---------------------------------------------------------- */
/* Replaces: end of filefmt.readSegment (src/filefmt.zig:270-284) and
 * Index.mergeToFileSegment (src/Index.zig:961-983) -- "segment becomes resident".
 * blocks: num_blocks fixed-size blocks + the empty terminator block (src/filefmt.zig:9-10),
 * blocks_len = (num_blocks + 1) * block_size; block_index: max hash per block, LE u32.
 * doc_ids/doc_alive: the segment's `docs` map (alive or tombstone), any order.
 * Every id that occurs in the blocks must be listed in doc_ids (reference invariant). */
int fpx_segment_create_file(fpx_ctx *ctx,
                            const uint8_t *blocks, size_t blocks_len, uint32_t block_size,
                            const uint32_t *block_index, uint32_t num_blocks,
                            uint32_t min_doc_id, uint32_t max_doc_id, uint64_t commit_id,
                            const uint32_t *doc_ids, const uint8_t *doc_alive, uint32_t num_docs,
                            fpx_segment **out);

/* Replaces: MemorySegment.build's result (src/MemorySegment.zig:81-148, src/Index.zig:531-534).
 * items: u64 = hash << 32 | id (src/segment.zig:87-89), sorted ascending, duplicates kept. */
int fpx_segment_create_memory(fpx_ctx *ctx, const uint64_t *items, size_t num_items,
                              uint32_t min_doc_id, uint32_t max_doc_id, uint64_t commit_id,
                              const uint32_t *doc_ids, const uint8_t *doc_alive, uint32_t num_docs,
                              fpx_segment **out);

/* A segment whose postings live on ANOTHER GPU (segment sharding, one process per GPU):
 * only its identity and `docs` map are needed here, for supersession (hasNewerCommit,
 * src/Index.zig:133-149).  It contributes no hits on this device. */
int fpx_segment_create_remote(fpx_ctx *ctx, uint32_t min_doc_id, uint32_t max_doc_id, uint64_t commit_id,
                              const uint32_t *doc_ids, const uint8_t *doc_alive, uint32_t num_docs,
                              fpx_segment **out);

/* SharedPtr.acquire / release (src/shared_ptr.zig:38-85): HBM is freed on the last release. */
void fpx_segment_retain(fpx_segment *seg);
void fpx_segment_release(fpx_segment *seg);

/* introspection */
uint64_t fpx_segment_num_items(const fpx_segment *seg);     /* FileSegment/MemorySegment.getSize */
uint32_t fpx_segment_num_blocks(const fpx_segment *seg);
uint32_t fpx_segment_block_size(const fpx_segment *seg);
uint64_t fpx_segment_device_bytes(const fpx_segment *seg);
/* How a resident file segment is kept in HBM: 0 = its blocks, as in the file (src/filefmt.zig); 1 = DIRECT-ADDRESSED -- a dense
 * segment of >= 2^20 items (FPX_DIRECT_MIN_ITEMS; FPX_DIRECT=0 turns it off) trades its blocks for an exact
 * presence bitmap with a rank directory and doc lists (csrc/fpx_direct.hpp).  Searches, counters, downloads and merges do
 * not depend on the form: a download re-encodes the file's blocks byte for byte. */
int fpx_segment_layout(const fpx_segment *seg);
/* ... and WHY, in words (a static string): e.g. "blocks: not enough free HBM to build the group next to the members' blocks" -- a
 * host watches this after fpx_snapshot_create instead of finding out from its query rate (a segment that settled in its blocks is
 * searched several times slower than a column of a group). */
const char *fpx_segment_layout_reason(const fpx_segment *seg);
/* What the group of a grouped segment (layout 2) looks like: info[0..13] = columns in use, columns of a directory line (8 / 16),
 * HBM bytes of the whole group, of its lines (directory), of its words, of its lists (packed form: lists + overflowing words),
 * positions stored as inline doubles, this segment's column, first and last hash of the group's hash window, 1 = the PACKED
 * form (a dense group: 128-byte lines of 4 / 8 hash values with their words inside, csrc/fpx_pgroup.hpp), lines, lines whose
 * words overflow into `ext`, words there.  (Introspection for benchmarks and capacity planning.) */
int fpx_segment_group_info(const fpx_segment *seg, uint64_t *info, uint32_t n);
/* Rebuild the groups of an index's file segments (no counterpart in the reference: the GPU form of what its merge policy does
 * for the segment count, src/segment_merge_policy.zig -- here for the count of GROUPS).  Groups form when segments first meet in
 * a snapshot and are never changed: after checkpoints and merges an index holds several (the old group with the merged-away
 * members as dead columns, a new group or a lone direct-addressed segment per merge result), and every one costs a probe
 * launch and an HBM line per query hash.  Of `segments` (the file segments of the next snapshot, in its order) the first 16
 * that hold the whole hash space and are grouped, direct-addressed or waiting for their first snapshot become ONE new group:
 * grouped members' blocks are encoded again from their columns (byte for byte), and the group is built chunk by chunk as for
 * fresh segments.  Snapshots made before keep the old groups (which go with the last of them); the next fpx_snapshot_create
 * sees the new one.  *regrouped = members of the new group; 0 = nothing to gain (one group without dead columns already).
 * FPX_E_NOMEM: no room for the members' blocks + the new group next to the old ones -- nothing has changed.  Call it from the
 * thread that publishes snapshots (it is serialised with fpx_snapshot_create), after a merge, off the query path. */
int fpx_segments_regroup(fpx_ctx *ctx, fpx_segment *const *segments, uint32_t n, uint32_t *regrouped);
/* copy a resident file segment's blocks (+terminator) and block index back to the host */
int fpx_segment_download(const fpx_segment *seg, uint8_t *blocks, size_t blocks_cap,
                         uint32_t *block_index, uint32_t index_cap);

/* ---- snapshot ---------------------------------------------------------- */
/* Replaces: Index.createSnapshot + swapSnapshot (src/Index.zig:450-485).  `segs` is the
 * Segments snapshot order: file[] then memory[], oldest -> newest, commit_id strictly
 * ascending (src/Index.zig:36-41).  Builds the supersession tables.  Retains the segments.  A segment that is resident
 * on another context's device takes part with its docs map only (like fpx_segment_create_remote).
 * A snapshot of [one packed group without superseded docs] + [other file segments] -- a live index between merges -- is also laid out
 * in TWO PARTS that batches are searched through apart and merged (a doc lives in one segment; fpx_stats.path_flags bit 7): nothing
 * for the caller to do, a few hundred microseconds of snapshot creation. */
int  fpx_snapshot_create(fpx_ctx *ctx, fpx_segment *const *segs, uint32_t num_segs, fpx_snapshot **out);
/* acquireReader / IndexReader.deinit (src/Index.zig:430-434, :157-163) */
/* What fpx_snapshot_create made of the segments on this context: info[0..11] = file segments searched in their blocks by the
 * lean kernel, by the generic kernel, small ones searched in their decoded items, direct-addressed on their own, columns of
 * groups, groups, PACKED groups, memory segments, candidates that SETTLED in their blocks (no HBM for another form: see
 * fpx_segment_layout_reason), HBM bytes of the snapshot's segments, 1 = batches take the one-launch path (groups and nothing
 * else), block-form file segments in all. */
int  fpx_snapshot_info(const fpx_snapshot *snap, uint64_t *info, uint32_t n);
void fpx_snapshot_retain(fpx_snapshot *snap);
void fpx_snapshot_release(fpx_snapshot *snap);

/* ---- search ------------------------------------------------------------ */
/* Replaces: IndexReader.search(hashes, results) + results.getResults()
 * (src/Index.zig:170-177, src/common.zig:131-173) as called by MultiIndex.search
 * (src/MultiIndex.zig:287-330).  `hashes` is the raw query: unsorted, duplicates allowed,
 * not modified.  timeout_ms == 0 means unbounded (src/MultiIndex.zig:286,315).  With a deadline the calling thread polls
 * the call's stream instead of blocking on it; when the deadline passes it raises a cancel word that every kernel of the
 * call checks per workgroup (the GPU form of the zio.maybeYield cancel point, src/FileSegment.zig:144), the device drains
 * within ~0.1 ms and the call returns FPX_E_TIMEOUT with no partial results (src/MultiIndex.zig:319-322).
 * Writes min(*out_n, out_cap) results ordered by (score desc, id asc). */
int fpx_search(fpx_snapshot *snap, const uint32_t *hashes, uint32_t num_hashes,
               const fpx_opts *opts, uint32_t timeout_ms,
               fpx_result *out, uint32_t out_cap, uint32_t *out_n, fpx_stats *stats);

/* Batched form (what a host-side request coalescer drives; BASELINE configs use B = 1024 / 8192).
 * Query q is hashes[offsets[q] .. offsets[q+1]); opts[q] its options; results of query q are
 * written to out[q * out_cap ..] and their count to out_n[q]. */
int fpx_search_batch(fpx_snapshot *snap, const uint32_t *hashes, const uint64_t *offsets,
                     uint32_t num_queries, const fpx_opts *opts, uint32_t timeout_ms,
                     fpx_result *out, uint32_t out_cap, uint32_t *out_n, fpx_stats *stats);

/* fpx_search_batch + per-QUERY scan statistics.  The reference observes num_blocks and num_docs of every (hash, segment) walk
 * into fpindex_scanned_blocks_per_hash / fpindex_scanned_docs_per_hash (src/FileSegment.zig:177-178, src/metrics.zig:93-101); a
 * host that keeps feeding those per request needs more than a batch's totals: scanned_blocks_q[q] / scanned_docs_q[q] (each
 * [num_queries], either may be null) receive query q's sums over its unique hashes and the snapshot's FILE segments -- the same
 * quantities as fpx_stats.scanned_blocks / scanned_docs, which are their totals.  Costs one atomic per walk when asked for. */
int fpx_search_batch_stats(fpx_snapshot *snap, const uint32_t *hashes, const uint64_t *offsets,
                           uint32_t num_queries, const fpx_opts *opts, uint32_t timeout_ms,
                           fpx_result *out, uint32_t out_cap, uint32_t *out_n, fpx_stats *stats,
                           uint64_t *scanned_blocks_q, uint64_t *scanned_docs_q);

/* The reference's two per-(hash, segment) HISTOGRAMS, fed from a sample of the traffic.
 * Replaces: metrics.observeScannedDocsPerHash(num_docs) / observeScannedBlocksPerHash(num_blocks) at the end of every hash's
 * walk in FileSegment.search (src/FileSegment.zig:177-178), i.e. fpindex_scanned_docs_per_hash / fpindex_scanned_blocks_per_hash
 * with the bucket bounds of src/metrics.zig:9-10.  fpx_search_batch_stats gives the SUMS of those observations per query (exact
 * `_sum`); the buckets need every observation on its own, and the probe kernels answer a hash for sixteen segments at once.
 * This call REPLAYS the queries it is given for the statistics alone: every query is sorted and de-duplicated as
 * IndexReader.search does (src/Index.zig:170-172), and every unique hash is searched as a query of its own against each FILE
 * segment of the snapshot that is resident on the snapshot's context, alone (a one-segment snapshot: a column of a group with
 * the others masked out, a block-form segment in its blocks) -- so that the per-query statistics of fpx_search_batch_stats ARE
 * the (num_blocks, num_docs) the reference observes for that (hash, segment).  A hash-window slice observes the hashes of its
 * window only (the others are another rank's).  Memory segments observe nothing (src/MemorySegment.zig:44-54 has no metrics).
 * Meant for a sample -- one request in a few hundred --: the replay costs about as much as searching the queries once per
 * segment.  The observations are ADDED to *acc (zero it first, or keep it as the process's running histogram):
 *   docs_bucket[i]   observations v with bound[i-1] < v <= bound[i], bounds 1 2 3 5 10 50 100 500 1000, [9] = above 1000 (+Inf)
 *   blocks_bucket[i] bounds 1 2 3 5 10, [5] = above 10 (+Inf; the reference visits at most 4 blocks per hash)
 *   docs_sum / blocks_sum / count   the histograms' `_sum` and `_count` (count: the same for both)
 * Buckets are NOT cumulative; a Prometheus exporter adds them up. */
typedef struct {
    uint64_t docs_bucket[10];
    uint64_t blocks_bucket[6];
    uint64_t docs_sum, blocks_sum, count;
} fpx_scan_histograms;
int fpx_scan_histograms_observe(fpx_snapshot *snap, const uint32_t *hashes, const uint64_t *offsets, uint32_t num_queries,
                                uint32_t timeout_ms, fpx_scan_histograms *acc);
/* The same two histograms for ALL the traffic of a context, exact and free.  The direct-addressed forms' kernels hold every (hash,
 * segment) walk's (num_docs, num_blocks) explicitly -- absent | one doc | two docs inline | a list's header --, the block forms' kernels
 * have them when a walk ends (src/FileSegment.zig:171-178); all of them bucket the walks as they answer them; a context keeps
 * the RUNNING totals of every search that succeeded on it (any entry point, any thread; a batch that was redone on another path
 * counts once; a step of the fpx_shard_* protocols that the ranks redo for larger buffers is observed again).  *out receives the
 * totals since the context was created -- the process-wide histograms src/metrics.zig keeps --, in fpx_scan_histograms' layout.
 * *unbucketed (may be NULL): walks that were counted but not bucketed -- 0 now that the block forms' kernels bucket theirs too (it
 * counted the walks answered from BLOCKS: segments below "direct_min_items" between merges, "direct" = 0); kept for its callers. */
int fpx_ctx_scan_histograms(const fpx_ctx *ctx, fpx_scan_histograms *out, uint64_t *unbucketed);

/* Query batch already resident in HBM: what a host-side coalescer that keeps its staging buffers on the
 * device would hand over, and what bench.py times ("inputs resident in HBM when the timed region starts").
 * fpx_search_resident(snap, qb, ...) == fpx_search_batch(snap, <the arrays qb was created from>, ...). */
typedef struct fpx_query_batch fpx_query_batch;
int  fpx_query_batch_create(fpx_ctx *ctx, const uint32_t *hashes, const uint64_t *offsets, uint32_t num_queries,
                            const fpx_opts *opts, fpx_query_batch **out);
void fpx_query_batch_release(fpx_query_batch *qb);
int  fpx_search_resident(fpx_snapshot *snap, const fpx_query_batch *qb, uint32_t timeout_ms,
                         fpx_result *out, uint32_t out_cap, uint32_t *out_n, fpx_stats *stats);
int  fpx_search_resident_partial(fpx_snapshot *snap, const fpx_query_batch *qb, uint32_t timeout_ms,
                                 void *d_out, uint32_t out_cap, void *d_out_n, fpx_stats *stats);

/* Segment-sharded multi-GPU: stage 1 on every rank.  Same as fpx_search_batch but the
 * per-query tables stay in HBM (d_out: num_queries * out_cap fpx_result, d_out_n: num_queries u32,
 * both DEVICE pointers) and only the absolute min_score floor is applied, so the tables of all
 * ranks can be exchanged with an RCCL all-gather and merged by fpx_merge_partials. */
int fpx_search_batch_partial(fpx_snapshot *snap, const uint32_t *hashes, const uint64_t *offsets,
                             uint32_t num_queries, const fpx_opts *opts, uint32_t timeout_ms,
                             void *d_out, uint32_t out_cap, void *d_out_n, fpx_stats *stats);

/* Stage 2: merge `world` gathered partial tables (DEVICE pointers, rank-major:
 * d_parts[r][q][out_cap], d_counts[r][q]) into the final per-query top-k on the host
 * (relative min_score_pct cut-off anchored on the global best score, src/common.zig:162). */
int fpx_merge_partials(fpx_ctx *ctx, const void *d_parts, const void *d_counts, uint32_t world,
                       uint32_t num_queries, uint32_t part_cap, const fpx_opts *opts,
                       const uint64_t *offsets,
                       fpx_result *out, uint32_t out_cap, uint32_t *out_n);

/* ---- ONE process, several GPUs: segment sharding behind one call ------------------------------------------------------
 * The reference answers a search with one call from one process (IndexReader.search, src/Index.zig:170-177, on the
 * executors of src/main.zig:272-276).  A host that owns every GPU of a node keeps that shape: segments are made resident
 * on the device of the context they are created with (fpx_segment_create_*(ctx_k, ...), one context per GPU), a sharded
 * snapshot takes the whole segment list in snapshot order (as fpx_snapshot_create: file[] then memory[], commit ids
 * ascending) and builds one local snapshot per participating device -- that device's postings plus the docs maps of all
 * other segments, so that supersession (hasNewerCommit, src/Index.zig:133-149) stays exact --, and ONE call searches them
 * all: the per-device partial searches run concurrently on worker threads, the per-query tables [B][limit]{id, score} +
 * counts travel to the first context's device with hipMemcpyPeerAsync (xGMI; direct when peer access is available) and
 * are merged there (k-way merge, relative cut-off anchored on the global best score).  Results are identical to an
 * unsharded snapshot of the same segments.  The handle retains the local snapshots (hence the segments).
 * (The one-process-per-GPU form of the same protocol -- fpx_search_resident_partial, an RCCL all-gather issued by the
 * launcher, fpx_merge_partials -- is what bench.py --gpus N runs under torch.distributed.) */
typedef struct fpx_sharded_snapshot fpx_sharded_snapshot;
int  fpx_sharded_snapshot_create(fpx_segment *const *segs, uint32_t num_segs, fpx_sharded_snapshot **out);
/* the same with the root named: `root` merges the tables and answers for an index without a single segment (no results, as
 * the reference does).  When every participating context has a device of its own the tables travel with RCCL (grouped
 * ncclSend / ncclRecv over xGMI; librccl is loaded at run time, FPX_SHARDED_RCCL=0 turns it off), else with peer copies. */
int  fpx_sharded_snapshot_create_on(fpx_ctx *root, fpx_segment *const *segs, uint32_t num_segs, fpx_sharded_snapshot **out);
void fpx_sharded_snapshot_retain(fpx_sharded_snapshot *snap);
void fpx_sharded_snapshot_release(fpx_sharded_snapshot *snap);
uint32_t fpx_sharded_snapshot_num_devices(const fpx_sharded_snapshot *snap);   /* contexts that hold postings */
/* fpx_search / fpx_search_batch over a sharded snapshot: same arguments, same results, same errors.  In `stats` the
 * counters are summed over the devices and the times are the slowest device's. */
int  fpx_sharded_search(fpx_sharded_snapshot *snap, const uint32_t *hashes, uint32_t num_hashes,
                        const fpx_opts *opts, uint32_t timeout_ms,
                        fpx_result *out, uint32_t out_cap, uint32_t *out_n, fpx_stats *stats);
int  fpx_sharded_search_batch(fpx_sharded_snapshot *snap, const uint32_t *hashes, const uint64_t *offsets,
                              uint32_t num_queries, const fpx_opts *opts, uint32_t timeout_ms,
                              fpx_result *out, uint32_t out_cap, uint32_t *out_n, fpx_stats *stats);

/* The same one-call shape with the index sharded by HASH RANGE (the layout that scales): context k = rank k holds
 * window k -- hashes [k 2^32 / world, (k + 1) 2^32 / world) -- of ALL segments.
 *   fpx_segment_create_file_windows      one segment file -> its `world` window slices, slice k resident on ctxs[k] (the blocks that
 *                                        hold the window's hashes + 3 halo blocks, src/FileSegment.zig:25,153-174; the whole docs
 *                                        map with every slice).  Replaces the end of filefmt.readSegment (src/filefmt.zig:270-284).
 *   fpx_sharded_snapshot_create_windows  slices[k * num_segs + j] = slice k of segment j (snapshot order, commit ids ascending);
 *                                        rank k's slices become one group with its window.  world: 1, 2, 4 .. 64.  MEMORY segments
 *                                        (fpx_segment_create_memory on ctxs[k], one copy per rank: a live index publishes one with every
 *                                        update, src/Index.zig:515-587) take their place in the list like on one GPU; a rank looks up
 *                                        the keys of its window in them.  (A snapshot whose file segments are slices of ONE hash
 *                                        window answers for that window's hashes in its memory segments too, whichever entry point
 *                                        searches it: the ranks' answers add up.)
 * fpx_sharded_search(_batch) on such a snapshot runs the routed-key protocol behind the one call: 1 / world of the batch's hashes
 * goes to each device (H2D), the devices make the keys of their share and deal them to the windows' ranks (all-to-all #1: RCCL
 * grouped send / recv over xGMI when every context has a device of its own, peer copies otherwise), every rank probes the keys of
 * ITS window and drops the records into the batch's bins, the bins travel to the rank their queries came from (all-to-all #2)
 * and that rank writes their final results straight into the caller's rows.  Same results as an unsharded snapshot of the whole
 * segments.  Batches the bin protocol does not take (a score floor of 1 or 2 -- the legacy front end's options, src/legacy.zig:185-196 --,
 * queries of more than 2048 hashes) run the RECORD protocol behind the same call: every rank probes its window with the whole batch,
 * the hit records travel to the rank that counts their doc (doc & (world - 1)), the ranks' per-query tables are merged on rank 0. */
int fpx_segment_create_file_windows(fpx_ctx *const *ctxs, uint32_t world,
                                    const uint8_t *blocks, size_t blocks_len, uint32_t block_size,
                                    const uint32_t *block_index, uint32_t num_blocks,
                                    uint32_t min_doc_id, uint32_t max_doc_id, uint64_t commit_id,
                                    const uint32_t *doc_ids, const uint8_t *doc_alive, uint32_t num_docs,
                                    fpx_segment **out /* [world] */);
int fpx_sharded_snapshot_create_windows(fpx_ctx *const *ctxs, uint32_t world, fpx_segment *const *slices, uint32_t num_segs,
                                        fpx_sharded_snapshot **out);

/* ---- pinned host memory --------------------------------------------------------------------------------------------
 * fpx_search_batch copies the caller's hashes to HBM and the results back.  From ordinary (pageable) memory the runtime
 * stages those copies and the calling thread waits for them; from page-locked memory they are asynchronous DMA.  A host
 * that assembles its batches in buffers from fpx_host_alloc (the request coalescer's staging, the result arrays) gets the
 * resident-batch rate end to end.  Plain hipHostMalloc underneath; any other pinned memory (hipHostRegister) does as well. */
int  fpx_host_alloc(size_t bytes, void **out);
void fpx_host_free(void *p);

/* ---- synthetic index builder (benchmarks / tests; not part of the reference surface) --- */
/* Builds, entirely on the GPU, the file segment holding documents
 * [first_doc, first_doc + num_docs) x hashes_per_doc seeded hashes (definition in DESIGN.md,
 * identical to oracle/fpx_oracle.c:orc_synth_hash), sorted and block-encoded with the
 * reference's fill rule (src/block.zig:501-567, src/filefmt.zig:94-138). dist: 0 uniform, 1 hot-pool. */
int fpx_synth_segment(fpx_ctx *ctx, uint64_t seed, uint32_t first_doc, uint32_t num_docs,
                      uint32_t hashes_per_doc, int dist, uint32_t block_size, uint64_t commit_id,
                      fpx_segment **out);

/* ---- hash-range sharding of ONE segment across GPUs (second mode) ------------------------------------
 * When a single segment must be split (more GPUs than segments, or a segment larger than one GPU's HBM), it is cut by
 * hash range at block boundaries.  A slice holds the blocks it owns plus the MAX_BLOCKS_PER_HASH - 1 = 3 following halo
 * blocks, so that a hash's run of <= 4 blocks (src/FileSegment.zig:25,153-174) stays with the slice that owns its first
 * block; `block_index` is the matching sub-array.  Only hashes in (lo_excl, hi_incl] are probed in the slice:
 * lo_excl = max hash of the block before the slice's first block (has_lo = 0 for the first slice),
 * hi_incl = max hash of the slice's last OWNED block (has_hi = 0 for the last slice).  Every slice carries the whole
 * docs map of its segment. */
int fpx_segment_create_file_slice(fpx_ctx *ctx, const uint8_t *blocks, size_t blocks_len, uint32_t block_size,
                                  const uint32_t *block_index, uint32_t num_blocks,
                                  int has_lo, uint32_t lo_excl, int has_hi, uint32_t hi_incl,
                                  uint32_t min_doc_id, uint32_t max_doc_id, uint64_t commit_id,
                                  const uint32_t *doc_ids, const uint8_t *doc_alive, uint32_t num_docs,
                                  fpx_segment **out);

/* The same cut on the device: a hash-window slice of a RESIDENT file segment that is still in its blocks (no snapshot has held
 * it yet); the source stays as it is and can be released afterwards. */
int fpx_segment_slice(fpx_segment *seg, int has_lo, uint32_t lo_excl, int has_hi, uint32_t hi_incl, fpx_segment **out);

/* With slices a document's postings come from several GPUs, so its score is a true sum across ranks: the pipeline is
 * cut at the hit records.  Stage 1 (fpx_probe_resident) runs the probes of the local snapshot only and writes the
 * records (q << 32 | doc) to `d_records` (device memory, room for `records_cap`), grouped by destination rank
 * doc & (world - 1) (`world` a power of two); counts[r] = records for rank r.  If the buffer is too small the call fails
 * with FPX_E_INVAL after filling `counts`, so the caller can retry with room for their sum.  The caller exchanges the
 * groups (all-to-all over RCCL) and every rank feeds what it received to stage 2 (fpx_score_partial), which counts per
 * (query, doc), applies the absolute floor and emits the per-query tables fpx_search_resident_partial would -- exact,
 * because all records of a document land on one rank.  fpx_merge_partials then finishes as in segment sharding. */
int fpx_probe_resident(fpx_snapshot *snap, const fpx_query_batch *qb, uint32_t world, uint32_t timeout_ms,
                       void *d_records, uint64_t records_cap, uint64_t *counts /* [world] */, fpx_stats *stats);
int fpx_score_partial(fpx_ctx *ctx, const fpx_query_batch *qb, const void *d_records, uint64_t num_records,
                      uint32_t timeout_ms, void *d_out /* [B][out_cap] fpx_result */, uint32_t out_cap,
                      void *d_out_n /* [B] uint32 */);

/* ---- an index sharded by HASH RANGE over the GPUs of a node: the scalable multi-GPU shape ------------------
 * Every rank holds the SAME window of the hash space of ALL segments -- hash-window slices (fpx_segment_create_file_slice, or
 * fpx_segment_slice of a resident segment) with one window; fpx_snapshot_create puts them into a group with that window -- so
 * a rank makes, sorts and probes only the query hashes of its window: 1/N of the batch's work, where segment sharding leaves
 * every rank the whole batch.  A hash's walk is independent of every other hash (src/FileSegment.zig:143-176) and
 * SearchResults.incr is a keyed sum (src/common.zig:121-129), so a query's HIT RECORDS (q << 32 | doc) may be counted wherever
 * they are brought together: the batch's bins of 8 queries are dealt to the ranks in contiguous runs of
 * bpr = fpx_shard_bins_per_rank(num_queries, world), and rank r FINISHES the queries of bins [r bpr, (r + 1) bpr).
 *   fpx_shard_probe   the records of this rank's window, dropped straight into the batch's bins: d_send = [world * bpr][cell_cap]
 *                     8-byte CELLS (DEVICE memory), d_send_counts = [world * bpr] uint32.  The contents are opaque to the caller: a
 *                     cell holds one record, or -- where every doc id of the snapshot is below 2^29 -- two 4-byte ones
 *                     (doc << 3 | query-in-bin); bits 0..30 of a count = the bin's records, bit 31 = "4-byte records", which is
 *                     how the receiving rank reads every piece the way its sender wrote it.
 *                     FPX_E_AGAIN: a bin outgrew cell_cap -- *needed_cell_cap says what to allocate; retry.
 *                     FPX_E_INVAL for snapshots that hold anything but such groups: use fpx_probe_resident / fpx_score_partial.
 *   (the caller's all-to-all, e.g. RCCL: rows [r bpr, (r + 1) bpr) of d_send and of d_send_counts travel to rank r; fixed shapes)
 *   fpx_shard_score   d_recv = [world][bpr][cell_cap], d_recv_counts = [world][bpr] as received (piece s = what rank s sent): the
 *                     FINAL results of this rank's queries -- *first_query, *num_queries say which -- written to out[i * out_cap ..],
 *                     out_n[i] for i = query - first_query (host memory, room for bpr * 8 queries).  No table exchange, no merge.
 * Counters in `stats` (scanned blocks / docs / probes) are this rank's share: their sum over the ranks is the unsharded total.
 * One bin size for all ranks, without a collective of its own: every rank must run the exchange with the same cell_cap, and a
 * rank only learns its own need (FPX_E_AGAIN from fpx_shard_probe).  Such a rank still takes part in the step's exchange, with
 * EVERY count it sends set to FPX_SHARD_NEED_MARK | needed_cell_cap; as every rank receives a piece from every sender, each
 * fpx_shard_score of that step then returns FPX_E_AGAIN with the same *needed_cell_cap (the largest mark seen), and all ranks redo
 * the step with it.  Batches with a score floor of 1 or 2 are refused by fpx_shard_probe (FPX_E_INVAL): the record protocol
 * (fpx_probe_resident / fpx_score_partial) answers those. */
#define FPX_SHARD_NEED_MARK 0x40000000u
uint32_t fpx_shard_bins_per_rank(uint32_t num_queries, uint32_t world);
int fpx_shard_probe(fpx_snapshot *snap, const fpx_query_batch *qb, uint32_t world, uint32_t timeout_ms,
                    void *d_send, uint64_t cell_cap, void *d_send_counts, uint64_t *needed_cell_cap, fpx_stats *stats);
int fpx_shard_score(fpx_ctx *ctx, const fpx_query_batch *qb, uint32_t world, uint32_t rank, const void *d_recv, uint64_t cell_cap,
                    const void *d_recv_counts, uint32_t timeout_ms, fpx_result *out, uint32_t out_cap, uint32_t *out_n,
                    uint32_t *first_query, uint32_t *num_queries, uint64_t *needed_cell_cap /* may be null */);

/* The same protocol with the KEYS routed instead of the hashes replicated -- what scales: fpx_shard_probe wants the whole batch's
 * hashes on every rank (at N ranks every query hash crosses a link N - 1 times and every rank reads all of them).  Here rank r
 * uploads only ITS share of the batch -- `qb_share` holds exactly the queries it finishes, [r bpr 8, (r + 1) bpr 8) of the
 * batch of num_queries_global --, and
 *   fpx_shard_keys         makes their keys (dedupSorted where the keys are made; query numbers of the global batch) and deals them
 *                          to the ranks by the hash's window: d_keys_send = [world][key_cap] 8-byte keys, in (hash bucket, query)
 *                          order inside a slot, d_key_counts = [world] uint64 (both DEVICE memory).  FPX_E_AGAIN: a slot outgrew
 *                          key_cap -- *needed_key_cap says what to allocate
 *   (the caller's all-to-all #1: slot w and its count travel to rank w; ~1/8 of a query's hashes x 8 bytes per link)
 *   fpx_shard_probe_keys   the N slots this rank received (slot s = the keys of its window from source s) -> the batch's bins, exactly
 *                          as fpx_shard_probe leaves them (same d_send / d_send_counts, same FPX_E_AGAIN)
 *   (all-to-all #2: the bins, as above)
 *   fpx_shard_score_share  fpx_shard_score with the rank's share in place of the whole batch: the final results of ITS queries,
 *                          out[i] for i = query - first_query.
 * A query hash crosses one link once, as a key, whatever N; the results end on the rank the query came from.  world must be a
 * power of two (the windows are the top bits of the hash).  Snapshots that are not groups of slices alone, queries of more than
 * 2048 hashes and floors of 1 or 2: FPX_E_INVAL (the record protocol answers those). */
int fpx_shard_keys(fpx_ctx *ctx, const fpx_query_batch *qb_share, uint32_t world, uint32_t rank, uint32_t num_queries_global,
                   void *d_keys_send, uint64_t key_cap, void *d_key_counts, uint64_t *needed_key_cap);
int fpx_shard_probe_keys(fpx_snapshot *snap, const void *d_keys_recv, uint64_t key_cap, const void *d_key_counts_recv, uint32_t world,
                         uint32_t num_queries_global, uint32_t timeout_ms, void *d_send, uint64_t cell_cap, void *d_send_counts,
                         uint64_t *needed_cell_cap, fpx_stats *stats);
int fpx_shard_score_share(fpx_ctx *ctx, const fpx_query_batch *qb_share, uint32_t world, uint32_t rank, uint32_t num_queries_global,
                          const void *d_recv, uint64_t cell_cap, const void *d_recv_counts, uint32_t timeout_ms,
                          fpx_result *out, uint32_t out_cap, uint32_t *out_n, uint32_t *first_query, uint32_t *num_queries,
                          uint64_t *needed_cell_cap /* may be null */);

/* ---- device-side segment build and merge -------------------------------------------------------
 * fpx_segment_build: filefmt.writeBlocks + BlockEncoder (src/filefmt.zig:94-138, src/block.zig:438-567) run on the GPU
 * over caller-provided items (hash << 32 | id); the result is a resident FileSegment whose bytes equal what the
 * reference writer produces for the same sorted items.  sorted = 0: the items are sorted on the device first
 * (Item order, src/segment.zig:90-94).  Every item's id must lie in [min_doc_id, max_doc_id]. */
int fpx_segment_build(fpx_ctx *ctx, const uint64_t *items, size_t num_items, int sorted, uint32_t block_size,
                      uint32_t min_doc_id, uint32_t max_doc_id, uint64_t commit_id,
                      const uint32_t *doc_ids, const uint8_t *doc_alive, uint32_t num_docs, fpx_segment **out);

/* fpx_segment_merge: SegmentMerger.prepare + read/advance feeding writeSegment, i.e. Index.mergeToFileSegment
 * (src/segment_merger.zig:85-155, src/Index.zig:961-983) for checkpoints (memory sources) and merges (file sources).
 * `sources` are segments of `collection`, oldest first.  Docs with a newer commit in the collection are dropped with
 * their items; the others keep their alive/tombstone status; min/max_doc_id cover the kept docs; commit_id is the
 * smallest of the sources (SegmentInfo.merge, src/segment.zig:38-51; the caller tracks `merges`). */
int fpx_segment_merge(fpx_snapshot *collection, fpx_segment *const *sources, uint32_t num_sources, uint32_t block_size,
                      fpx_segment **out);

/* the docs map and id range of a segment (after a merge the caller needs them for the manifest / segment file) */
uint64_t fpx_segment_commit_id(const fpx_segment *seg);
uint32_t fpx_segment_min_doc_id(const fpx_segment *seg);
uint32_t fpx_segment_max_doc_id(const fpx_segment *seg);
uint32_t fpx_segment_num_docs(const fpx_segment *seg);
int fpx_segment_docs(const fpx_segment *seg, uint32_t *doc_ids, uint8_t *doc_alive, uint32_t cap);   /* ids ascending */

/* CRC-64/XZ over `len` bytes continuing from `crc` (start with 0): the checksum the reference stores in the
 * segment file footer over the data blocks (std.hash.crc.Crc64Xz, src/filefmt.zig:101,119,261-284).  Host code. */
uint64_t fpx_crc64_xz(uint64_t crc, const uint8_t *data, size_t len);

/* HBM streaming-read and random-block-read bandwidth of this device (GB/s), measured by trivial
 * kernels: the second denominator SURVEY 8(d) asks for next to the 8 TB/s spec peak. */
int fpx_measure_bandwidth(fpx_ctx *ctx, size_t bytes, uint32_t block_size, double *stream_gbs, double *random_gbs);

/* Calibration kernels for the memory-side performance counters (bench.py's rocprofv3 --pmc passes): `lanes` threads read, at
 * addresses that are a permutation of a `bytes`-sized buffer's lines / words (every line once, nothing served twice from a
 * cache), mode 0: one whole 128-byte line each (eight 16-byte loads: a directory line of k_probe_group); 1: one 16-byte piece
 * at a 4-byte-aligned address (a hash's words); 2: one aligned 64-byte half line; 3: a line and two pieces (round 3's mix);
 * 4: a line per eight lanes, read together; 5: mode 0 with non-temporal loads; 6: the first 16 bytes of a line per lane (what
 * k_probe_pgroup asks of its line).  *ms = the launch's HIP-event time.  The 128-byte lines the memory side must be asked
 * for are known exactly: `lanes` (modes 0, 2, 4, 5, 6), lanes x 35/32 (mode 1: 3 pieces in 32 straddle a line), lanes x
 * (1 + 2 x 35/32) (mode 3) -- on gfx950 every memory-side read is a 128-byte request, whatever the access width. */
int fpx_measure_access(fpx_ctx *ctx, size_t bytes, int mode, uint64_t lanes, double *ms);

#ifdef __cplusplus
}
#endif
#endif /* FPX_H */
