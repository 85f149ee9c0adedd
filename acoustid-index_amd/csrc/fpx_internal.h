// fpx_internal.h -- declarations shared by the libfpx translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/fpx.h"

namespace fpx {

// ---------------------------------------------------------------- device-visible layout
// One resident file segment as the kernels see it (mirrors the fields FileSegment.search reads,
// src/FileSegment.zig:33-48, plus two derived acceleration tables).
struct SegDesc {
    const uint8_t*  blocks;        // (num_blocks + 1) * block_size bytes, 512-B aligned
    const uint32_t* block_index;   // max hash per block (src/filefmt.zig:119)
    const uint32_t* bucket;        // [nbuckets + 1]: lower_bound(block_index, k << bucket_shift)
    const uint32_t* dead;          // sorted ids of this segment's docs that a newer segment mentions
    const uint32_t* cont;          // bit b: block b+1 starts with block b's last hash (a run may continue there)
    const uint32_t* dead_bits;     // bitmap of `dead` over [shadow_lo, shadow_hi] (bit d - shadow_lo), or null when that range is too wide
    // Segments of >= 2^20 items (the lean kernel's) carry PROBE RECORDS: one 64-byte record per 2^(present_shift + 8) hash
    // values -- everything phase 1 of k_probe_lean8 needs for a probe in ONE cache line and ONE level of loads:
    //   words 0..7   256 presence bits: bit i = some item of the segment has a hash in [(r << 8 | i) << present_shift, +2^present_shift)
    //                (the shift keeps <= 16 % of the bits set where the hash space allows -- 0.7 byte per item -- and is 0
    //                beyond 750 M items: 31 % set at 1.6 G).  A probe whose bit is clear is answered -- and accounted for as the
    //                reference accounts for it (one visited block unless h falls in the gap before it) -- without reading the block
    //   word 8       lo = lower_bound(block_index, first hash of the record), bits 30..31: block boundaries inside the record's
    //                span (hi - lo: 0, 1, 2 = more -> binary search), hi in word 14
    //   words 9..13  {max hash, first hash} of blocks lo and lo + 1, first hash of block lo + 2 (copies of `blockrec`)
    // (Before: a 512-MB bitmap, a bucket table and the block records apart -- three lines per probe in two dependent
    // levels; at batch 1024 those lines were 60 % of the kernel's HBM requests.)
    const uint32_t* proberec;      // 2^(24 - present_shift) records of 16 words, or null
    const uint2*    blockrec;      // [num_blocks + 3] {max hash, first hash} of each block (+ all-ones sentinels): what the lean
                                   // kernel needs of block_index, the header and the continuation bitmap in one 8-byte record
    // small segments (< 2^20 items) are also kept DECODED: sorted items + where each block starts among them
    const uint64_t* items;         // hash << 32 | doc, or null
    const uint32_t* bstart;        // [num_blocks + 1] item offset of each block
    // ... with what a KEY needs to find its hash among them in a couple of dependent loads (k_probe_small): sbucket[k] = the first item
    // whose hash is >= k << sshift ([2^(32 - sshift) + 1] entries: ~2 items per bucket), sfirst: bit i = item i is the first of its block
    // scode: two bits per 2^cshift hash values (eight such cells per item: seven keys in eight are answered by this ONE load) -- 0: the
    // cell holds items (look them up), 1: none, and it lies inside a block's hash range (an absent hash there costs the reference one
    // visited block), 2: none, in the gap between two blocks / outside them (no visit, src/FileSegment.zig:153,164)
    const uint32_t* sbucket;
    const uint32_t* sfirst;
    const uint32_t* scode;
    uint32_t sshift, cshift, num_items;
    uint32_t num_blocks;
    uint32_t block_size;
    uint32_t bucket_shift;         // bucket of hash h = h >> bucket_shift (32 -> one bucket)
    uint32_t min_doc_id;
    uint32_t num_dead;
    uint32_t shadow_lo, shadow_hi; // id range covered by `dead`
    // hash-range slice of a segment (second mode): only hashes in (own_lo, own_hi] are probed here
    uint32_t own_flags;            // bit 0: own_lo is set, bit 1: own_hi is set
    uint32_t own_lo, own_hi;
    uint32_t present_shift;
    // DIRECT-ADDRESSED segments (fpx_direct.hpp; Segment::direct): the blocks are gone, the postings are reached through the
    // exact presence bitmap -- see the layout there
    const uint32_t* drec;          // 2^24 records of 16 words: 256 position bits (hash present, or gap), rank base, prefix counts
    const uint32_t* primary;       // [set bits] doc - min_doc_id, or bit 31 | offset into `extras`, or 0xFFFFFFFF: a gap position
    const uint32_t* extras;        // doc lists of the hashes with several docs
    uint32_t first_hash, last_hash; // of the segment: a hash outside is absent and costs the reference no block visit (:153,164)
    uint32_t extras_shift;         // list offsets in `primary` count words (0) or pairs of words (1: more than 2^31 words of lists)
};

constexpr uint32_t FUSE_MAX = 16;          // columns of a group

// A GROUP of up to 16 direct-addressed segments whose postings are stored together, hash-major and segment-minor, behind one
// directory (layout: fpx_group.hpp; built by fpx_group.hip).  This is what k_probe_group gets, by value.
constexpr uint32_t GROUP_LINE_WORDS = 32;  // a line: 128 bytes
constexpr uint32_t GROUP_INLINE = 29;      // words a line holds itself (behind 64 position bits + 32 double flags); with more: 28 + the offset of the rest
constexpr uint32_t GROUP_CHUNK_LOG2 = 26;  // the group's `ext` arrays are allocated per chunk of 2^26 hash values
constexpr uint32_t GROUP_CHUNKS = 64;
struct GroupDesc {
    const uint32_t* lines;                 // the line of hash h: lines + ((h >> 5) - line0) * 2 ns words; PACKED form (fpx_pgroup.hpp):
                                           // lines + ((h >> hvl) - line0) * 32 words, hvl = 2 (16 columns) | 3 (8)
    const uint32_t* const* ext_tab;        // packed form: [nchunks] where each chunk's `ext` (overflowing words + lists) starts
    uint32_t line0;                        // first line held (a hash-window slice of the group; 0: the whole hash space)
    uint32_t chunk0, nchunks;              // packed form: first chunk held, chunks held
    uint32_t gmin;                         // packed form: the words hold doc - gmin, ONE base for all columns (min_doc[s] == gmin)
    uint32_t lo_all, hi_all;               // every ACTIVE column's [first_hash, last_hash] contains [lo_all, hi_all] (the usual hash: no per-column test)
    uint32_t nseg;                         // columns in use
    uint32_t active;                       // bit s: column s belongs to the snapshot being searched (a group outlives merged-away members)
    uint32_t any_dead;
    uint32_t win_lo, win_hi;               // hashes outside are not probed here (0, 0xFFFFFFFF: no window)
    uint32_t block_size;                   // of the members' blocks (one size per group): a visited block's algorithmic bytes (src/filefmt.zig:29-31,71)
    uint32_t min_doc[FUSE_MAX], first_hash[FUSE_MAX], last_hash[FUSE_MAX];
    uint32_t seg_index[FUSE_MAX];          // the column's descriptor in Snapshot::d_direct (supersession filter)
    uint32_t has_dead[FUSE_MAX];
};

// One resident memory segment (src/MemorySegment.zig:27-28).
struct MemDesc {
    const uint64_t* items;         // hash << 32 | id, sorted
    const uint32_t* dead;
    uint64_t num_items;
    uint32_t num_dead;
    uint32_t shadow_lo, shadow_hi;
    uint32_t win_lo, win_hi;       // the snapshot's hash window (its file segments are slices of one window: a rank of an index sharded by hash
                                   // range): the memory segments -- held whole by every rank -- answer for the window's hashes only
    uint32_t pad;
};

// k_probe_lean8 ends every workgroup with four statistics atomics.  Atomics on ONE cache line complete about every 12 ns
// (83 M/s), so 64 k workgroups x 4 cost 3 ms of serialised L2 atomic time on a 5.6-ms kernel; they go to
// LEAN_STAT_SETS copies of the slots on separate lines (workgroup i -> set i % LEAN_STAT_SETS), summed on the host.
constexpr uint32_t LEAN_STAT_SETS = 64;
constexpr uint32_t DEF_COUNT_STRIDE = 32;      // 32-bit words between the segments' deferred-list counts: a line each, for the same reason
// The scan histograms of the direct-addressed kernels (src/FileSegment.zig:177-178, buckets of src/metrics.zig:9-10): every (hash, segment)
// walk is ONE observation of (num_docs, num_blocks).  Nearly all observe (0 | 1 doc, 0 | 1 block) -- the first bucket of both histograms --,
// so the kernels count the OTHERS and the totals; the first buckets are what is left.  HIST_SLOTS 64-bit slots per set (a line), behind the
// LEAN_STAT_SETS x 8 statistics slots:
//   0..8   observations in docs bucket 1..9 (2 | 3 | 4-5 | 6-10 | 11-50 | 51-100 | 101-500 | 501-1000 | more)
//   9..11  observations in blocks bucket 1..3 (2 | 3 | 4-5 blocks; the reference stops after four)
//   12     observations in all, 13 their docs, 14 their blocks (the histograms' _count and _sum)
constexpr uint32_t HIST_SLOTS = 16;
constexpr uint32_t HIST_COUNT = 12, HIST_DOCS = 13, HIST_BLOCKS = 14;
constexpr size_t LEAN_STAT_WORDS = (size_t)LEAN_STAT_SETS * (8 + HIST_SLOTS) * 2;      // in 32-bit words ([SETS][8] u64 statistics, then [SETS][HIST_SLOTS] u64)

// per-batch counters living in device memory (one 64-bit word each)
enum Counter : int {
    CTR_HITS = 0,        // hit records appended (may exceed capacity -> rerun with a larger buffer)
    CTR_CANDS = 1,       // candidate records appended
    CTR_BLOCKS = 2,      // visited blocks
    CTR_DOCS = 3,        // matched docs before supersession filtering
    CTR_BYTES = 4,       // algorithmic bytes
    CTR_PROBES = 5,      // valid (unique hash, file segment) probes
    CTR_MAXSCORE = 6,
    CTR_GENERIC = 7,     // wave iterations of k_probe that took the generic (per-value) decode path    // largest score of any candidate (sizes the score field of the candidate key)
    CTR_LEAN_READS = 8,  // blocks k_probe_lean8 fetched (probes whose hash the presence bitmap knows to be absent read none)
    CTR_HEAVY = 9,       // queries k_score handed to its CLASSED launch
    CTR_CANCEL = 10,     // != 0: the search's deadline passed (copied off the host's cancel word by polling workgroups) -- every
                         // workgroup leaves at its next cancel point, the results are discarded (error.SearchTimeout)
    CTR_TOTAL = 11,      // device-sized path: hit records of the batch (sum of the queries' counts, written by k_l2_scan)
    CTR_BINFAIL = 12,    // k_score_bin: a bin met more distinct (query, doc) pairs than its table takes: the batch is redone on the general path
    CTR_PADS = 15,       // "no record" entries that pad the bins' reservations to whole sectors (BIN_ALIGN): the bins' fill counts include them
    CTR_SLOTCANDS = 14,  // candidates handed from k_score to k_finish through the queries' own slots (statistics)
    CTR_HIST = 16,       // [16..31]: the HIST_SLOTS histogram slots of a launch too small for the spread sets (LEAN_STAT_SETS)
    CTR_COUNT = 32       // [8..15]: the same statistics slots, written by k_probe_lean8 (ctr_off = 8)
};

// ---------------------------------------------------------------- host objects
struct Ctx;

// The superseded docs of a segment within one snapshot, on the host and in HBM (sorted list + bitmap).  Successive
// snapshots mostly find the same set for their older segments: the set is cached on the segment and shared.
struct DeadSet {
    int device = 0;
    std::vector<uint32_t> ids;
    uint32_t* d_list = nullptr;
    uint32_t* d_bits = nullptr;
    ~DeadSet()
    {
        if (d_list || d_bits) (void)hipSetDevice(device);
        if (d_list) (void)hipFree(d_list);
        if (d_bits) (void)hipFree(d_bits);
    }
};

// The per-segment direct-addressed arrays (fpx_direct.hpp).  Snapshots that probe the segment on its own (k_probe_direct) keep
// them alive; a segment that joins a group lets go of its reference.
struct DirectStore {
    int device = 0;
    uint32_t* drec = nullptr; uint32_t* primary = nullptr; uint32_t* extras = nullptr;
    ~DirectStore()
    {
        (void)hipSetDevice(device);
        if (drec) (void)hipFree(drec);
        if (primary) (void)hipFree(primary);
        if (extras) (void)hipFree(extras);
    }
};

// A device buffer backed PIECE BY PIECE (hipMemAddressReserve + hipMemCreate / hipMemMap): one contiguous range for the kernels, whose pieces can
// be taken and given back one at a time.  A dense segment's blocks live in one (their head goes back as the group that replaces them grows,
// chunk of the hash space by chunk: release_below), and so do a group's lines (mapped chunk by chunk as they are filled: map_range) -- the
// conversion of an index no longer needs its blocks AND its group side by side (268 of 288 GB for the 100 M index until round 5).
struct VmBuf {
    int device = 0;
    uint8_t* va = nullptr;
    size_t reserved = 0, piece = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;      // per piece
    std::vector<uint8_t> mapped;                               // per piece
    ~VmBuf();
    static bool supported(int device);                         // (FPX_VM=0 switches the whole mechanism off)
    int reserve(int device, size_t bytes, size_t piece_bytes); // the address range alone
    int map_range(size_t lo, size_t hi);                       // backs the pieces that cover [lo, hi) (those not backed yet); FPX_E_NOMEM: out of memory
    void release_below(size_t upto);                           // gives back the whole pieces below `upto`
    size_t mapped_bytes() const;
};

// Host side of a group (GroupDesc is its device view).  Member segments and the snapshots that search it share it; the HBM
// goes when the last of them does.  A member that was merged away stays as a dead column until the group is rebuilt.
struct Group {
    int device = 0;
    uint32_t ns = 16;                      // 8 or 16: a directory line is 2 ns words
    uint32_t nseg = 0;
    uint32_t line0 = 0; uint64_t nlines = 0;
    uint32_t chunk0 = 0, nchunks = 0;
    uint32_t win_lo = 0, win_hi = 0xFFFFFFFFu;
    uint32_t block_size = 512;             // the members' (one block size per group: the statistics' byte counts)
    uint32_t* d_lines = nullptr; size_t lines_alloc_bytes = 0;
    std::unique_ptr<VmBuf> vm_lines;       // the lines' backing when they were mapped chunk by chunk (d_lines = its range; else one hipMalloc)
    // ... and then the chunks' arrays are carved from ONE piece-backed range too: this runtime credits memory that hipMemRelease gives back only when
    // the whole address range goes (tools/vmm_check2.hip), so while the members' blocks are going back hipMalloc believes in less memory than there is
    std::unique_ptr<VmBuf> vm_ext; size_t vm_ext_used = 0;
    std::vector<uint32_t*> word_chunks, list_chunks;       // one pair per hash-space chunk (the lines hold their addresses)
    // the PACKED form (fpx_pgroup.hpp: a hash's words inside its line) of a dense group
    bool packed = false;
    uint32_t gmin = 0;                     // packed form: the one doc id base of the group's words
    std::vector<uint32_t*> ext_chunks;     // one per hash-space chunk: overflowing words + lists
    // what the chunks' arrays were allocated as: big ones on their own, small ones (a sparse group: a test's, a fresh index's) carved from
    // slabs of 64 MB -- 64 chunks x 2 arrays of a few KB each were 128 allocations per group, and freeing an allocation costs milliseconds
    std::vector<void*> chunk_allocs;
    uint8_t* slab_cur = nullptr; size_t slab_left = 0;
    uint32_t* chunk_alloc(size_t bytes);   // null: out of memory
    uint32_t** d_ext_tab = nullptr;        // the same addresses in device memory (GroupDesc::ext_tab)
    uint32_t min_doc[FUSE_MAX] = {}, first_hash[FUSE_MAX] = {}, last_hash[FUSE_MAX] = {};
    uint64_t device_bytes = 0, total_words = 0, total_list_words = 0, doubles = 0, overflow_lines = 0, overflow_words = 0;
    ~Group();
};

struct Segment {
    std::atomic<int> refs{1};
    Ctx* ctx = nullptr;
    int kind = 0;                  // 0 file, 1 memory, 2 remote (docs only)
    uint64_t commit_id = 0;
    uint32_t min_doc_id = 0, max_doc_id = 0;
    std::vector<uint32_t> doc_ids; // sorted ascending (alive flag irrelevant to search: tombstones supersede too)
    std::vector<uint8_t> doc_alive; // parallel to doc_ids; carried for merges (src/segment_merger.zig:112-118)
    // file
    uint8_t* d_blocks = nullptr; size_t blocks_len = 0; uint32_t block_size = 0;
    std::unique_ptr<VmBuf> vm_blocks;      // the blocks' backing when they are large (d_blocks = its range): their head can go back piece by piece
    bool blocks_lost = false;              // a group build gave part of the blocks back and then failed: the segment cannot be searched any more
    uint32_t* d_block_index = nullptr; uint32_t num_blocks = 0;
    uint32_t* d_bucket = nullptr; uint32_t bucket_shift = 32; uint32_t num_buckets = 1;
    uint32_t* d_cont = nullptr;    // continuation bitmap, (num_blocks + 31) / 32 + 1 words
    uint32_t* d_proberec = nullptr; uint2* d_blockrec = nullptr; uint32_t present_shift = 0;   // probe records, block records (see SegDesc)
    uint32_t head_lines = 4;       // 2: in EVERY block the header, the hashes and the docid control bytes end before byte 252, so
                                   // k_probe_lean8<2> may fetch two 128-B lines first and the matching docid bytes afterwards
    uint64_t* d_small_items = nullptr; uint32_t* d_bstart = nullptr;   // decoded copy of a small segment (see SegDesc)
    uint32_t* d_small_aux = nullptr; uint32_t small_shift = 0, small_cshift = 0;   // ... its bucket table [2^(32 - small_shift) + 1], the block-first bits of its items, the cells' codes (SegDesc)
    // direct-addressed form (fpx_direct.hpp): replaces the blocks of a dense segment; d_bstart (item offset of every block) and
    // d_block_index stay, so that the blocks can be written out again byte for byte (materialize_blocks)
    const char* why = "";          // why the segment is kept the way it is (fpx_segment_layout_reason): a static string
    bool candidate = false;        // direct_candidate() accepted it: a snapshot decides between a group, the direct form on its own, its blocks
    bool settled = false;          // ... and it settled in its blocks (no room for another form)
    bool direct = false;
    std::shared_ptr<DirectStore> dstore;   // owner of d_drec / d_primary / d_extras while the segment is direct-addressed on its own
    uint32_t* d_drec = nullptr; uint32_t* d_primary = nullptr; uint32_t* d_extras = nullptr;
    std::shared_ptr<Group> home; uint32_t col = 0;   // ... or, once grouped: the group that holds its postings, and its column there
    uint64_t num_distinct = 0, num_positions = 0, extras_words = 0;     // distinct hashes; set bits = hashes + gap positions; list words
    uint32_t first_hash = 0, last_hash = 0, extras_shift = 0;
    uint32_t own_flags = 0, own_lo = 0, own_hi = 0;   // hash window of a slice (see SegDesc)
    std::mutex dead_mu; std::shared_ptr<DeadSet> last_dead;   // the dead set of the latest snapshot that holds this segment
    uint64_t num_items = 0;
    // memory
    uint64_t* d_items = nullptr;
    uint64_t device_bytes = 0;
};

struct Snapshot {
    std::atomic<int> refs{1};
    Ctx* ctx = nullptr;
    std::vector<Segment*> segs;          // retained
    std::vector<SegDesc> h_file;         // host copies of the descriptors
    std::vector<MemDesc> h_mem;
    SegDesc* d_file = nullptr; uint32_t n_file = 0;
    // file segments split by the kernel that suits them: dense 512-B segments (lean) and the rest (generic)
    SegDesc* d_lean = nullptr; uint32_t n_lean = 0;
    uint32_t n_lean2 = 0;                // the first n_lean2 of them qualify for the partial block fetch (Segment::head_lines == 2)
    SegDesc* d_gen = nullptr; uint32_t n_gen = 0; bool gen_all_512 = true;
    SegDesc* d_small = nullptr; uint32_t n_small = 0;     // small segments searched in their decoded items
    std::vector<SegDesc> h_direct;       // direct-addressed segments: not part of h_file / n_file
    SegDesc* d_direct = nullptr; uint32_t n_direct = 0;
    // ... of which the grouped ones are probed group by group (k_probe_group) and the rest (n_solo) one by one (k_probe_direct)
    std::vector<std::shared_ptr<Group>> groups;                  // grouped segments: one k_probe_group launch per group
    std::vector<GroupDesc> h_group; uint32_t n_group = 0;
    uint32_t max_doc_declared = 0;       // the largest doc id the segments' headers declare (sizes the bins' records, fpx_partition.hpp)
    uint32_t rec32_refused = 0;          // a posting exceeded it: this snapshot's bins hold wide records from then on
    uint32_t qs_skip = 0;                // batches that stay off the one-workgroup-per-query path (fpx_qsearch.hpp) after a query's records outgrew its LDS array
    std::vector<std::shared_ptr<DirectStore>> solo_stores;       // the arrays d_solo points into
    SegDesc* d_solo = nullptr; uint32_t n_solo = 0;
    Snapshot* part[2] = {nullptr, nullptr};   // [0]: its ONE packed group + the memory segments, [1]: its other file segments -- searched apart, tables merged (fpx_snapshot_create); or none
    uint32_t max_small_blocks = 0;
    MemDesc* d_mem = nullptr; uint32_t n_mem = 0;
    // ... and ONE table of all memory segments' LIVE postings (superseded docs dropped), sorted by hash, behind a 2^20-entry bucket
    // table: a query hash is looked up once for all memory segments, in any key order (fpx_probe_small.hpp: k_probe_memtab)
    uint64_t* d_memtab = nullptr; uint32_t* d_membucket = nullptr; uint64_t n_memtab = 0;
    uint32_t* d_membits = nullptr;         // one bit per 256 hash values: the table holds a posting there (k_probe_memtab's first load)
    uint64_t mem_items = 0;                // items of all memory segments together (0: nothing to look up, table or not)
    std::vector<std::shared_ptr<DeadSet>> dead_sets;   // shared with the segments' caches
    uint32_t max_block_size = 0;
    bool all_512 = true;                 // every file segment uses 512-B blocks (the only size the reference writes)
};

// counters [CTR_COUNT] | result count | up to 1024 results: the single-query fast path returns all of it in one copy
constexpr size_t COUNTERS_BYTES = (CTR_COUNT + 1) * sizeof(unsigned long long) + 1024 * sizeof(fpx_result);

constexpr size_t STAGE_BYTES = 64 * 1024;

// Pooled per-call device workspace (analogue of SearchResultsPool, src/common.zig:186-300).
struct Workspace {
    hipStream_t stream = nullptr;
    hipEvent_t ev_begin = nullptr, ev_probe0 = nullptr, ev_probe1 = nullptr, ev_probe2 = nullptr, ev_end = nullptr;
    // device buffers (capacity in elements)
    uint32_t* d_hashes = nullptr; size_t cap_hashes = 0;     // raw concatenated query hashes
    uint64_t* d_offsets = nullptr; size_t cap_queries = 0;   // [B+1]
    uint32_t* d_opts = nullptr;                               // [B][4]: max_results, min_score, pct, raw_len
    uint64_t* d_keys[2] = {nullptr, nullptr}; size_t cap_keys = 0;    // (hash, q) pairs
    uint64_t* d_hits[2] = {nullptr, nullptr}; size_t cap_hits = 0;    // (q, doc) records
    uint64_t* d_cands[2] = {nullptr, nullptr}; size_t cap_cands = 0;  // candidate keys
    void* d_temp = nullptr; size_t cap_temp = 0;              // radix sort temp
    uint64_t* d_qrange = nullptr; size_t cap_qrange = 0;      // [B][2] begin/end of each query's hit records
    uint64_t* d_qcand = nullptr; size_t cap_qcand = 0;        // [B][QCAND_SLOTS] candidate keys, then [B] u32 counts
    unsigned long long* d_counters = nullptr;                 // [CTR_COUNT]
    uint32_t* d_def_list = nullptr; size_t cap_def = 0;       // deferred probes of the lean kernel [n_file][def_cap]
    unsigned int* d_def_count = nullptr; unsigned int* h_def_count = nullptr; size_t cap_def_segs = 0;
    fpx_result* d_out = nullptr; uint32_t* d_out_n = nullptr; size_t cap_out = 0; // [B*cap], [B]
    // device-sized path (fpx_partition.hpp): [MAX_BINS * BIN_STRIDE bin fill counters | cap_binq per-query counts], scatter cursors
    uint32_t* d_binq = nullptr; unsigned long long* d_qcursor = nullptr; size_t cap_binq = 0; uint32_t* h_bins = nullptr;
    // results leave through pinned staging: [B] counts then [B * out_cap] results.  A copy into the caller's (pageable) arrays
    // straight from the device blocks the host until the stream gets there and costs ~100 us of driver time per call; a
    // copy into pinned memory is asynchronous, and the host moves the bytes on after the batch's synchronisation
    uint8_t* h_out = nullptr; size_t cap_h_out = 0;
    uint4* d_refs = nullptr; size_t cap_refs = 0;                     // hot lists by reference: [bins][ref_cap] entries (ProbeArgs::refs)
    uint32_t* d_kocnt = nullptr; size_t cap_kocnt = 0;                // the key order's count table [B][buckets] + totals + the key count (fpx_keyorder.hpp)
    unsigned long long* d_qstats = nullptr; size_t cap_qstats = 0;    // per-query scan statistics (blocks | docs << 32), when asked for
    uint32_t* d_cells = nullptr; uint32_t* h_cells = nullptr; size_t cap_cells = 0;   // fpx_shard_probe: the cells' fill counters + statistics slots
    // a large batch handed over in host memory and searched a query per workgroup (fpx_qsearch.hpp) is uploaded in UP_CHUNKS pieces on a stream of
    // its own; the kernel of piece c waits for ev_chunk[c] only: the upload of piece c + 1 travels under it
    static constexpr uint32_t UP_CHUNKS = 4;
    hipStream_t copy_stream = nullptr; hipEvent_t ev_chunk[UP_CHUNKS] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<uint32_t> h_opts;         // the batch's options as the kernels read them (kept here: their copy is not waited for)
    uint32_t hint_def = 0;                // longest deferred list of the last batch (sizes the deferred pass's grid)
    uint64_t hint_misc = 0;               // records the last binned batch left in the misc buffer (sizes k_bin's grid)
    uint64_t hint_P = 0, hint_H = 0;      // pairs and hit records of the last batch this workspace ran (sizes the next one)
    uint32_t fast_penalty = 0;            // batches left before the device-sized path is tried again after it had to be redone
    fpx_result* d_parts = nullptr; size_t cap_parts = 0;       // a snapshot in two parts: their tables [2][B][cap] and counts [2][B] (search_parts)
    uint32_t* d_parts_n = nullptr; size_t cap_parts_n = 0;
    uint64_t batch_hist[HIST_SLOTS + 1] = {};  // the scan histogram slots of the batch run_batch just finished, [HIST_SLOTS]: walks answered from blocks (search_split adds them to the context's)
    // pinned host staging
    unsigned long long* h_counters = nullptr;
    // one small query travels in ONE pinned copy: [offsets 2 x u64 | opts 4 x u32 | hashes]
    uint8_t* h_stage = nullptr; uint8_t* d_stage = nullptr;      // pinned + its device-mapped address
    unsigned long long* d_ret = nullptr;                          // device-mapped address of h_counters
    // the deadline's cancel word: pinned host memory the waiting host thread sets, read by the kernels through its mapping
    uint32_t* h_cancel = nullptr; const uint32_t* d_cancel = nullptr;
};

// A query batch already uploaded to HBM (fpx_query_batch_create): the timed region of a resident
// search starts with the inputs in device memory.
struct QueryBatch {
    Ctx* ctx = nullptr;
    uint32_t B = 0;
    uint32_t* d_hashes = nullptr;
    uint64_t* d_offsets = nullptr;       // [B+1], absolute
    uint32_t* d_opts = nullptr;          // [B][4]
    std::vector<uint64_t> offsets;       // host copy
    std::vector<fpx_opts> opts;
};

// Everything that steers the library's behaviour is an OPTION OF A CONTEXT (fpx_ctx_set_option / fpx_ctx_get_option): the value the
// context was given, else the environment variable FPX_<NAME> (the tests' and the A/B tools' way in), else the built-in default.
// Thresholds that decide a storage form or a kernel path belong to the index, not to the process's environment.
enum CtxOpt : int {
    OPT_DIRECT, OPT_DIRECT_MIN_ITEMS, OPT_FUSE_MIN, OPT_GROUP_PACKED,            // storage forms (read when a segment is created / first held)
    OPT_PRESENCE_MIN_ITEMS, OPT_LEAN_HEAD,
    OPT_FAST, OPT_BINNED, OPT_REC32,                                             // search paths (read per batch)
    OPT_LOCAL_SORT_MAX, OPT_ORDER_MIN_PAIRS, OPT_LEAN_MIN,
    OPT_SHARDED_WORKERS,
    OPT_HOT_REFS,
    OPT_QUERY_WG,
    OPT_COUNT
};
constexpr int64_t OPT_UNSET = -2;

struct Ctx {
    int device = 0;
    std::atomic<int64_t> opts[OPT_COUNT];
    Ctx() { for (auto& o : opts) o.store(OPT_UNSET, std::memory_order_relaxed); for (auto& h : scan_hist) h.store(0, std::memory_order_relaxed); }
    std::mutex mu;
    std::mutex group_mu;                  // serialises the grouping of segments (fpx_snapshot_create, fpx_segments_group)
    std::vector<Workspace*> free_ws;
    std::atomic<int> live_ws{0};
    std::atomic<int> qs_running{0};       // batches between the launch of their k_search_query and its end (run_batch: a batch that is alone is run differently)
    // the running scan histograms of the walks the probe kernels answered (fpx_ctx_scan_histograms), in HIST_SLOTS slots;
    // [HIST_SLOTS]: (hash, segment) walks that were counted but not bucketed (none: every probe kernel buckets its walks)
    std::atomic<uint64_t> scan_hist[HIST_SLOTS + 1];
};
void ctx_hist_add(Ctx* c, const uint64_t* slots, uint64_t unbucketed);     // (a batch's slots, after it has succeeded)

// the options of a context (fpx_ctx_set_option), falling back to the environment and the defaults
int64_t ctx_opt(const Ctx* c, CtxOpt o);   // the value in force (c may be null: environment, then default)
bool ctx_direct_enabled(const Ctx* c);
uint64_t ctx_direct_min_items(const Ctx* c);
uint32_t ctx_fuse_min(const Ctx* c);
int ctx_group_packed(const Ctx* c);       // -1: decided per group by its density
void set_error(const char* fmt, ...);
int  hip_fail(hipError_t e, const char* what);
// ---- device memory.  A group's LINES cost memory by the hash space (8.6 .. 137 GB), and the runtime takes SECONDS to map and to unmap an
// allocation of that size (measured on an MI355X: hipMalloc of 137 GB 5.3 s, hipFree of 64 GB 1.9 s -- an index that regroups after every
// merge pays that each time, a test suite of small indexes a thousand times).  The line buffer a group leaves behind stays with its DEVICE
// -- one buffer at most -- and the next group of exactly that size takes it over; a group of any other size frees it first, and so does
// every allocation of the library that runs out of memory (dmalloc), so the process never holds more than it held at its peak.
// the cached buffer if it holds `bytes` and no more than `max_bytes` (its size in *got; any other is freed), else null
void*  line_pool_take(int device, size_t bytes, size_t max_bytes, size_t* got);
void   line_pool_put(int device, void* p, size_t bytes);     // p becomes the cached buffer (what it replaces is freed); small ones are freed at once
size_t line_pool_flush(int device);                          // frees the cached buffer; the bytes that went (device < 0: of every device)
size_t line_pool_bytes(int device);
hipError_t dmalloc_raw(void** p, size_t bytes);              // hipMalloc on the current device; out of memory: line_pool_flush + once more
template <class T> inline hipError_t dmalloc(T** p, size_t bytes) { return dmalloc_raw(reinterpret_cast<void**>(p), bytes); }
hipError_t mem_info(size_t* free_b, size_t* total_b);        // hipMemGetInfo, the current device's cached buffer counted as free
// A group is built from 64 chunks x up to 16 members' pieces, and every piece used to take fourteen allocations of its own (and give
// them back: the runtime unmaps -- and the driver wipes -- what is freed): 14 000 hipMalloc / hipFree pairs for the 100 M index, 8 000 for
// a test's group of nine tiny segments.  DevArena: ONE allocation the pieces of a chunk carve their arrays from, rewound after every
// chunk.  What does not fit is allocated the old way and the arena grows at its next rewind (the first chunk sizes it).
struct DevArena {
    uint8_t* base = nullptr;
    size_t cap = 0, used = 0, missed = 0;          // missed: bytes asked for since the last rewind that did not fit
    const char* name = "arena";
    // FPX_ARENA_GUARD=<bytes> (a debugging aid): that many bytes of 0xA5 behind every allocation, checked at the rewind -- a kernel that
    // writes past the end of its buffer used to hit an allocation's padding, here it would hit its neighbour
    std::vector<std::pair<size_t, size_t>> guards; // (offset of the guard, bytes of the allocation before it)
    static size_t guard_bytes();
    void check_guards();
    ~DevArena() { check_guards(); if (base) (void)hipFree(base); }
    void* take(size_t bytes)                       // null: does not fit (the caller allocates on its own)
    {
        const size_t g = guard_bytes();
        const size_t body = (bytes + 255u) & ~(size_t)255u, need = body + g;
        if (used + need > cap) { missed += need; return nullptr; }
        void* p = base + used;
        if (g) { (void)hipMemset(base + used + body, 0xA5, g); guards.emplace_back(used + body, bytes); }
        used += need;
        return p;
    }
    void rewind()                                  // (nothing that uses the arena's memory is under way: the caller has waited)
    {
        check_guards();
        if (missed) {
            const size_t want = (used + missed) * 5 / 4 + ((size_t)1 << 20);
            if (base) (void)hipFree(base);
            base = nullptr; cap = 0;
            if (dmalloc(&base, want) == hipSuccess) cap = want; else { (void)hipGetLastError(); base = nullptr; }
        }
        used = 0; missed = 0;
    }
};
// Two of them while a group is built: a chunk's OUTPUTS (the members' pieces: alive until the chunk's lines are filled) and a piece's
// SCRATCH (decoded items, counts, bases: rewound piece by piece, so that sixteen members' scratch never adds up)
extern thread_local DevArena* tl_arena;                      // outputs of the chunk being built on this thread (null: none)
extern thread_local DevArena* tl_scratch;                    // scratch of the piece being built
void* arena_take(size_t bytes);                              // from tl_arena, or null
void* scratch_take(size_t bytes);                            // from tl_scratch, or null
#define FPX_HIP(expr)                                              \
    do {                                                           \
        hipError_t _e = (expr);                                    \
        if (_e != hipSuccess) return ::fpx::hip_fail(_e, #expr);   \
    } while (0)

Workspace* ws_acquire(Ctx* ctx);
void ws_release(Ctx* ctx, Workspace* ws);
void ws_destroy(Workspace* ws);

// fpx_sort.hip
size_t sort_u64_temp_bytes(size_t n, unsigned begin_bit, unsigned end_bit);
hipError_t sort_u64(void* temp, size_t temp_bytes, uint64_t* buf0, uint64_t* buf1, size_t n,
                    unsigned begin_bit, unsigned end_bit, hipStream_t stream, int* result_in);

size_t select_u64_temp_bytes(size_t n);
hipError_t select_u64(void* temp, size_t temp_bytes, const uint64_t* in, const uint8_t* flags, uint64_t* out,
                      unsigned long long* d_count, size_t n, hipStream_t stream);

// fpx_search.hip
int build_bucket_table(Segment* seg, hipStream_t stream);
// fpx_api.hip: a file segment's block buffer -- piece-backed from BLOCKS_VM_MIN bytes on where the runtime can, one allocation otherwise
hipError_t blocks_alloc(Segment* s, size_t bytes);
void blocks_free(Segment* s);
int query_batch_create_impl(Ctx* ctx, const uint32_t* hashes, const uint64_t* offsets, uint32_t B,
                            const fpx_opts* opts, QueryBatch** out);
void query_batch_free(QueryBatch* qb);
int search_batch_impl(Snapshot* snap, const QueryBatch* resident, const uint32_t* hashes, const uint64_t* offsets, uint32_t B,
                      const fpx_opts* opts, uint32_t timeout_ms, bool partial,
                      fpx_result* out, uint32_t out_cap, uint32_t* out_n,   // host (final) or device (partial)
                      fpx_stats* stats, uint64_t* q_blocks = nullptr, uint64_t* q_docs = nullptr);   // per-query scanned blocks / docs (host, [B]) or null
// the two halves of a partial search, for exchanging hit records between them (hash-range sharding)
int probe_records_impl(Snapshot* snap, const QueryBatch* qb, uint32_t world, uint32_t timeout_ms,
                       uint64_t* d_records, uint64_t records_cap, uint64_t* counts, fpx_stats* stats);
int score_records_impl(Ctx* ctx, const QueryBatch* qb, const uint64_t* d_records, uint64_t num_records, uint32_t timeout_ms,
                       fpx_result* d_out, uint32_t out_cap, uint32_t* d_out_n);
uint64_t group_bytes_lower_bound(const Ctx* ctx, Segment* const* segs, uint32_t k);
constexpr uint32_t SHARD_DEDUP_MAX = 2048;      // = DEDUP_MAX (fpx_kernels_common.hpp): the routed keys' queries hold at most this many hashes
int shard_bins_per_rank(uint32_t B, uint32_t world);
int shard_probe_impl(Snapshot* snap, const QueryBatch* qb, uint32_t world, uint32_t timeout_ms,
                     uint64_t* d_send, uint64_t cell_cap, uint32_t* d_send_counts, uint64_t* needed_cell_cap, fpx_stats* stats);
int shard_score_impl(Ctx* ctx, const QueryBatch* qb, uint32_t world, uint32_t rank, const uint64_t* d_recv, uint64_t cell_cap, const uint32_t* d_recv_counts,
                     uint32_t timeout_ms, fpx_result* out, uint32_t out_cap, uint32_t* out_n, uint32_t* first_query, uint32_t* num_queries,
                     uint64_t* needed_cell_cap = nullptr, uint32_t B_global = 0);
int shard_keys_impl(Ctx* ctx, const QueryBatch* qb, uint32_t world, uint32_t rank, uint32_t B_global,
                    uint64_t* d_keys_send, uint64_t key_cap, unsigned long long* d_key_counts, uint64_t* needed_key_cap);
int shard_probe_keys_impl(Snapshot* snap, const uint64_t* d_keys_recv, uint64_t key_cap, const unsigned long long* d_key_counts, uint32_t world, uint32_t B_global,
                          uint32_t timeout_ms, uint64_t* d_send, uint64_t cell_cap, uint32_t* d_send_counts, uint64_t* needed_cell_cap, fpx_stats* stats);
int merge_partials_impl(Ctx* ctx, const void* d_parts, const void* d_counts, uint32_t world,
                        uint32_t B, uint32_t part_cap, const fpx_opts* opts, const uint64_t* offsets,
                        fpx_result* out, uint32_t out_cap, uint32_t* out_n);
int build_memtab(Snapshot* sn);            // fills Snapshot::d_memtab / d_membucket from d_mem (fpx_search.hip)
int measure_bandwidth_impl(Ctx* ctx, size_t bytes, uint32_t block_size, double* stream_gbs, double* random_gbs);
int measure_access_impl(Ctx* ctx, size_t bytes, int mode, uint64_t lanes, double* ms_out);

// fpx_api.hip
int finish_file_segment(Segment* seg);   // bucket table + item count for blocks already in HBM

// fpx_build.hip
int synth_segment_impl(Ctx* ctx, uint64_t seed, uint32_t first_doc, uint32_t num_docs, uint32_t H, int dist,
                       uint32_t block_size, uint64_t commit_id, Segment** out);
// `s` arrives with its docs set; on failure the caller frees it
int segment_build_impl(Ctx* ctx, const uint64_t* items_host, uint64_t n, bool sorted, uint32_t block_size,
                       uint32_t min_doc_id, uint32_t max_doc_id, uint64_t commit_id, Segment* s);
// exclusive scan of n 32-bit counts into 64-bit offsets (+ their total), one workgroup
int scan_counts_u32(const uint32_t* counts, uint64_t n, uint64_t* offsets, uint64_t* total, hipStream_t st);
int decode_small_segment(Segment* s);     // fills d_small_items / d_bstart of a resident file segment
int build_presence(Segment* s);           // fills d_blockrec and d_proberec (both required by the lean kernel) of a resident file segment
struct HashRange { uint32_t lo, hi, rec0, whole; };      // hashes [lo, hi] into records counted from rec0; whole: the blocks are the whole segment
// the direct-addressed arrays of a hash range of a segment (fpx_direct.hpp: records of 256 hash values, primary, extras)
struct DirectPiece {
    uint32_t* drec = nullptr; uint32_t* primary = nullptr; uint32_t* extras = nullptr;
    bool own_drec = true, own_primary = true, own_extras = true;      // false: carved from the builder's arena (DevArena)
    uint32_t nrec = 0, xshift = 0;
    uint64_t distinct = 0, positions = 0, extras_words = 0;
    void release();
};
int direct_candidate(Segment* s, bool* ok);   // may this resident file segment become direct-addressed?  (fills d_bstart, first / last hash)
int build_direct_piece(const Segment* s, uint32_t b0, uint32_t nbl, HashRange hr, uint32_t nrec, bool has_prev, uint32_t prev_last, DirectPiece* out);
int build_direct(Segment* s);             // turns a candidate into its direct-addressed form on its own (or leaves it as it is)
void free_block_form(Segment* s);         // the blocks, bucket table and continuation bitmap of a segment that does not need them any more
// the blocks (+ terminator block + 16 B) of a direct-addressed segment, re-encoded into a fresh device buffer the caller frees
int materialize_blocks(const Segment* s, uint8_t** d_blocks_out);
// its items (hash << 32 | doc, sorted) into `items` [num_items]
int materialize_items(const Segment* s, uint64_t* items, hipStream_t st);
// fpx_group.hip
// moves `k` direct-addressed segments of one context (2..16; 1 with FPX_FUSE_MIN=1) into a new group and frees their own arrays
int group_segments(Ctx* ctx, Segment* const* segs, uint32_t k, std::shared_ptr<Group>* out);
int group_column_items(const Segment* s, uint64_t* items, hipStream_t st);      // the items of a grouped segment, sorted (downloads, merges)
struct MergeSource { const Segment* seg; std::vector<uint32_t> dead; };   // dead = skip_docs, sorted
int segment_merge_device(Ctx* ctx, const std::vector<MergeSource>& srcs, uint32_t block_size, uint32_t min_doc_id, Segment* s);

}  // namespace fpx
