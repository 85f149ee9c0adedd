/*
 * fpx_oracle.c -- CPU oracle (TEST INFRASTRUCTURE, see fpx_oracle.h).
 *
 * Plain-C restatement of the reference CPU search path of acoustid/acoustid-index:
 *   src/streamvbyte.zig  (StreamVByte 0124 / 1234 codec)
 *   src/block.zig        (BlockEncoder / BlockReader)
 *   src/filefmt.zig:94-138 (writeBlocks: block fill + block index + terminator)
 *   src/FileSegment.zig:83-89,135-180 (search with the 4-block / 1000-doc caps)
 *   src/MemorySegment.zig:44-54,81-148
 *   src/common.zig:61-71,121-171 (SearchResults.incr / finish)
 *   src/Index.zig:133-149,170-177,489-499 (hasNewerCommit, IndexReader.search, dedupSorted)
 *   src/MultiIndex.zig:302-306 (SearchOptions derivation)
 * Each function names the lines it follows.  Pinned by tests/test_oracle_kat.py.
 */
#include "fpx_oracle.h"

#include <pthread.h>
#include <sched.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

/* ======================================================================= */
/* StreamVByte                                                              */
/* ======================================================================= */

static uint8_t g_len_0124[256], g_len_1234[256];
static uint8_t g_shuf_0124[256][16] __attribute__((aligned(16)));
static uint8_t g_shuf_1234[256][16] __attribute__((aligned(16)));
static int g_tables_ready = 0;
static int g_simd = 0;

/* src/streamvbyte.zig:100-211 -- shuffle masks (0x80 = emit zero byte) and byte lengths */
static void init_tables(void)
{
    static const uint8_t bytes_0124[4] = {0, 1, 2, 4};
    static const uint8_t bytes_1234[4] = {1, 2, 3, 4};
    for (int control = 0; control < 256; control++) {
        uint8_t off0 = 0, off1 = 0;
        memset(g_shuf_0124[control], 0x80, 16);
        memset(g_shuf_1234[control], 0x80, 16);
        for (int i = 0; i < 4; i++) {
            int code = (control >> (2 * i)) & 3;
            for (int b = 0; b < bytes_0124[code]; b++) g_shuf_0124[control][i * 4 + b] = (uint8_t)(off0 + b);
            off0 += bytes_0124[code];
            for (int b = 0; b < bytes_1234[code]; b++) g_shuf_1234[control][i * 4 + b] = (uint8_t)(off1 + b);
            off1 += bytes_1234[code];
        }
        g_len_0124[control] = off0;
        g_len_1234[control] = off1;
    }
#if defined(__x86_64__)
    g_simd = __builtin_cpu_supports("ssse3") ? 1 : 0;
#endif
    g_tables_ready = 1;
}

__attribute__((constructor)) static void orc_ctor(void) { init_tables(); }

void orc_set_simd(int on)
{
#if defined(__x86_64__)
    g_simd = (on && __builtin_cpu_supports("ssse3")) ? 1 : 0;
#else
    (void)on; g_simd = 0;
#endif
}
int orc_get_simd(void) { return g_simd; }

uint8_t orc_svb_length(int variant, uint8_t control)
{
    return variant == ORC_V1234 ? g_len_1234[control] : g_len_0124[control];
}

/* scalar form of svbDecodeQuadBase (:216-247): the pshufb table picks, for value i, its
 * little-endian bytes from the packed stream and zero-fills the rest */
static inline size_t decode_quad_scalar(int variant, uint8_t control, const uint8_t *in, uint32_t out[4])
{
    const uint8_t *p = in;
    for (int i = 0; i < 4; i++) {
        int code = (control >> (2 * i)) & 3;
        uint32_t v = 0;
        int nb = (variant == ORC_V1234) ? code + 1 : (code == 3 ? 4 : code);
        for (int b = 0; b < nb; b++) v |= (uint32_t)p[b] << (8 * b);
        p += nb;
        out[i] = v + (variant == ORC_V0124_MINUS1 ? 1u : 0u);
    }
    return (size_t)(p - in);
}

#if defined(__x86_64__)
__attribute__((target("ssse3")))
static inline size_t decode_quad_ssse3(int variant, uint8_t control, const uint8_t *in, uint32_t out[4])
{
    __m128i data = _mm_loadu_si128((const __m128i *)in);
    const uint8_t *mask = (variant == ORC_V1234) ? g_shuf_1234[control] : g_shuf_0124[control];
    __m128i r = _mm_shuffle_epi8(data, _mm_load_si128((const __m128i *)mask));
    if (variant == ORC_V0124_MINUS1) r = _mm_add_epi32(r, _mm_set1_epi32(1));
    _mm_storeu_si128((__m128i *)out, r);
    return (variant == ORC_V1234) ? g_len_1234[control] : g_len_0124[control];
}
#endif

size_t orc_svb_decode_quad(int variant, uint8_t control, const uint8_t *in, uint32_t out[4])
{
#if defined(__x86_64__)
    if (g_simd) return decode_quad_ssse3(variant, control, in, out);
#endif
    return decode_quad_scalar(variant, control, in, out);
}

/* svbDecodeQuadWithDelta (:264-283): in-quad inclusive prefix sum, then + carry (wrapping u32) */
size_t orc_svb_decode_quad_delta(int variant, uint8_t control, const uint8_t *in, uint32_t out[4], uint32_t carry)
{
    uint32_t v[4];
    size_t consumed = orc_svb_decode_quad(variant, control, in, v);
    v[1] += v[0]; v[2] += v[1]; v[3] += v[2];
    out[0] = v[0] + carry; out[1] = v[1] + carry; out[2] = v[2] + carry; out[3] = v[3] + carry;
    return consumed;
}

/* svbDeltaDecodeInPlace (:287-339): data[0] += first; data[i] += data[i-1] */
void orc_svb_delta_decode_in_place(uint32_t *data, size_t n, uint32_t first)
{
    if (n == 0) return;
    data[0] += first;
    for (size_t i = 1; i < n; i++) data[i] += data[i - 1];
}

/* decodeValues (:341-412) */
void orc_svb_decode_values(size_t total_items, size_t start_item, size_t end_item,
                           const uint8_t *in, uint32_t *out, int variant, int delta, uint32_t first_value)
{
    const uint8_t *len = (variant == ORC_V1234) ? g_len_1234 : g_len_0124;
    size_t start_quad = start_item / 4;
    size_t end_quad = (end_item + 3) / 4;
    size_t total_quads = (total_items + 3) / 4;

    size_t data_offset = total_quads;                       /* :361 */
    for (size_t q = 0; q < start_quad; q++) data_offset += len[in[q]];
    const uint8_t *ctrl = in + start_quad;
    const uint8_t *data = in + data_offset;
    uint32_t *o = out + start_quad * 4;
    size_t remaining = end_quad - start_quad;

    if (delta) {
        uint32_t carry = first_value;                       /* :376 */
        while (remaining > 0) {
            size_t c = orc_svb_decode_quad_delta(variant, *ctrl, data, o, carry);
            carry = o[3];
            ctrl++; data += c; o += 4; remaining--;
        }
    } else {
        while (remaining > 0) {                             /* :392-411 (the 8x unroll is an optimisation) */
            size_t c = orc_svb_decode_quad(variant, *ctrl, data, o);
            ctrl++; data += c; o += 4; remaining--;
        }
    }
}

/* svbEncodeValue0124 / svbEncodeQuad0124 (:418-469) */
size_t orc_svb_encode_quad_0124(const uint32_t in[4], uint8_t *out_data, uint8_t *out_control)
{
    uint8_t *p = out_data;
    uint8_t control = 0;
    for (int i = 0; i < 4; i++) {
        uint32_t v = in[i];
        if (v == 0) {
            /* code 0, no bytes */
        } else if (v < (1u << 8)) {
            p[0] = (uint8_t)v; p += 1; control |= (uint8_t)(1u << (2 * i));
        } else if (v < (1u << 16)) {
            p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p += 2; control |= (uint8_t)(2u << (2 * i));
        } else {
            p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
            p += 4; control |= (uint8_t)(3u << (2 * i));
        }
    }
    *out_control = control;
    return (size_t)(p - out_data);
}

/* svbEncodeValue1234 / svbEncodeQuad1234 (:441-480) */
size_t orc_svb_encode_quad_1234(const uint32_t in[4], uint8_t *out_data, uint8_t *out_control)
{
    uint8_t *p = out_data;
    uint8_t control = 0;
    for (int i = 0; i < 4; i++) {
        uint32_t v = in[i];
        int code = (v < (1u << 8)) ? 0 : (v < (1u << 16)) ? 1 : (v < (1u << 24)) ? 2 : 3;
        for (int b = 0; b <= code; b++) p[b] = (uint8_t)(v >> (8 * b));
        p += code + 1;
        control |= (uint8_t)(code << (2 * i));
    }
    *out_control = control;
    return (size_t)(p - out_data);
}

size_t orc_svb_encode_quad_size_0124(const uint32_t in[4])   /* :483-499 */
{
    size_t s = 0;
    for (int i = 0; i < 4; i++) s += in[i] == 0 ? 0 : in[i] < (1u << 8) ? 1 : in[i] < (1u << 16) ? 2 : 4;
    return s;
}

size_t orc_svb_encode_quad_size_1234(const uint32_t in[4])   /* :501-516 */
{
    size_t s = 0;
    for (int i = 0; i < 4; i++) s += in[i] < (1u << 8) ? 1 : in[i] < (1u << 16) ? 2 : in[i] < (1u << 24) ? 3 : 4;
    return s;
}

/* ======================================================================= */
/* Block codec                                                              */
/* ======================================================================= */

static inline uint32_t item_hash(uint64_t it) { return (uint32_t)(it >> 32); }  /* src/segment.zig:87-89 */
static inline uint32_t item_id(uint64_t it) { return (uint32_t)it; }

void orc_block_header_decode(const uint8_t *d, orc_block_header *h)   /* src/block.zig:46-56 (LE) */
{
    h->min_hash = (uint32_t)d[0] | (uint32_t)d[1] << 8 | (uint32_t)d[2] << 16 | (uint32_t)d[3] << 24;
    h->num_items = (uint16_t)(d[4] | d[5] << 8);
    h->docids_offset = (uint16_t)(d[6] | d[7] << 8);
}

static void block_header_encode(const orc_block_header *h, uint8_t *d)
{
    d[0] = (uint8_t)h->min_hash; d[1] = (uint8_t)(h->min_hash >> 8);
    d[2] = (uint8_t)(h->min_hash >> 16); d[3] = (uint8_t)(h->min_hash >> 24);
    d[4] = (uint8_t)h->num_items; d[5] = (uint8_t)(h->num_items >> 8);
    d[6] = (uint8_t)h->docids_offset; d[7] = (uint8_t)(h->docids_offset >> 8);
}

/* BlockEncoder state (src/block.zig:420-436) */
typedef struct {
    uint16_t num_items;
    uint32_t last_hash, last_docid;
    uint8_t hashes[ORC_MAX_BLOCK_SIZE + 16], hashes_ctrl[ORC_MAX_BLOCK_SIZE + 16];
    uint8_t docids[ORC_MAX_BLOCK_SIZE + 16], docids_ctrl[ORC_MAX_BLOCK_SIZE + 16];
    size_t hashes_len, hashes_ctrl_len, docids_len, docids_ctrl_len;
} block_encoder;

/* encodeChunk (:438-499): returns 0 on success, 1 on BlockFull */
static int encode_chunk(block_encoder *e, const uint64_t *items, size_t n, uint32_t min_doc_id, size_t block_size)
{
    uint32_t ch[4] = {0, 0, 0, 0}, cd[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < n; i++) {
        uint32_t h = item_hash(items[i]), d = item_id(items[i]);
        ch[i] = h - e->last_hash;
        cd[i] = (h != e->last_hash) ? d - min_doc_id : d - e->last_docid;   /* :453-460 */
        e->last_hash = h;
        e->last_docid = d;
    }
    size_t hs = orc_svb_encode_quad_0124(ch, e->hashes + e->hashes_len, &e->hashes_ctrl[e->hashes_ctrl_len]);
    size_t ds = orc_svb_encode_quad_1234(cd, e->docids + e->docids_len, &e->docids_ctrl[e->docids_ctrl_len]);
    size_t new_size = ORC_BLOCK_HEADER_SIZE + e->hashes_len + hs + e->hashes_ctrl_len + 1 +
                      e->docids_len + ds + e->docids_ctrl_len + 1;              /* :480-482 */
    if (new_size > block_size) return 1;
    e->hashes_len += hs; e->hashes_ctrl_len += 1;
    e->docids_len += ds; e->docids_ctrl_len += 1;
    e->num_items = (uint16_t)(e->num_items + n);
    return 0;
}

/* encodeBlock (:501-567) */
static size_t encode_block(block_encoder *e, const uint64_t *items, size_t n, uint32_t min_doc_id,
                           uint8_t *out, size_t block_size)
{
    if (n == 0) { memset(out, 0, block_size); return 0; }
    e->num_items = 0;
    e->hashes_len = e->hashes_ctrl_len = e->docids_len = e->docids_ctrl_len = 0;
    e->last_hash = item_hash(items[0]);
    e->last_docid = min_doc_id;

    const uint64_t *p = items;
    size_t left = n;
    while (left >= 4) {
        if (encode_chunk(e, p, 4, min_doc_id, block_size)) { left = 0; break; }   /* :524-533 */
        p += 4; left -= 4;
    }
    if (left > 0) (void)encode_chunk(e, p, left, min_doc_id, block_size);        /* :536-542 */

    orc_block_header h;
    h.min_hash = item_hash(items[0]);
    h.num_items = e->num_items;
    h.docids_offset = (uint16_t)(e->hashes_len + e->hashes_ctrl_len);
    block_header_encode(&h, out);
    size_t w = ORC_BLOCK_HEADER_SIZE;
    memcpy(out + w, e->hashes_ctrl, e->hashes_ctrl_len); w += e->hashes_ctrl_len;
    memcpy(out + w, e->hashes, e->hashes_len); w += e->hashes_len;
    memcpy(out + w, e->docids_ctrl, e->docids_ctrl_len); w += e->docids_ctrl_len;
    memcpy(out + w, e->docids, e->docids_len); w += e->docids_len;
    memset(out + w, 0, block_size - w);
    return e->num_items;
}

size_t orc_block_encode(const uint64_t *items, size_t n, uint32_t min_doc_id, uint8_t *out, size_t block_size)
{
    block_encoder *e = (block_encoder *)malloc(sizeof *e);
    if (!e) return 0;
    size_t r = encode_block(e, items, n, min_doc_id, out, block_size);
    free(e);
    return r;
}

/* BlockReader (src/block.zig:66-312).  block_data may be shorter than 16 bytes past the
 * last value; the reference guarantees 16 readable bytes (FileSegment.loadBlockData :83-89),
 * here a padded private copy is used when the caller's slice is not long enough. */
typedef struct {
    uint32_t min_doc_id;
    const uint8_t *block;
    size_t block_len;
    uint32_t hashes[ORC_MAX_ITEMS_PER_BLOCK + 4];
    uint32_t docids[ORC_MAX_ITEMS_PER_BLOCK + 4];
    int hashes_loaded;
    orc_block_header h;
} block_reader;

static void reader_load(block_reader *r, const uint8_t *block, size_t len)   /* :104-117 (lazy) */
{
    r->block = block; r->block_len = len; r->hashes_loaded = 0;
    orc_block_header_decode(block, &r->h);
}

static void reader_ensure_hashes(block_reader *r)   /* :137-158 */
{
    if (r->hashes_loaded) return;
    if (r->h.num_items != 0)
        orc_svb_decode_values(r->h.num_items, 0, r->h.num_items, r->block + ORC_BLOCK_HEADER_SIZE,
                              r->hashes, ORC_V0124, 1, r->h.min_hash);
    r->hashes_loaded = 1;
}

/* std.sort.equalRange over hashes[0..n) (:217-231) */
static void reader_find_hash(block_reader *r, uint32_t hash, uint32_t *start, uint32_t *end)
{
    if (r->h.num_items == 0) { *start = 0; *end = 0; return; }
    reader_ensure_hashes(r);
    uint32_t n = r->h.num_items, lo = 0, hi = n;
    while (lo < hi) { uint32_t m = lo + (hi - lo) / 2; if (r->hashes[m] < hash) lo = m + 1; else hi = m; }
    *start = lo;
    hi = n;
    while (lo < hi) { uint32_t m = lo + (hi - lo) / 2; if (r->hashes[m] <= hash) lo = m + 1; else hi = m; }
    *end = lo;
}

/* getDocidsForRange (:235-265): returns pointer into r->docids, *n = count */
static const uint32_t *reader_docids_for_range(block_reader *r, uint32_t start, uint32_t end, size_t *n)
{
    if (start >= end) { *n = 0; return r->docids; }
    orc_svb_decode_values(r->h.num_items, start, end,
                          r->block + ORC_BLOCK_HEADER_SIZE + r->h.docids_offset,
                          r->docids, ORC_V1234, 0, 0);
    orc_svb_delta_decode_in_place(r->docids + start, end - start, r->min_doc_id);
    *n = end - start;
    return r->docids + start;
}

/* pad helper for the public single-block entry points */
static uint8_t *padded_copy(const uint8_t *block, size_t len)
{
    uint8_t *c = (uint8_t *)calloc(len + 32, 1);
    if (c) memcpy(c, block, len);
    return c;
}

void orc_block_find_hash(const uint8_t *block, size_t block_len, uint32_t hash, uint32_t *start, uint32_t *end)
{
    uint8_t *c = padded_copy(block, block_len);
    block_reader *r = (block_reader *)malloc(sizeof *r);
    r->min_doc_id = 0;
    reader_load(r, c, block_len + 32);
    reader_find_hash(r, hash, start, end);
    free(r); free(c);
}

size_t orc_block_search_hash(const uint8_t *block, size_t block_len, uint32_t min_doc_id, uint32_t hash, uint32_t *out)
{
    uint8_t *c = padded_copy(block, block_len);
    block_reader *r = (block_reader *)malloc(sizeof *r);
    r->min_doc_id = min_doc_id;
    reader_load(r, c, block_len + 32);
    uint32_t s, e; size_t n;
    reader_find_hash(r, hash, &s, &e);
    const uint32_t *d = reader_docids_for_range(r, s, e, &n);
    memcpy(out, d, n * sizeof(uint32_t));
    free(r); free(c);
    return n;
}

/* ensureDocidsLoaded + getItems (:159-203, :282-305): docid delta base resets to min_doc_id
 * at every hash change */
size_t orc_block_decode_items(const uint8_t *block, size_t block_len, uint32_t min_doc_id,
                              uint32_t *hashes, uint32_t *docids)
{
    uint8_t *c = padded_copy(block, block_len);
    orc_block_header h;
    orc_block_header_decode(c, &h);
    size_t n = h.num_items;
    if (n) {
        uint32_t *th = (uint32_t *)malloc((n + 4) * sizeof(uint32_t));
        uint32_t *td = (uint32_t *)malloc((n + 4) * sizeof(uint32_t));
        orc_svb_decode_values(n, 0, n, c + ORC_BLOCK_HEADER_SIZE, th, ORC_V0124, 1, h.min_hash);
        orc_svb_decode_values(n, 0, n, c + ORC_BLOCK_HEADER_SIZE + h.docids_offset, td, ORC_V1234, 0, 0);
        uint32_t last_docid = min_doc_id, last_hash = th[0];
        for (size_t i = 0; i < n; i++) {
            if (th[i] != last_hash) { last_docid = min_doc_id; last_hash = th[i]; }
            td[i] += last_docid;
            last_docid = td[i];
        }
        memcpy(hashes, th, n * sizeof(uint32_t));
        memcpy(docids, td, n * sizeof(uint32_t));
        free(th); free(td);
    }
    free(c);
    return n;
}

/* ======================================================================= */
/* Segment build: filefmt.writeBlocks (src/filefmt.zig:94-138)             */
/* ======================================================================= */

void orc_free(void *p) { free(p); }

int orc_build_blocks(const uint64_t *items, size_t n, uint32_t min_doc_id, uint32_t block_size,
                     uint8_t **blocks_out, size_t *blocks_len, uint32_t **index_out, uint32_t *num_blocks_out)
{
    if (block_size < ORC_MIN_BLOCK_SIZE || block_size > ORC_MAX_BLOCK_SIZE) return -1;
    block_encoder *e = (block_encoder *)malloc(sizeof *e);
    size_t cap_blocks = n / 8 + 16;                /* grows below if needed */
    uint8_t *blocks = (uint8_t *)malloc(cap_blocks * (size_t)block_size);
    uint32_t *index = (uint32_t *)malloc(cap_blocks * sizeof(uint32_t));
    if (!e || !blocks || !index) { free(e); free(blocks); free(index); return -1; }
    size_t nb = 0, pos = 0;
    for (;;) {
        if (nb + 2 > cap_blocks) {
            cap_blocks *= 2;
            blocks = (uint8_t *)realloc(blocks, cap_blocks * (size_t)block_size);
            index = (uint32_t *)realloc(index, cap_blocks * sizeof(uint32_t));
            if (!blocks || !index) { free(e); return -1; }
        }
        /* the writer refills a MAX_ITEMS_PER_BLOCK window before every block (:108-113) */
        size_t window = n - pos;
        if (window > ORC_MAX_ITEMS_PER_BLOCK) window = ORC_MAX_ITEMS_PER_BLOCK;
        size_t consumed = encode_block(e, items + pos, window, min_doc_id, blocks + nb * (size_t)block_size, block_size);
        if (consumed == 0) {
            if (window != 0) { free(e); free(blocks); free(index); return -2; } /* block too small for a chunk */
            break;                                   /* empty terminator block written (:117) */
        }
        index[nb] = item_hash(items[pos + consumed - 1]);    /* :119 */
        nb++;
        pos += consumed;
    }
    free(e);
    *blocks_out = blocks;
    *blocks_len = (nb + 1) * (size_t)block_size;
    *index_out = index;
    *num_blocks_out = (uint32_t)nb;
    return 0;
}

/* ======================================================================= */
/* Segments                                                                 */
/* ======================================================================= */

struct orc_segment {
    int is_file;
    int owns_blocks;
    /* file */
    const uint8_t *blocks; size_t blocks_len; uint32_t block_size;
    const uint32_t *block_index; uint32_t num_blocks;
    /* memory */
    uint64_t *items; size_t num_items;
    /* common */
    uint32_t min_doc_id, max_doc_id; uint64_t commit_id;
    uint32_t *doc_ids; uint8_t *doc_alive; uint32_t num_docs;   /* sorted by id */
    /* test aid for hash-range sharding: only hashes in (win_lo, win_hi] are scanned in this (slice of a) segment */
    int win_has_lo, win_has_hi; uint32_t win_lo, win_hi;
};

static int cmp_u32(const void *a, const void *b)
{
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : x > y;
}

static int set_docs(orc_segment *s, const uint32_t *ids, const uint8_t *alive, uint32_t n)
{
    s->num_docs = n;
    s->doc_ids = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    s->doc_alive = (uint8_t *)malloc(n ? n : 1);
    if (!s->doc_ids || !s->doc_alive) return -1;
    /* sort (id, alive) by id */
    uint64_t *tmp = (uint64_t *)malloc((n ? n : 1) * sizeof(uint64_t));
    if (!tmp) return -1;
    for (uint32_t i = 0; i < n; i++) tmp[i] = (uint64_t)ids[i] << 8 | (alive ? (alive[i] ? 1u : 0u) : 1u);
    orc_sort_u64(tmp, n);
    for (uint32_t i = 0; i < n; i++) { s->doc_ids[i] = (uint32_t)(tmp[i] >> 8); s->doc_alive[i] = (uint8_t)(tmp[i] & 1); }
    free(tmp);
    return 0;
}

static orc_segment *create_file(const uint8_t *blocks, size_t blocks_len, uint32_t block_size,
                                const uint32_t *block_index, uint32_t num_blocks,
                                uint32_t min_doc_id, uint32_t max_doc_id, uint64_t commit_id,
                                const uint32_t *doc_ids, const uint8_t *doc_alive, uint32_t num_docs, int copy)
{
    orc_segment *s = (orc_segment *)calloc(1, sizeof *s);
    if (!s) return NULL;
    s->is_file = 1; s->owns_blocks = copy;
    if (copy) {
        uint8_t *b = (uint8_t *)malloc(blocks_len + 16);
        uint32_t *ix = (uint32_t *)malloc((num_blocks ? num_blocks : 1) * sizeof(uint32_t));
        if (!b || !ix) { free(b); free(ix); free(s); return NULL; }
        memcpy(b, blocks, blocks_len); memset(b + blocks_len, 0, 16);
        memcpy(ix, block_index, (size_t)num_blocks * sizeof(uint32_t));
        s->blocks = b; s->block_index = ix;
    } else {
        s->blocks = blocks; s->block_index = block_index;
    }
    s->blocks_len = blocks_len; s->block_size = block_size; s->num_blocks = num_blocks;
    s->min_doc_id = min_doc_id; s->max_doc_id = max_doc_id; s->commit_id = commit_id;
    if (set_docs(s, doc_ids, doc_alive, num_docs)) { orc_segment_free(s); return NULL; }
    return s;
}

orc_segment *orc_segment_create_file(const uint8_t *blocks, size_t blocks_len, uint32_t block_size,
                                     const uint32_t *block_index, uint32_t num_blocks,
                                     uint32_t min_doc_id, uint32_t max_doc_id, uint64_t commit_id,
                                     const uint32_t *doc_ids, const uint8_t *doc_alive, uint32_t num_docs)
{
    return create_file(blocks, blocks_len, block_size, block_index, num_blocks, min_doc_id, max_doc_id,
                       commit_id, doc_ids, doc_alive, num_docs, 1);
}

orc_segment *orc_segment_create_file_borrowed(const uint8_t *blocks, size_t blocks_len, uint32_t block_size,
                                     const uint32_t *block_index, uint32_t num_blocks,
                                     uint32_t min_doc_id, uint32_t max_doc_id, uint64_t commit_id,
                                     const uint32_t *doc_ids, const uint8_t *doc_alive, uint32_t num_docs)
{
    return create_file(blocks, blocks_len, block_size, block_index, num_blocks, min_doc_id, max_doc_id,
                       commit_id, doc_ids, doc_alive, num_docs, 0);
}

orc_segment *orc_segment_create_memory(const uint64_t *items, size_t n,
                                       uint32_t min_doc_id, uint32_t max_doc_id, uint64_t commit_id,
                                       const uint32_t *doc_ids, const uint8_t *doc_alive, uint32_t num_docs)
{
    orc_segment *s = (orc_segment *)calloc(1, sizeof *s);
    if (!s) return NULL;
    s->items = (uint64_t *)malloc((n ? n : 1) * sizeof(uint64_t));
    if (!s->items) { free(s); return NULL; }
    memcpy(s->items, items, n * sizeof(uint64_t));
    s->num_items = n;
    s->min_doc_id = min_doc_id; s->max_doc_id = max_doc_id; s->commit_id = commit_id;
    if (set_docs(s, doc_ids, doc_alive, num_docs)) { orc_segment_free(s); return NULL; }
    return s;
}

/* MemorySegment.build (src/MemorySegment.zig:81-148): reverse scan, first occurrence of an
 * id wins (so the LAST change for an id in the batch is the one kept); items sorted as u64 */
orc_segment *orc_segment_build_memory(const uint8_t *kind, const uint32_t *ids,
                                      const uint32_t *hashes, const uint64_t *hash_off,
                                      size_t num_changes, uint64_t commit_id)
{
    orc_segment *s = (orc_segment *)calloc(1, sizeof *s);
    if (!s) return NULL;
    size_t total = 0;
    for (size_t i = 0; i < num_changes; i++) if (kind[i] == 0) total += (size_t)(hash_off[i + 1] - hash_off[i]);
    s->items = (uint64_t *)malloc((total ? total : 1) * sizeof(uint64_t));
    uint32_t *dids = (uint32_t *)malloc((num_changes ? num_changes : 1) * sizeof(uint32_t));
    uint8_t *dalive = (uint8_t *)malloc(num_changes ? num_changes : 1);
    uint32_t nd = 0; size_t ni = 0;
    s->commit_id = commit_id;
    for (size_t i = num_changes; i-- > 0;) {
        uint32_t id = ids[i];
        int seen = 0;
        for (uint32_t k = 0; k < nd; k++) if (dids[k] == id) { seen = 1; break; }   /* docs.getOrPut */
        if (seen) continue;
        dids[nd] = id; dalive[nd] = (kind[i] == 0); nd++;
        if (kind[i] == 0)
            for (uint64_t j = hash_off[i]; j < hash_off[i + 1]; j++) s->items[ni++] = (uint64_t)hashes[j] << 32 | id;
        if (s->min_doc_id == 0 || id < s->min_doc_id) s->min_doc_id = id;
        if (s->max_doc_id == 0 || id > s->max_doc_id) s->max_doc_id = id;
    }
    s->num_items = ni;
    orc_sort_u64(s->items, ni);
    int rc = set_docs(s, dids, dalive, nd);
    free(dids); free(dalive);
    if (rc) { orc_segment_free(s); return NULL; }
    return s;
}

void orc_segment_set_window(orc_segment *s, int has_lo, uint32_t lo_excl, int has_hi, uint32_t hi_incl)
{
    s->win_has_lo = has_lo; s->win_lo = lo_excl; s->win_has_hi = has_hi; s->win_hi = hi_incl;
}

void orc_segment_free(orc_segment *s)
{
    if (!s) return;
    if (s->is_file && s->owns_blocks) { free((void *)s->blocks); free((void *)s->block_index); }
    free(s->items); free(s->doc_ids); free(s->doc_alive);
    free(s);
}

size_t orc_segment_num_items(const orc_segment *s) { return s->num_items; }
const uint64_t *orc_segment_items(const orc_segment *s) { return s->items; }
uint32_t orc_segment_min_doc_id(const orc_segment *s) { return s->min_doc_id; }
uint32_t orc_segment_max_doc_id(const orc_segment *s) { return s->max_doc_id; }
uint32_t orc_segment_num_docs(const orc_segment *s) { return s->num_docs; }
const uint32_t *orc_segment_doc_ids(const orc_segment *s) { return s->doc_ids; }
const uint8_t *orc_segment_doc_alive(const orc_segment *s) { return s->doc_alive; }

static int docs_contains(const orc_segment *s, uint32_t id)
{
    uint32_t lo = 0, hi = s->num_docs;
    while (lo < hi) { uint32_t m = lo + (hi - lo) / 2; if (s->doc_ids[m] < id) lo = m + 1; else hi = m; }
    return lo < s->num_docs && s->doc_ids[lo] == id;
}

struct orc_snapshot {
    orc_segment **file; uint32_t n_file;
    orc_segment **memory; uint32_t n_memory;
};

orc_snapshot *orc_snapshot_create(orc_segment *const *file, uint32_t n_file,
                                  orc_segment *const *memory, uint32_t n_memory)
{
    orc_snapshot *s = (orc_snapshot *)calloc(1, sizeof *s);
    if (!s) return NULL;
    s->file = (orc_segment **)malloc((n_file ? n_file : 1) * sizeof(void *));
    s->memory = (orc_segment **)malloc((n_memory ? n_memory : 1) * sizeof(void *));
    if (n_file) memcpy(s->file, file, n_file * sizeof(void *));
    if (n_memory) memcpy(s->memory, memory, n_memory * sizeof(void *));
    s->n_file = n_file; s->n_memory = n_memory;
    return s;
}

void orc_snapshot_free(orc_snapshot *s) { if (s) { free(s->file); free(s->memory); free(s); } }

/* Segments.hasNewerCommit (src/Index.zig:133-149) */
int orc_snapshot_has_newer_commit(const orc_snapshot *s, uint32_t id, uint64_t commit_id)
{
    for (uint32_t i = s->n_memory; i-- > 0;) {
        const orc_segment *g = s->memory[i];
        if (g->commit_id <= commit_id) return 0;
        if (id >= g->min_doc_id && id <= g->max_doc_id && docs_contains(g, id)) return 1;
    }
    for (uint32_t j = s->n_file; j-- > 0;) {
        const orc_segment *g = s->file[j];
        if (g->commit_id <= commit_id) return 0;
        if (id >= g->min_doc_id && id <= g->max_doc_id && docs_contains(g, id)) return 1;
    }
    return 0;
}

/* ======================================================================= */
/* SegmentMerger (src/segment_merger.zig)                                   */
/* ======================================================================= */

/* all items of a segment in Item order: what Segment.Reader.read()/advance() yields one by one
 * (FileSegment: block after block, src/FileSegment.zig reader; MemorySegment: the items array) */
static uint64_t *segment_all_items(const orc_segment *s, size_t *n_out)
{
    if (!s->is_file) {
        uint64_t *v = (uint64_t *)malloc((s->num_items ? s->num_items : 1) * sizeof(uint64_t));
        if (!v) return NULL;
        memcpy(v, s->items, s->num_items * sizeof(uint64_t));
        *n_out = s->num_items;
        return v;
    }
    size_t total = 0;
    for (uint32_t b = 0; b < s->num_blocks; b++) {
        orc_block_header h;
        orc_block_header_decode(s->blocks + (size_t)b * s->block_size, &h);
        total += h.num_items;
    }
    uint64_t *v = (uint64_t *)malloc((total ? total : 1) * sizeof(uint64_t));
    uint32_t *th = (uint32_t *)malloc(ORC_MAX_ITEMS_PER_BLOCK * sizeof(uint32_t));
    uint32_t *td = (uint32_t *)malloc(ORC_MAX_ITEMS_PER_BLOCK * sizeof(uint32_t));
    if (!v || !th || !td) { free(v); free(th); free(td); return NULL; }
    size_t k = 0;
    for (uint32_t b = 0; b < s->num_blocks; b++) {
        size_t n = orc_block_decode_items(s->blocks + (size_t)b * s->block_size, s->block_size, s->min_doc_id, th, td);
        for (size_t i = 0; i < n; i++) v[k++] = (uint64_t)th[i] << 32 | td[i];
    }
    free(th); free(td);
    *n_out = k;
    return v;
}

int orc_merge_segments(const orc_snapshot *collection, orc_segment *const *sources, uint32_t n_sources,
                       uint64_t **items_out, size_t *num_items_out,
                       uint32_t **doc_ids_out, uint8_t **doc_alive_out, uint32_t *num_docs_out,
                       uint32_t *min_doc_id_out, uint32_t *max_doc_id_out, uint64_t *commit_id_out)
{
    if (n_sources == 0) return -2;                                           /* :88 error.NoSources */
    int rc = -1;
    uint64_t **src_items = (uint64_t **)calloc(n_sources, sizeof(*src_items));
    size_t *src_n = (size_t *)calloc(n_sources, sizeof(*src_n));
    size_t *cursor = (size_t *)calloc(n_sources, sizeof(*cursor));
    uint8_t **skip = (uint8_t **)calloc(n_sources, sizeof(*skip));           /* skip_docs, indexed like doc_ids */
    uint64_t *docs = NULL, *out = NULL;
    if (!src_items || !src_n || !cursor || !skip) goto done;

    /* prepare(): merged info, docs map, skip_docs (:90-131) */
    uint64_t commit = sources[0]->commit_id;
    size_t total_docs = 0, total_items = 0;
    for (uint32_t i = 0; i < n_sources; i++) {
        if (sources[i]->commit_id < commit) commit = sources[i]->commit_id;
        total_docs += sources[i]->num_docs;
    }
    docs = (uint64_t *)malloc((total_docs ? total_docs : 1) * sizeof(uint64_t));   /* id<<8 | alive */
    if (!docs) goto done;
    size_t nd = 0;
    uint32_t mn = 0, mx = 0;
    for (uint32_t i = 0; i < n_sources; i++) {
        const orc_segment *g = sources[i];
        skip[i] = (uint8_t *)calloc(g->num_docs ? g->num_docs : 1, 1);
        if (!skip[i]) goto done;
        for (uint32_t d = 0; d < g->num_docs; d++) {
            const uint32_t id = g->doc_ids[d];
            if (!orc_snapshot_has_newer_commit(collection, id, g->commit_id)) {
                docs[nd++] = (uint64_t)id << 8 | g->doc_alive[d];
                if (mn == 0 || id < mn) mn = id;
                if (mx == 0 || id > mx) mx = id;
            } else {
                skip[i][d] = 1;
            }
        }
        src_items[i] = segment_all_items(g, &src_n[i]);
        if (!src_items[i]) goto done;
        total_items += src_n[i];
    }
    orc_sort_u64(docs, nd);                     /* ids are unique here: an older copy of a doc is always skipped */

    /* read()/advance() (:133-155): smallest head item wins, the first source on ties */
    out = (uint64_t *)malloc((total_items ? total_items : 1) * sizeof(uint64_t));
    if (!out) goto done;
    size_t k = 0;
    for (;;) {
        int best = -1;
        uint64_t best_item = 0;
        for (uint32_t i = 0; i < n_sources; i++) {
            const orc_segment *g = sources[i];
            while (cursor[i] < src_n[i]) {                                   /* Source.read(): drop skipped docs (:44-53) */
                const uint32_t id = (uint32_t)src_items[i][cursor[i]];
                uint32_t lo = 0, hi = g->num_docs;
                while (lo < hi) { uint32_t m = lo + (hi - lo) / 2; if (g->doc_ids[m] < id) lo = m + 1; else hi = m; }
                if (lo < g->num_docs && g->doc_ids[lo] == id && skip[i][lo]) cursor[i]++; else break;
            }
            if (cursor[i] < src_n[i] && (best < 0 || src_items[i][cursor[i]] < best_item)) {
                best = (int)i; best_item = src_items[i][cursor[i]];
            }
        }
        if (best < 0) break;
        cursor[best]++;
        out[k++] = best_item;
    }

    *doc_ids_out = (uint32_t *)malloc((nd ? nd : 1) * sizeof(uint32_t));
    *doc_alive_out = (uint8_t *)malloc(nd ? nd : 1);
    if (!*doc_ids_out || !*doc_alive_out) { free(*doc_ids_out); free(*doc_alive_out); goto done; }
    for (size_t i = 0; i < nd; i++) { (*doc_ids_out)[i] = (uint32_t)(docs[i] >> 8); (*doc_alive_out)[i] = (uint8_t)(docs[i] & 1); }
    *items_out = out; out = NULL;
    *num_items_out = k;
    *num_docs_out = (uint32_t)nd;
    *min_doc_id_out = mn; *max_doc_id_out = mx; *commit_id_out = commit;
    rc = 0;
done:
    if (src_items) for (uint32_t i = 0; i < n_sources; i++) free(src_items[i]);
    if (skip) for (uint32_t i = 0; i < n_sources; i++) free(skip[i]);
    free(src_items); free(src_n); free(cursor); free(skip); free(docs); free(out);
    return rc;
}

/* ======================================================================= */
/* SearchResults (src/common.zig:73-176)                                    */
/* ======================================================================= */

typedef struct { uint32_t id; uint32_t score; uint64_t commit_id; } hit_slot;   /* id==0 -> empty */

typedef struct {
    hit_slot *slots; size_t cap; size_t count;    /* cap is a power of two */
    int has_zero; hit_slot zero;                 /* id 0 is illegal upstream but keep the map total */
    int oom;
} hit_map;

static inline uint64_t hit_hash(uint32_t key)     /* HitContext.hash (:61-68) */
{
    uint64_t x = key;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
    return x ^ (x >> 31);
}

static int map_init(hit_map *m, size_t cap)
{
    m->slots = (hit_slot *)calloc(cap, sizeof(hit_slot));
    m->cap = cap; m->count = 0; m->has_zero = 0; m->oom = 0;
    return m->slots ? 0 : -1;
}

static hit_slot *map_slot(hit_map *m, uint32_t id, int *found)
{
    if (id == 0) { *found = m->has_zero; m->has_zero = 1; return &m->zero; }
    size_t mask = m->cap - 1, i = (size_t)hit_hash(id) & mask;
    for (;;) {
        hit_slot *s = &m->slots[i];
        if (s->id == id) { *found = 1; return s; }
        if (s->id == 0) { *found = 0; return s; }
        i = (i + 1) & mask;
    }
}

static int map_grow(hit_map *m)
{
    hit_map n;
    if (map_init(&n, m->cap * 2)) return -1;
    for (size_t i = 0; i < m->cap; i++) if (m->slots[i].id) {
        int f; hit_slot *s = map_slot(&n, m->slots[i].id, &f);
        *s = m->slots[i]; n.count++;
    }
    n.has_zero = m->has_zero; n.zero = m->zero;
    free(m->slots);
    *m = n;
    return 0;
}

/* SearchResults.incr (:121-129) */
static void results_incr(hit_map *m, uint32_t id, uint64_t commit_id)
{
    if (m->oom) return;
    if ((m->count + 1) * 5 > m->cap * 4) { if (map_grow(m)) { m->oom = 1; return; } }
    int found; hit_slot *s = map_slot(m, id, &found);
    if (!found || s->commit_id < commit_id) {
        if (!found) { s->id = id; if (id) m->count++; }
        s->score = 1; s->commit_id = commit_id;
    } else if (s->commit_id == commit_id) {
        s->score += 1;
    }
}

/* ======================================================================= */
/* FileSegment.search (src/FileSegment.zig:135-180)                         */
/* ======================================================================= */

#define MAX_BLOCKS_PER_HASH 4      /* :25 */
#define MAX_DOCS_PER_HASH 1000     /* :26 */

typedef struct { size_t block_no; block_reader reader; } cache_entry;

/* `cache`: MAX_BLOCKS_PER_HASH reader slots owned by the caller (the reference keeps them on the search's stack,
 * src/FileSegment.zig:138-141) */
static int file_segment_search(const orc_segment *seg, const uint32_t *sorted_hashes, size_t n,
                               hit_map *results, orc_stats *st, cache_entry *cache)
{
    for (int i = 0; i < MAX_BLOCKS_PER_HASH; i++) { cache[i].block_no = (size_t)-1; cache[i].reader.min_doc_id = seg->min_doc_id; }

    size_t prev = 0;
    for (size_t qi = 0; qi < n; qi++) {
        uint32_t hash = sorted_hashes[qi];
        if ((seg->win_has_lo && hash <= seg->win_lo) || (seg->win_has_hi && hash > seg->win_hi)) continue;   /* another slice's */
        /* lowerBound over block_index[prev..] (:145-151) */
        size_t lo = prev, hi = seg->num_blocks;
        while (lo < hi) { size_t m = lo + (hi - lo) / 2; if (seg->block_index[m] < hash) lo = m + 1; else hi = m; }
        size_t block_no = lo;
        prev = block_no;

        size_t num_docs = 0; uint64_t num_blocks = 0;
        for (; block_no < seg->num_blocks; block_no++) {
            cache_entry *ce = &cache[block_no % MAX_BLOCKS_PER_HASH];
            if (ce->block_no != block_no) {
                ce->block_no = block_no;
                size_t start = block_no * (size_t)seg->block_size;
                size_t end = start + seg->block_size + 16;                 /* loadBlockData :83-89 */
                if (end > seg->blocks_len) end = seg->blocks_len;
                reader_load(&ce->reader, seg->blocks + start, end - start);
            }
            block_reader *r = &ce->reader;
            if (r->h.min_hash > hash) break;                                /* :164 */
            uint32_t s, e; size_t cnt;
            reader_find_hash(r, hash, &s, &e);
            const uint32_t *d = reader_docids_for_range(r, s, e, &cnt);
            for (size_t k = 0; k < cnt; k++) results_incr(results, d[k], seg->commit_id);
            num_blocks += 1;
            num_docs += cnt;
            if (num_blocks >= MAX_BLOCKS_PER_HASH) break;                   /* :173 */
            if (num_docs > MAX_DOCS_PER_HASH) break;                        /* :174 */
        }
        if (st) { st->scanned_blocks += num_blocks; st->scanned_docs += num_docs; st->probes += 1; }
    }
    return results->oom ? -1 : 0;
}

/* MemorySegment.search (src/MemorySegment.zig:44-54): equalRange by hash over the shrinking tail */
static int memory_segment_search(const orc_segment *seg, const uint32_t *sorted_hashes, size_t n, hit_map *results)
{
    const uint64_t *items = seg->items;
    size_t len = seg->num_items;
    for (size_t qi = 0; qi < n; qi++) {
        uint32_t hash = sorted_hashes[qi];
        size_t lo = 0, hi = len;
        while (lo < hi) { size_t m = lo + (hi - lo) / 2; if ((uint32_t)(items[m] >> 32) < hash) lo = m + 1; else hi = m; }
        size_t first = lo;
        hi = len;
        while (lo < hi) { size_t m = lo + (hi - lo) / 2; if ((uint32_t)(items[m] >> 32) <= hash) lo = m + 1; else hi = m; }
        for (size_t i = first; i < lo; i++) results_incr(results, (uint32_t)items[i], seg->commit_id);
        items += lo; len -= lo;
    }
    return results->oom ? -1 : 0;
}

uint32_t orc_default_min_score(uint32_t raw_query_len) { return (uint32_t)(((uint64_t)raw_query_len + 19) / 20); } /* MultiIndex.zig:304 */

/* Per-search working memory.  The reference takes it from a per-request arena and a pooled collector
 * (src/MultiIndex.zig:300-312, src/common.zig:186-300); a worker of orc_search_many keeps one of these for its lifetime
 * so that concurrent searches never meet in the allocator. */
typedef struct {
    uint32_t *q; size_t qcap;                 /* sorted + deduped copy of the query */
    cache_entry *cache;                       /* MAX_BLOCKS_PER_HASH block readers */
    orc_result *cand; size_t ccap;            /* finish()'s candidate list */
} search_scratch;

static void scratch_free(search_scratch *sc) { free(sc->q); free(sc->cache); free(sc->cand); memset(sc, 0, sizeof *sc); }

static int scratch_reserve(search_scratch *sc, size_t n)
{
    if (!sc->cache) {
        sc->cache = (cache_entry *)malloc(sizeof(cache_entry) * MAX_BLOCKS_PER_HASH);
        if (!sc->cache) return -1;
    }
    if (n + 1 > sc->qcap) {
        free(sc->q);
        sc->qcap = (n + 1) * 2;
        sc->q = (uint32_t *)malloc(sc->qcap * sizeof(uint32_t));
        if (!sc->q) { sc->qcap = 0; return -1; }
    }
    return 0;
}

/* sort + dedupSorted (src/Index.zig:171-172, :489-499) then the segment scans (:173-175) */
static int run_scans(const orc_snapshot *snap, const uint32_t *hashes, uint32_t n, hit_map *m, orc_stats *st, search_scratch *sc)
{
    if (scratch_reserve(sc, n)) return -1;
    uint32_t *q = sc->q;
    memcpy(q, hashes, (size_t)n * sizeof(uint32_t));
    qsort(q, n, sizeof(uint32_t), cmp_u32);
    uint32_t w = 0;
    if (n) { w = 1; for (uint32_t i = 1; i < n; i++) if (q[i] != q[w - 1]) q[w++] = q[i]; }
    int rc = 0;
    for (uint32_t i = 0; i < snap->n_file && !rc; i++) rc = file_segment_search(snap->file[i], q, w, m, st, sc->cache);
    for (uint32_t i = 0; i < snap->n_memory && !rc; i++) rc = memory_segment_search(snap->memory[i], q, w, m);
    return rc;
}

static int cmp_result(const void *a, const void *b)     /* compareResults (src/common.zig:169-171) */
{
    const orc_result *x = (const orc_result *)a, *y = (const orc_result *)b;
    if (x->score != y->score) return x->score > y->score ? -1 : 1;
    return x->id < y->id ? -1 : x->id > y->id;
}

/* the body of orc_search over a caller-owned hit map (cleared here, capacity retained: the reference recycles its
 * collectors through SearchResultsPool, src/common.zig:186-300) */
static int search_with_map(const orc_snapshot *snap, hit_map *m, search_scratch *sc, const uint32_t *hashes, uint32_t n,
                           uint32_t max_results, uint32_t min_score_opt, uint32_t min_score_pct,
                           orc_result *out, uint32_t out_cap, orc_stats *stats)
{
    memset(m->slots, 0, m->cap * sizeof(hit_slot));
    m->count = 0; m->has_zero = 0; m->oom = 0;
    if (stats) memset(stats, 0, sizeof *stats);
    if (run_scans(snap, hashes, n, m, stats, sc)) return -1;
    if (stats) stats->hits_unique = m->count + (size_t)m->has_zero;

    /* finish (src/common.zig:131-167) */
    uint32_t min_score = min_score_opt;
    size_t nc = 0;
    if (m->count + 2 > sc->ccap) {
        free(sc->cand);
        sc->ccap = (m->count + 2) * 2;
        sc->cand = (orc_result *)malloc(sc->ccap * sizeof(orc_result));
        if (!sc->cand) { sc->ccap = 0; return -1; }
    }
    orc_result *cand = sc->cand;
    for (size_t i = 0; i < m->cap; i++)
        if (m->slots[i].id && m->slots[i].score >= min_score) { cand[nc].id = m->slots[i].id; cand[nc].score = m->slots[i].score; nc++; }
    if (m->has_zero && m->zero.score >= min_score) { cand[nc].id = 0; cand[nc].score = m->zero.score; nc++; }
    qsort(cand, nc, sizeof(orc_result), cmp_result);

    uint32_t outn = 0;
    for (size_t i = 0; i < nc; i++) {
        if (outn == max_results) break;
        int f; hit_slot *h = map_slot(m, cand[i].id, &f);
        if (orc_snapshot_has_newer_commit(snap, cand[i].id, h->commit_id)) continue;
        if (cand[i].score < min_score) break;
        if (outn == 0) {
            /* :162; score_pct is an unclamped u32 upstream (src/server.zig:189-193): 64-bit product, saturating quotient */
            uint64_t rel64 = (uint64_t)cand[i].score * min_score_pct / 100;
            uint32_t rel = rel64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)rel64;
            if (rel > min_score) min_score = rel;
        }
        if (outn < out_cap) out[outn] = cand[i];
        outn++;
    }
    return (int)(outn < out_cap ? outn : out_cap);
}

int orc_search(const orc_snapshot *snap, const uint32_t *hashes, uint32_t n,
               uint32_t max_results, uint32_t min_score_opt, uint32_t min_score_pct,
               orc_result *out, uint32_t out_cap, orc_stats *stats)
{
    hit_map m;
    search_scratch sc;
    memset(&sc, 0, sizeof sc);
    if (map_init(&m, 1024)) return -1;
    int rc = search_with_map(snap, &m, &sc, hashes, n, max_results, min_score_opt, min_score_pct, out, out_cap, stats);
    free(m.slots);
    scratch_free(&sc);
    return rc;
}

/* ======================================================================= */
/* Many searches at once: the CPU baseline of bench.py                       */
/* ======================================================================= */
/* The reference serves searches from N OS-thread executors, each running one search at a time on the shared immutable
 * snapshot (zio.Runtime.init(.executors = .auto), src/main.zig:272-276; lock-free readers, src/Index.zig:1-6), with
 * pooled collectors (SearchResultsPool, src/common.zig:186-300).  orc_search_many does the same with pthreads:
 * `nthreads` persistent workers pull query indices off one atomic counter, each with its own recycled hit map, and keep
 * cycling over the query set until `min_seconds` have passed (every query at least once). */
typedef struct {
    const orc_snapshot *snap;
    const uint32_t *hashes; const uint64_t *offsets; uint32_t nq;
    uint32_t max_results; int has_min_score; uint32_t min_score; uint32_t min_score_pct;
    orc_result *out; uint32_t out_cap; uint32_t *out_n;
    float *lat_ms; uint64_t lat_cap;
    double min_seconds;
    struct timespec t0;
    atomic_ullong next, done;
    atomic_int stop, failed, go;
} many_job;

static double seconds_since(const struct timespec *t0)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)(t.tv_sec - t0->tv_sec) + 1e-9 * (double)(t.tv_nsec - t0->tv_nsec);
}

static void *many_worker(void *arg)
{
    many_job *j = (many_job *)arg;
    hit_map m;
    search_scratch sc;
    memset(&sc, 0, sizeof sc);
    int have_map = map_init(&m, 1024) == 0;
    orc_result *tmp = (orc_result *)malloc((j->out_cap ? j->out_cap : 1) * sizeof(orc_result));
    if (!have_map || !tmp) atomic_store(&j->failed, 1);
    while (!atomic_load_explicit(&j->go, memory_order_acquire)) sched_yield();
    while (have_map && tmp && !atomic_load_explicit(&j->stop, memory_order_relaxed)) {
        const unsigned long long i = atomic_fetch_add(&j->next, 1ull);
        const uint32_t q = (uint32_t)(i % j->nq);
        const uint32_t *h = j->hashes + j->offsets[q];
        const uint32_t n = (uint32_t)(j->offsets[q + 1] - j->offsets[q]);
        const uint32_t floor_ = j->has_min_score ? j->min_score : orc_default_min_score(n);
        const double t_a = seconds_since(&j->t0);
        /* results are kept from the first pass only (later passes recompute the same lists) */
        orc_result *dst = i < j->nq ? j->out + (size_t)q * j->out_cap : tmp;
        const int r = search_with_map(j->snap, &m, &sc, h, n, j->max_results, floor_, j->min_score_pct, dst, j->out_cap, NULL);
        const double t_b = seconds_since(&j->t0);
        if (r < 0) { atomic_store(&j->failed, 1); break; }
        if (i < j->nq) j->out_n[q] = (uint32_t)r;
        if (j->lat_ms && i < j->lat_cap) j->lat_ms[i] = (float)((t_b - t_a) * 1e3);
        atomic_fetch_add(&j->done, 1ull);
        if (t_b >= j->min_seconds && i + 1 >= j->nq) {
            /* time is up -- but the first pass must be complete: every index below nq has been handed out once
             * `next` >= nq, and the workers holding one finish it before they look at `stop` again */
            if (atomic_load(&j->next) >= j->nq) atomic_store(&j->stop, 1);
        }
    }
    if (have_map) free(m.slots);
    scratch_free(&sc);
    free(tmp);
    return NULL;
}

/* How many threads really run at once on this box (containers may cap CPU time below the visible core count): every
 * thread spins on integer work for `seconds`; returns total iterations / (iterations of one thread running alone). */
typedef struct { double seconds; struct timespec t0; atomic_int go; unsigned long long iters; } spin_job;
typedef struct { spin_job *job; unsigned long long iters; } spin_arg;

static void *spin_worker(void *p)
{
    spin_arg *a = (spin_arg *)p;
    while (!atomic_load_explicit(&a->job->go, memory_order_acquire)) sched_yield();
    uint64_t x = 0x9E3779B97F4A7C15ull, n = 0;
    do {
        for (int i = 0; i < 4096; i++) x = orc_mix64(x + (uint64_t)i);
        n += 4096;
    } while (seconds_since(&a->job->t0) < a->job->seconds);
    a->iters = n + (x == 42);
    return NULL;
}

static unsigned long long spin_run(uint32_t nthreads, double seconds)
{
    spin_job j;
    j.seconds = seconds; atomic_init(&j.go, 0);
    pthread_t *th = (pthread_t *)malloc(nthreads * sizeof(pthread_t));
    spin_arg *args = (spin_arg *)calloc(nthreads, sizeof(spin_arg));
    if (!th || !args) { free(th); free(args); return 0; }
    uint32_t started = 0;
    for (; started < nthreads; started++) {
        args[started].job = &j;
        if (pthread_create(&th[started], NULL, spin_worker, &args[started])) break;
    }
    clock_gettime(CLOCK_MONOTONIC, &j.t0);
    atomic_store_explicit(&j.go, 1, memory_order_release);
    unsigned long long total = 0;
    for (uint32_t k = 0; k < started; k++) { pthread_join(th[k], NULL); total += args[k].iters; }
    free(th); free(args);
    return total;
}

double orc_cpu_parallelism(uint32_t nthreads, double seconds)
{
    const unsigned long long one = spin_run(1, seconds);
    const unsigned long long all = spin_run(nthreads, seconds);
    return one ? (double)all / (double)one : 0.0;
}

int orc_search_many(const orc_snapshot *snap, const uint32_t *hashes, const uint64_t *offsets, uint32_t nq,
                    uint32_t max_results, int has_min_score, uint32_t min_score, uint32_t min_score_pct,
                    uint32_t nthreads, double min_seconds,
                    orc_result *out, uint32_t out_cap, uint32_t *out_n,
                    float *latency_ms, uint64_t latency_cap,
                    double *wall_seconds, uint64_t *queries_done)
{
    if (!snap || !offsets || !out_n || nq == 0 || nthreads == 0) return -1;
    many_job j;
    memset(&j, 0, sizeof j);
    j.snap = snap; j.hashes = hashes; j.offsets = offsets; j.nq = nq;
    j.max_results = max_results; j.has_min_score = has_min_score; j.min_score = min_score; j.min_score_pct = min_score_pct;
    j.out = out; j.out_cap = out_cap; j.out_n = out_n; j.lat_ms = latency_ms; j.lat_cap = latency_cap;
    j.min_seconds = min_seconds;
    atomic_init(&j.next, 0ull); atomic_init(&j.done, 0ull); atomic_init(&j.stop, 0); atomic_init(&j.failed, 0);
    pthread_t *th = (pthread_t *)malloc(nthreads * sizeof(pthread_t));
    if (!th) return -1;
    atomic_init(&j.go, 0);
    uint32_t started = 0;
    for (; started < nthreads; started++)
        if (pthread_create(&th[started], NULL, many_worker, &j)) break;
    if (started < nthreads) { atomic_store(&j.stop, 1); atomic_store(&j.failed, 1); }
    clock_gettime(CLOCK_MONOTONIC, &j.t0);
    atomic_store_explicit(&j.go, 1, memory_order_release);
    for (uint32_t k = 0; k < started; k++) pthread_join(th[k], NULL);
    const double wall = seconds_since(&j.t0);
    free(th);
    if (wall_seconds) *wall_seconds = wall;
    if (queries_done) *queries_done = (uint64_t)atomic_load(&j.done);
    return atomic_load(&j.failed) ? -1 : 0;
}

int orc_search_hits(const orc_snapshot *snap, const uint32_t *hashes, uint32_t n,
                    uint32_t *ids, uint64_t *commit_ids, uint32_t *scores, uint32_t cap)
{
    hit_map m;
    if (map_init(&m, 1024)) return -1;
    search_scratch sc;
    memset(&sc, 0, sizeof sc);
    const int rc = run_scans(snap, hashes, n, &m, NULL, &sc);
    scratch_free(&sc);
    if (rc) { free(m.slots); return -1; }
    uint32_t k = 0;
    for (size_t i = 0; i < m.cap; i++) if (m.slots[i].id) {
        if (k < cap) { ids[k] = m.slots[i].id; commit_ids[k] = m.slots[i].commit_id; scores[k] = m.slots[i].score; }
        k++;
    }
    free(m.slots);
    return (int)k;
}

/* ======================================================================= */
/* Seeded synthetic fingerprints (definition shared with the GPU builder)   */
/* ======================================================================= */

uint64_t orc_mix64(uint64_t x)      /* splitmix64 finalizer */
{
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

uint32_t orc_synth_hash(uint64_t seed, uint32_t doc, uint32_t j, int dist)
{
    uint64_t a = orc_mix64(seed + (uint64_t)doc * 0xD1B54A32D192ED03ULL);
    uint64_t r = orc_mix64(a ^ (uint64_t)j);
    if (dist == 1 && (r & 0xFFFF) < 1311) {                /* ~2 % from a 4095-value hot pool, log-uniform rank */
        uint32_t e = (uint32_t)((r >> 16) & 0xFF) % 12;
        uint32_t k = (1u << e) + ((uint32_t)(r >> 24) & ((1u << e) - 1)) - 1;
        return (uint32_t)(orc_mix64(seed ^ 0x5bd1e9955bd1e995ULL ^ ((uint64_t)k << 32)) >> 32);
    }
    return (uint32_t)(r >> 32);
}

/* ---- a whole synthetic segment's items, SORTED, on `nthreads` host threads (tests/test_gpu_fullsize.py builds one 1.6 G-item segment of the
 *      100 M index here, on the host, so that full-size parity does not only see blocks the GPU encoded itself).  Items are generated doc by
 *      doc (orc_synth_hash), dealt into 2^11 buckets by their top hash bits -- every thread a contiguous range of docs, so a bucket holds
 *      its items in ascending doc order --, and each bucket is sorted by the remaining 21 hash bits with a stable LSD radix: (hash, doc)
 *      order, what filefmt.writeBlocks is fed (src/filefmt.zig:94-138).  `out` and `tmp`: num_docs * H items each. */
typedef struct {
    uint64_t seed; uint32_t first_doc, num_docs, H; int dist;
    uint64_t *out, *tmp; uint32_t nthreads, tid;
    size_t *counts;            /* [nthreads][2048] */
    size_t *bucket_start;      /* [2049] */
    pthread_barrier_t *bar;
} synth_mt_job;

static void *synth_mt_worker(void *arg)
{
    synth_mt_job *j = (synth_mt_job *)arg;
    const uint32_t d0 = (uint32_t)((uint64_t)j->num_docs * j->tid / j->nthreads), d1 = (uint32_t)((uint64_t)j->num_docs * (j->tid + 1) / j->nthreads);
    size_t *cnt = j->counts + (size_t)j->tid * 2048;
    memset(cnt, 0, 2048 * sizeof(size_t));
    for (uint32_t d = d0; d < d1; d++)
        for (uint32_t k = 0; k < j->H; k++) cnt[orc_synth_hash(j->seed, j->first_doc + d, k, j->dist) >> 21]++;
    pthread_barrier_wait(j->bar);
    if (j->tid == 0) {                                   /* bucket b: thread 0's items, then thread 1's, ... */
        size_t sum = 0;
        for (uint32_t b = 0; b < 2048; b++) {
            j->bucket_start[b] = sum;
            for (uint32_t t = 0; t < j->nthreads; t++) { size_t c = j->counts[(size_t)t * 2048 + b]; j->counts[(size_t)t * 2048 + b] = sum; sum += c; }
        }
        j->bucket_start[2048] = sum;
    }
    pthread_barrier_wait(j->bar);
    for (uint32_t d = d0; d < d1; d++)
        for (uint32_t k = 0; k < j->H; k++) {
            const uint32_t h = orc_synth_hash(j->seed, j->first_doc + d, k, j->dist);
            j->tmp[cnt[h >> 21]++] = ((uint64_t)h << 32) | (uint64_t)(j->first_doc + d);
        }
    pthread_barrier_wait(j->bar);
    /* every 'nthreads'-th bucket: three stable passes over hash bits 0..6, 7..13, 14..20 (tmp -> out -> tmp -> out) */
    for (uint32_t b = j->tid; b < 2048; b += j->nthreads) {
        const size_t lo = j->bucket_start[b], n = j->bucket_start[b + 1] - lo;
        uint64_t *src = j->tmp + lo, *dst = j->out + lo;
        for (int pass = 0; pass < 3; pass++) {
            size_t c[128] = {0};
            const int shift = 32 + 7 * pass;
            for (size_t i = 0; i < n; i++) c[(src[i] >> shift) & 127]++;
            size_t sum = 0;
            for (int v = 0; v < 128; v++) { size_t t = c[v]; c[v] = sum; sum += t; }
            for (size_t i = 0; i < n; i++) dst[c[(src[i] >> shift) & 127]++] = src[i];
            uint64_t *t = src; src = dst; dst = t;
        }
        /* (three passes: the sorted bucket ends in `out`) */
    }
    return NULL;
}

int orc_synth_items_sorted_mt(uint64_t seed, uint32_t first_doc, uint32_t num_docs, uint32_t H, int dist, uint64_t *out, uint64_t *tmp, uint32_t nthreads)
{
    if (!out || !tmp || nthreads == 0 || nthreads > 256) return -1;
    synth_mt_job *jobs = (synth_mt_job *)calloc(nthreads, sizeof *jobs);
    pthread_t *th = (pthread_t *)malloc(nthreads * sizeof *th);
    size_t *counts = (size_t *)malloc((size_t)nthreads * 2048 * sizeof(size_t)), *bstart = (size_t *)malloc(2049 * sizeof(size_t));
    pthread_barrier_t bar;
    if (!jobs || !th || !counts || !bstart || pthread_barrier_init(&bar, NULL, nthreads)) { free(jobs); free(th); free(counts); free(bstart); return -1; }
    uint32_t started = 0;
    for (; started < nthreads; started++) {
        synth_mt_job *j = &jobs[started];
        j->seed = seed; j->first_doc = first_doc; j->num_docs = num_docs; j->H = H; j->dist = dist; j->out = out; j->tmp = tmp;
        j->nthreads = nthreads; j->tid = started; j->counts = counts; j->bucket_start = bstart; j->bar = &bar;
        if (pthread_create(&th[started], NULL, synth_mt_worker, j)) break;
    }
    int rc = started == nthreads ? 0 : -1;               /* (a thread that could not start leaves the others at the barrier: fatal, not handled) */
    for (uint32_t t = 0; t < started; t++) pthread_join(th[t], NULL);
    pthread_barrier_destroy(&bar);
    free(jobs); free(th); free(counts); free(bstart);
    return rc;
}

void orc_sort_u64(uint64_t *v, size_t n)     /* LSD radix sort, 8 x 8 bits */
{
    if (n < 2) return;
    uint64_t *tmp = (uint64_t *)malloc(n * sizeof(uint64_t));
    if (!tmp) return;
    uint64_t *src = v, *dst = tmp;
    for (int pass = 0; pass < 8; pass++) {
        size_t cnt[256] = {0};
        int shift = pass * 8;
        for (size_t i = 0; i < n; i++) cnt[(src[i] >> shift) & 0xFF]++;
        if (cnt[(src[0] >> shift) & 0xFF] == n) continue;     /* all equal in this digit */
        size_t sum = 0;
        for (int b = 0; b < 256; b++) { size_t c = cnt[b]; cnt[b] = sum; sum += c; }
        for (size_t i = 0; i < n; i++) dst[cnt[(src[i] >> shift) & 0xFF]++] = src[i];
        uint64_t *t = src; src = dst; dst = t;
    }
    if (src != v) memcpy(v, src, n * sizeof(uint64_t));
    free(tmp);
}

void orc_synth_items(uint64_t seed, uint32_t first_doc, uint32_t num_docs, uint32_t H, int dist, uint64_t *items)
{
    for (uint32_t d = 0; d < num_docs; d++)
        for (uint32_t j = 0; j < H; j++)
            items[(size_t)d * H + j] = (uint64_t)orc_synth_hash(seed, first_doc + d, j, dist) << 32 | (first_doc + d);
    orc_sort_u64(items, (size_t)num_docs * H);
}
