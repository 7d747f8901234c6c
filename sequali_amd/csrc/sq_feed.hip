/* sq_feed.hip -- the host side of FastqParser in the C ABI: the reference's buffer logic
 * (FastqParser_create_record_array, _qcmodule.c:964-1184) over pinned staging blocks.
 *
 * The reference creates a new bytes object of `read_in_size` per array, copies the incomplete
 * record the previous array ended in ("leftover") to its front, reads the rest from the file
 * object, checks the new bytes for ASCII, splits records until the buffer is exhausted, and
 * enlarges the buffer by `read_in_size` while it holds fewer than `min_records`.  An array is
 * therefore a *window* of the file: it starts where the last complete record of the previous
 * array ended and is `read_in_size` bytes long (more after enlarging, less at the end of the
 * file).  Here the file's text is read once, into a pinned block of up to 64 MiB; arrays are
 * windows of that block (no copy of the leftover: it already lies where the next array starts),
 * their metas are written once, relative to the block, which is what the device wants; a block
 * goes to HBM with one asynchronous copy from pinned memory and no copy on the host.
 *
 * What the caller (sequali_amd/_qc.py, or any binding of the C ABI) does:
 *     loop: r = sq_feeder_next(f, min, max, &a)
 *           r == SQ_FEED_MORE: p = sq_feeder_fill(f, &room); n = file.readinto(p[0:room]); sq_feeder_filled(f, n)
 *           r <  0: raise (sq_last_error)
 *           else: array a (a.n_records == 0: the file is exhausted)
 * Logical reads of the reference (`readinto` of exactly the free part of its buffer) are served
 * from what has been read ahead; the array boundaries are those of the reference for every file
 * object whose readinto() fills the buffer it is given unless the file ends (BytesIO, buffered
 * files, gzip streams).
 */
#include <algorithm>

#include "sq_common.h"
#include <time.h>

int64_t sq_split_range_ascii(const uint8_t *base, size_t start, size_t end, sq_meta *metas, size_t cap, size_t *consumed,
                             uint64_t stats[SQ_STATS_N], size_t ascii_from, int64_t *non_ascii);

namespace {

struct PinBuf { void *p; size_t bytes; bool pinned; };
/* pinned buffers are expensive to make (the pages are locked): parsers hand them back to a
   process-wide list instead of freeing them */
std::vector<PinBuf> g_pool;
/* where the feeder's time goes (scripts/exp_e2e_default_timeline.py): seconds in roll_block, in the record split,
   in fresh allocations of the pool; fresh allocations */
double g_feed_times[4] = {0, 0, 0, 0};
inline double feed_now()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

PinBuf pool_get(size_t bytes, bool want_pinned)
{
    /* the smallest buffer that is big enough (a parser's first, small block may sit in a big one: it is locked
       already; locking 64 MiB of pages takes 6-9 ms) */
    size_t best = g_pool.size();
    for (size_t i = 0; i < g_pool.size(); i++)
        if (g_pool[i].bytes >= bytes && g_pool[i].pinned == want_pinned && (best == g_pool.size() || g_pool[i].bytes < g_pool[best].bytes))
            best = i;
    if (best < g_pool.size()) {
        PinBuf b = g_pool[best];
        g_pool.erase(g_pool.begin() + (long)best);
        return b;
    }
    PinBuf b{nullptr, bytes, false};
    const double t0 = feed_now();
    g_feed_times[3] += 1;
    if (want_pinned && hipHostMalloc(&b.p, bytes, hipHostMallocDefault) == hipSuccess) { b.pinned = true; g_feed_times[2] += feed_now() - t0; return b; }
    (void)hipGetLastError();
    b.p = malloc(bytes);
    g_feed_times[2] += feed_now() - t0;
    return b;
}
void pool_put(PinBuf b)
{
    if (!b.p) return;
    size_t held = 0;
    for (const PinBuf &x : g_pool) held += x.bytes;
    if (held + b.bytes > (1u << 30)) {   /* two parsers' worth of 64 MiB blocks: the blocks of one that has just ended often come back after the next one has started */
        if (b.pinned) (void)hipHostFree(b.p); else free(b.p);
        return;
    }
    g_pool.push_back(b);
}

struct FeedBlock {
    PinBuf text{}, meta{};
    size_t cap = 0, meta_cap = 0;
    size_t used = 0;        /* bytes of file text in the block */
    size_t sealed_bytes = 0; /* bytes the block's records cover (set when it is sealed) */
    size_t n_records = 0;
    uint64_t id = 0;
    bool sealed = false;
    hipEvent_t copied = nullptr;   /* the upload of the block has left pinned memory */
    bool in_flight = false;
    uint64_t stats[SQ_STATS_N] = {};   /* bases, longest read, longest name, longest record span, ~(shortest read) */
    uint8_t *pin() const { return (uint8_t *)text.p; }
    sq_meta *metas() const { return (sq_meta *)meta.p; }
};

}  // namespace

struct sq_feeder {
    sq_ctx *ctx = nullptr;
    size_t read_in = 0, block_bytes = 0;
    std::vector<FeedBlock *> blocks;   /* the open one last; sealed ones until they are released */
    uint64_t next_id = 1;
    size_t blocks_made = 0;
    bool file_eof = false;
    size_t pos = 0;           /* start of the next array in the open block */
    size_t logical_end = 0;   /* where the reference's buffer of the previous array ended */
    /* the array being assembled (sq_feeder_next returned SQ_FEED_MORE in the middle of it) */
    bool in_array = false, first = true, arr_eof = false;
    size_t arr_len = 0;       /* bytes of the reference's buffer so far */
    size_t arr_first_record = 0;
    size_t need = 0;          /* bytes sq_feeder_fill must have room for */
};

namespace {

FeedBlock *open_block(sq_feeder *f) { return f->blocks.empty() || f->blocks.back()->sealed ? nullptr : f->blocks.back(); }

FeedBlock *new_block(sq_feeder *f, size_t min_bytes)
{
    /* the first block is small: a parser over a few records should not lock 64 MiB of pages */
    size_t cap = f->blocks_made == 0 ? std::max<size_t>((size_t)8 << 20, 4 * f->read_in) : f->block_bytes;
    cap = std::max(std::min(cap, std::max(f->block_bytes, (size_t)1 << 20)), min_bytes);
    f->blocks_made++;
    FeedBlock *b = new FeedBlock();
    b->text = pool_get(cap + 64, f->ctx != nullptr);
    b->cap = b->text.p ? cap : 0;
    b->meta_cap = cap / 96 + 1024;
    b->meta = pool_get(b->meta_cap * sizeof(sq_meta), f->ctx != nullptr);
    if (!b->text.p || !b->meta.p) { pool_put(b->text); pool_put(b->meta); delete b; return nullptr; }
    b->id = f->next_id++;
    b->stats[4] = 0;
    f->blocks.push_back(b);
    return b;
}

void free_block(FeedBlock *b)
{
    if (b->in_flight && b->copied) (void)hipEventSynchronize(b->copied);
    if (b->copied) (void)hipEventDestroy(b->copied);
    pool_put(b->text);
    pool_put(b->meta);
    delete b;
}

/* closes the open block behind its last complete array and carries what lies behind (the
   leftover and what was read ahead) over to a new one that has room for `room` more bytes */
int roll_block(sq_feeder *f, size_t room)
{
    FeedBlock *o = open_block(f);
    const size_t carry = o ? o->used - f->pos : 0;
    if (o) {
        o->sealed = true;
        o->sealed_bytes = f->pos;
    }
    FeedBlock *n = new_block(f, carry + room);
    if (!n) { sq_set_error("out of memory for a staging block"); return SQ_ERR_MEMORY; }
    if (carry) memcpy(n->pin(), o->pin() + f->pos, carry);
    n->used = carry;
    f->logical_end -= o ? f->pos : 0;
    f->pos = 0;
    if (o && o->n_records == 0) {   /* nothing in it: not a block anyone will ask for */
        f->blocks.erase(std::find(f->blocks.begin(), f->blocks.end(), o));
        free_block(o);
    }
    return SQ_OK;
}

size_t count_newlines(const uint8_t *p, size_t n, size_t stop_at)
{
    size_t c = 0;
    const uint8_t *e = p + n;
    while (c < stop_at && p < e) {
        const uint8_t *nl = (const uint8_t *)memchr(p, '\n', e - p);
        if (!nl) break;
        c++;
        p = nl + 1;
    }
    return c;
}

}  // namespace

/* FastqParser__new__, _qcmodule.c:905-945.  ctx may be NULL (no device: the blocks are plain
 * memory and cannot be uploaded; the host parser itself needs no GPU). */
SQ_EXPORT sq_feeder *sq_feeder_new(sq_ctx *ctx, size_t read_in_size, size_t block_bytes)
{
    if (read_in_size < 1) { sq_set_error("initial_buffersize must be at least 1, got %zu", read_in_size); return nullptr; }
    sq_feeder *f = new sq_feeder();
    f->ctx = ctx;
    f->read_in = read_in_size;
    f->block_bytes = block_bytes ? block_bytes : ((size_t)64 << 20);
    return f;
}

SQ_EXPORT void sq_feeder_free(sq_feeder *f)
{
    if (!f) return;
    for (FeedBlock *b : f->blocks) free_block(b);
    delete f;
}

/* Where the next bytes of the file go and how many of them fit (*room >= 1). */
SQ_EXPORT uint8_t *sq_feeder_fill(sq_feeder *f, size_t *room)
{
    FeedBlock *b = open_block(f);
    /* read ahead in pieces of 16 buffers (1 MiB at least): a block that is sealed early -- a
       getter asked for its records -- carries little over to the next one */
    const size_t piece = std::max<size_t>(std::max(f->need, 16 * f->read_in), (size_t)1 << 20);
    if (!b || b->cap - b->used < f->need || (b->cap - b->used < piece && !f->in_array && f->pos > 0 && b->cap - f->pos < f->block_bytes / 2)) {
        if (roll_block(f, piece) != SQ_OK) { *room = 0; return nullptr; }
        b = open_block(f);
    }
    *room = std::min(piece, b->cap - b->used);
    return b->pin() + b->used;
}

/* n bytes were put where sq_feeder_fill pointed; 0: the file has ended. */
SQ_EXPORT int sq_feeder_filled(sq_feeder *f, size_t n)
{
    FeedBlock *b = open_block(f);
    if (!b || n > b->cap - b->used) { sq_set_error("sq_feeder_filled: more bytes than there was room for"); return SQ_ERR_VALUE; }
    if (n == 0) f->file_eof = true;
    b->used += n;
    return SQ_OK;
}

/* The next record array (FastqParser_create_record_array, :964-1184; FastqParser__next__ :1201
 * calls it with min 1, max SIZE_MAX; FastqParser_read :1213 with n, n).  SQ_FEED_MORE (1): call
 * sq_feeder_fill / _filled and come back; 0: `out` is the array (n_records == 0: end of file);
 * < 0: the reference's exception in sq_last_error().  byte_start / byte_len: the window of the
 * block the reference's buffer object would hold; the metas of its records are
 * sq_feeder_block_metas(block_id) + first_record, their record_start relative to the block. */
SQ_EXPORT int sq_feeder_next(sq_feeder *f, size_t min_records, size_t max_records, sq_feed_array *out)
{
    FeedBlock *b = open_block(f);
    if (!b) {
        if (roll_block(f, f->read_in) != SQ_OK) return SQ_ERR_MEMORY;
        b = open_block(f);
    }
    if (!f->in_array) {
        f->in_array = true;
        f->first = true;
        f->arr_eof = false;
        f->arr_len = f->logical_end - f->pos;   /* the leftover of the previous array */
        f->arr_first_record = b->n_records;
    }
    auto non_ascii_error = [&](int64_t at) {   /* at: offset in the block of a byte >= 0x80 among the new bytes, < 0: none */
        if (at < 0 || (size_t)at < f->pos) return false;
        sq_set_error("Found non-ASCII character in file: %c", b->pin()[at]);
        f->in_array = false;
        return true;
    };
    for (;;) {
        size_t fresh_from = (size_t)-1;   /* offset in the block of the bytes this round has read */
        /* one readinto of the reference: the free part of a new buffer of read_in bytes, later read_in more */
        const size_t want = f->first ? (f->read_in > f->arr_len ? f->read_in - f->arr_len : 0) : f->read_in;
        if (want > 0) {
            const size_t have = b->used - (f->pos + f->arr_len);
            if (have < want && !f->file_eof) {
                if (b->cap - (f->pos + f->arr_len) < want) {
                    /* the block ends inside this array: its arrays so far are sealed, this one starts the next block */
                    if (f->arr_first_record != b->n_records) {   /* records of an earlier round of this array: split again over there */
                        b->n_records = f->arr_first_record;
                    }
                    const size_t keep_len = f->arr_len;
                    const double t_roll = feed_now();
                    int rc = roll_block(f, keep_len + want + f->read_in);
                    g_feed_times[0] += feed_now() - t_roll;
                    if (rc) { f->in_array = false; return rc; }
                    b = open_block(f);
                    f->arr_first_record = 0;
                    f->logical_end = 0;   /* of no use until this array is done */
                    f->arr_len = keep_len;
                }
                f->need = want - have;
                return SQ_FEED_MORE;
            }
            const size_t got = std::min(want, have);
            if (got == 0) f->arr_eof = true;
            /* :1055-1067: the bytes just read are checked for ASCII before anything else is looked at.
               The check rides on the newline scan of the record split below (fresh_from) */
            fresh_from = got ? f->pos + f->arr_len : (size_t)-1;
            f->arr_len += got;
        }
        f->first = false;
        f->need = 0;
        const uint8_t *buf = b->pin() + f->pos;
        if (f->arr_len == 0) break;   /* :1069 the entire file is read */
        if (f->arr_eof && count_newlines(buf, f->arr_len, 4) < 4) {   /* :1073-1081 buffer_contains_fastq */
            if (fresh_from != (size_t)-1) {
                const int64_t at = sq_first_non_ascii(b->pin() + fresh_from, f->pos + f->arr_len - fresh_from);
                if (at >= 0 && non_ascii_error(at + (int64_t)fresh_from)) return SQ_ERR_VALUE;
            }
            std::string s((const char *)buf, f->arr_len);
            sq_set_error("Incomplete record at the end of file %s", s.c_str());
            f->in_array = false;
            return SQ_ERR_EOF;
        }
        /* the records of the buffer (all of them again when the buffer was enlarged: the earlier
           round's metas are overwritten with the same values) */
        b->n_records = f->arr_first_record;
        size_t consumed = 0;
        int64_t n;
        uint64_t stats[SQ_STATS_N];
        for (;;) {
            const size_t cap = std::min(b->meta_cap - b->n_records, max_records);
            memcpy(stats, b->stats, sizeof stats);
            int64_t bad = -1;
            const double t_split = feed_now();
            n = sq_split_range_ascii(b->pin(), f->pos, f->pos + f->arr_len, b->metas() + b->n_records, cap, &consumed, stats, fresh_from, &bad);
            g_feed_times[1] += feed_now() - t_split;
            if (non_ascii_error(bad)) return SQ_ERR_VALUE;
            if (n < 0) { f->in_array = false; return (int)n; }
            if ((size_t)n < cap || (size_t)n == max_records) break;
            /* the meta area is full: a bigger one (the block keeps its text) */
            PinBuf bigger = pool_get(2 * b->meta_cap * sizeof(sq_meta), f->ctx != nullptr);
            if (!bigger.p) { sq_set_error("out of memory for the record table"); f->in_array = false; return SQ_ERR_MEMORY; }
            memcpy(bigger.p, b->meta.p, b->n_records * sizeof(sq_meta));
            pool_put(b->meta);
            b->meta = bigger;
            b->meta_cap *= 2;
        }
        if ((size_t)n >= min_records || f->arr_eof) {
            if (f->arr_eof && n == 0) {
                std::string s((const char *)buf, f->arr_len);
                sq_set_error("Incomplete record at the end of file %s", s.c_str());
                f->in_array = false;
                return SQ_ERR_EOF;
            }
            memcpy(b->stats, stats, sizeof stats);
            out->block_id = b->id;
            out->byte_start = f->pos;
            out->byte_len = f->arr_len;
            out->first_record = b->n_records;
            out->n_records = (uint64_t)n;
            b->n_records += (size_t)n;
            f->logical_end = f->pos + f->arr_len;
            f->pos += consumed;
            f->in_array = false;
            return SQ_OK;
        }
    }
    out->block_id = b->id;
    out->byte_start = f->pos;
    out->byte_len = 0;
    out->first_record = b->n_records;
    out->n_records = 0;
    f->in_array = false;
    return SQ_OK;
}

/* The block that is still open is closed behind its last array (what it holds of later arrays
 * moves on to a new block).  Needed before its records can be uploaded. */
SQ_EXPORT int sq_feeder_seal(sq_feeder *f)
{
    FeedBlock *b = open_block(f);
    if (!b || f->in_array) return SQ_OK;
    if (b->n_records == 0) return SQ_OK;
    return roll_block(f, f->read_in);
}

namespace {
FeedBlock *find_block(sq_feeder *f, uint64_t id)
{
    for (FeedBlock *b : f->blocks)
        if (b->id == id) return b;
    sq_set_error("staging block %llu is gone", (unsigned long long)id);
    return nullptr;
}
}  // namespace

SQ_EXPORT const uint8_t *sq_feeder_block_text(sq_feeder *f, uint64_t block_id)
{
    FeedBlock *b = find_block(f, block_id);
    return b ? b->pin() : nullptr;
}
SQ_EXPORT const sq_meta *sq_feeder_block_metas(sq_feeder *f, uint64_t block_id)
{
    FeedBlock *b = find_block(f, block_id);
    return b ? b->metas() : nullptr;
}
SQ_EXPORT uint64_t sq_feeder_block_records(sq_feeder *f, uint64_t block_id)
{
    FeedBlock *b = find_block(f, block_id);
    return b ? b->n_records : 0;
}
SQ_EXPORT uint64_t sq_feeder_block_bytes(sq_feeder *f, uint64_t block_id)
{
    FeedBlock *b = find_block(f, block_id);
    return b ? b->used : 0;
}
SQ_EXPORT int sq_feeder_block_is_open(sq_feeder *f, uint64_t block_id)
{
    FeedBlock *b = find_block(f, block_id);
    return b && !b->sealed;
}

/* A sealed block as a record array in HBM: one asynchronous copy of its text and one of its
 * metas from pinned memory on the context's stream (add_record_array's staging copy, SURVEY 8b
 * "ownership": the caller's array is borrowed for the call only; here the block IS the copy). */
SQ_EXPORT sq_batch *sq_feeder_upload(sq_feeder *f, uint64_t block_id)
{
    FeedBlock *fb = find_block(f, block_id);
    if (!fb) return nullptr;
    if (!fb->sealed) { sq_set_error("sq_feeder_upload: block %llu is still open", (unsigned long long)block_id); return nullptr; }
    if (!f->ctx) { sq_set_error("sq_feeder_upload: the parser was made without a device context"); return nullptr; }
    sq_ctx *ctx = f->ctx;
    sq_batch *b = new sq_batch();
    b->ctx = ctx;
    b->buf_len = fb->sealed_bytes;
    b->n = fb->n_records;
    b->owns = b->slack = true;
    b->total_bases = fb->stats[0];
    b->max_length = fb->stats[1];
    b->max_name_length = fb->stats[2];
    b->max_record_span = fb->stats[3];
    b->min_length = fb->n_records ? ~fb->stats[4] : 0;
    b->len_hist.resize(SQ_LEN_BINS);
    for (int i = 0; i < SQ_LEN_BINS; i++) b->len_hist[i] = (uint32_t)fb->stats[5 + i];
    if (hipMalloc((void **)&b->d_buf, b->buf_len + 64) != hipSuccess ||
        hipMalloc((void **)&b->d_metas, (b->n ? b->n : 1) * sizeof(sq_meta)) != hipSuccess) {
        sq_set_error("sq_feeder_upload: out of device memory");
        if (b->d_buf) (void)hipFree(b->d_buf);
        delete b;
        return nullptr;
    }
    auto fail = [&](const char *what, hipError_t e) -> sq_batch * {
        sq_set_error("sq_feeder_upload: %s: %s", what, hipGetErrorString(e));
        (void)hipStreamSynchronize(ctx->stream);   /* nothing may still be writing the blocks that go back */
        if (b->ready) (void)hipEventDestroy(b->ready);
        (void)hipFree(b->d_buf);
        (void)hipFree(b->d_metas);
        delete b;
        return nullptr;
    };
    hipError_t e = hipSuccess;
    if (b->buf_len && (e = hipMemcpyAsync(b->d_buf, fb->pin(), b->buf_len, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) return fail("text upload", e);
    if ((e = hipMemsetAsync(b->d_buf + b->buf_len, 0, 64, ctx->stream)) != hipSuccess) return fail("padding", e);
    if (b->n && (e = hipMemcpyAsync(b->d_metas, fb->metas(), b->n * sizeof(sq_meta), hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) return fail("meta upload", e);
    fb->in_flight = true;   /* from here on the pinned block may be read by a copy: free_block waits (for the event, or the stream) */
    if (!fb->copied && (e = hipEventCreateWithFlags(&fb->copied, hipEventDisableTiming)) != hipSuccess) return fail("event", e);
    if ((e = hipEventRecord(fb->copied, ctx->stream)) != hipSuccess) return fail("event", e);
    if ((e = hipEventCreateWithFlags(&b->ready, hipEventDisableTiming)) != hipSuccess) return fail("event", e);
    if ((e = hipEventRecord(b->ready, ctx->stream)) != hipSuccess) return fail("event", e);
    return b;
}

/* The caller no longer needs the host copy of a sealed block (its pinned memory goes back to the
 * pool once the upload has left it). */
SQ_EXPORT void sq_feeder_debug_times(double *out, int reset)
{
    for (int i = 0; i < 4; i++) { out[i] = g_feed_times[i]; if (reset) g_feed_times[i] = 0; }
}

SQ_EXPORT void sq_feeder_release(sq_feeder *f, uint64_t block_id)
{
    for (size_t i = 0; i < f->blocks.size(); i++)
        if (f->blocks[i]->id == block_id && f->blocks[i]->sealed) {
            free_block(f->blocks[i]);
            f->blocks.erase(f->blocks.begin() + i);
            return;
        }
}

/* Page-locked host memory for callers that keep FASTQ text on the host and want the uploads to
 * run at the bus rate (sequali_amd.PinnedReader; scripts/bench_e2e.py).  Plain memory when no
 * device is there. */
SQ_EXPORT void *sq_host_alloc(size_t bytes, int *pinned)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess) {
        if (pinned) *pinned = 1;
        return p;
    }
    (void)hipGetLastError();
    if (pinned) *pinned = 0;
    return malloc(bytes ? bytes : 1);
}
SQ_EXPORT void sq_host_free(void *p, int pinned)
{
    if (!p) return;
    if (pinned) (void)hipHostFree(p); else free(p);
}
