/* sq_feed.hip -- the host side of FastqParser in the C ABI: the reference's buffer logic
 * (FastqParser_create_record_array, _qcmodule.c:964-1184) over pinned staging blocks.
 *
 * The reference creates a new bytes object of `read_in_size` per array, copies the incomplete
 * record the previous array ended in ("leftover") to its front, reads the rest from the file
 * object, checks the new bytes for ASCII, splits records until the buffer is exhausted, and
 * enlarges the buffer by `read_in_size` while it holds fewer than `min_records`.  An array is
 * therefore a *window* of the file: it starts where the last complete record of the previous
 * array ended and is `read_in_size` bytes long (more after enlarging, less at the end of the
 * file).  Here the file's text is read once, into a pinned block of up to 128 MiB; arrays are
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
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "sq_common.h"
#include <time.h>
#include <unistd.h>

int64_t sq_split_range_ascii(const uint8_t *base, size_t start, size_t end, sq_meta *metas, size_t cap, size_t *consumed,
                             uint64_t stats[SQ_STATS_N], size_t ascii_from, int64_t *non_ascii);
int64_t sq_split_range_indexed(const uint8_t *base, size_t start, size_t end_off, sq_meta *metas, size_t cap, size_t *consumed,
                               uint64_t stats[SQ_STATS_N], size_t ascii_from, int64_t *non_ascii, const SqNlPiece *pieces, size_t n_pieces);
size_t sq_scan_newlines(const uint8_t *p, size_t n, uint32_t base, uint32_t *out, size_t cap, size_t *scanned, uint32_t *high);   /* sq_hostsimd.cpp */
int64_t sq_first_non_ascii_fast(const uint8_t *p, size_t n);
size_t sq_copy_scan_newlines(uint8_t *dst, const uint8_t *src, size_t n, uint32_t base, uint32_t *out, uint8_t *after, size_t cap, size_t *copied,
                             uint32_t *high);   /* sq_hostsimd.cpp */

namespace {

struct PinBuf { void *p; size_t bytes; bool pinned; };
/* pinned buffers are expensive to make (the pages are locked): parsers hand them back to a
   process-wide list instead of freeing them */
std::vector<PinBuf> g_pool;
/* where the feeder's time goes (scripts/exp_e2e_default_timeline.py): seconds in roll_block, in the record split,
   in fresh allocations of the pool; fresh allocations */
double g_feed_times[4] = {0, 0, 0, 0};
/* seconds the caller's thread waited for text and for the walker; seconds the walker and the workers were at work (sq_feeder_debug_waits) */
std::atomic<uint64_t> g_feed_ns[4];
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
    if (held + b.bytes > ((size_t)2 << 30)) {   /* two parsers' worth of 128 MiB blocks (and their metas): the blocks of one that has just ended often come back after the next one has started */
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
    /* a feeder with a source of its own (sq_feeder_set_source_*): what its workers noted of the stretches they copied in,
       in order of offset, back to back from 0; `used` is the end of the longest run of finished ones.  Touched under
       sq_feeder::mu (a finished piece itself does not change any more) */
    struct Piece {
        size_t from = 0, to = 0;
        std::vector<uint32_t> nl;
        std::vector<uint8_t> after;   /* the byte behind each newline (0: it lies behind the piece, or is a zero byte) */
        uint32_t first_high = UINT32_MAX;
        bool done = false;
    };
    std::vector<Piece *> pieces;
    size_t reserved = 0;    /* end of the last piece handed to a worker */
    /* the block's place in HBM, taken when the block is opened (a feeder with a source and a device): every piece goes up
       as soon as its worker has it, the metas every 64 K records, on the feeder's copy stream -- when the block is sealed
       all but its tail is there already (early_bad: a copy failed, sq_feeder_upload sends everything again) */
    /* the walker (feed_walker): the records of the block's text, split once, ahead of the caller's loop -- metas()[0 ..
       walk_n) are written, every newline below walk_scanned has been looked at; walk_stopped: it gave up in front of
       record walk_n, which starts at walk_next (a malformed record, the meta area full): sq_feeder_next splits that window
       itself.  The stats of the walked records are kept per 4096 of them (the block is sealed behind any record) */
    std::atomic<size_t> walk_n{0}, walk_scanned{0};
    std::atomic<bool> walk_stopped{false};
    std::atomic<uint32_t> high_min{UINT32_MAX};   /* the lowest first_high of the finished pieces */
    size_t walk_next = 0;
    struct Chunk { uint64_t s[SQ_STATS_N]; };
    std::vector<Chunk> chunks;
    bool walked = false;     /* arrays were handed out from the walker's metas: `stats` does not hold them yet */
    bool walk_off = false;   /* the caller's thread splits this block's remaining windows itself */
    sq_ctx *ctx = nullptr;
    hipStream_t early_stream = nullptr;
    uint8_t *d_text = nullptr;
    size_t sent_to = 0;      /* the walker sends: the text in front of this offset is on its way (it belongs to the walker while the block is open) */
    sq_meta *d_metas = nullptr;
    size_t metas_sent = 0;
    std::atomic<bool> early_bad{false};
    ~FeedBlock() { for (Piece *p : pieces) delete p; }
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
    /* A source the feeder reads by itself (sq_feeder_set_source_memory / _fd): worker threads copy the next stretches of
       it into the open block, as far as the block has room, and note the newlines and the first byte >= 0x80 of what
       they copy (FeedBlock::Piece); the record split takes its newlines from there (sq_split_range_indexed) and
       sq_feeder_next never asks the caller for bytes.  Round 6: of the 85-95 ms a pass over 2 M reads through the
       reference's call pattern took, 21-24 were the caller's readinto and 34-47 the newline scan, both on the one thread
       that also runs the Python loop. */
    bool has_source = false;
    const uint8_t *src_mem = nullptr;
    int src_fd = -1;
    uint64_t src_off = 0, src_end = 0;   /* the next byte to hand to a worker; where the source ends */
    bool src_failed = false;
    std::mutex mu;
    std::condition_variable cv_work, cv_data;
    std::vector<std::thread> workers;
    int busy = 0;             /* workers copying right now */
    bool stop = false, paused = false;
    bool walk_on = false;     /* a walker thread splits the records ahead of sq_feeder_next */
    std::thread walker;
    hipStream_t copy_stream = nullptr;   /* the early uploads (FeedBlock::d_text); null: blocks go up when they are sealed */
    hipEvent_t copy_done = nullptr;
    bool want_early = false;    /* somebody will ask for the blocks in HBM (sq_feeder_expect_uploads, or the first sq_feeder_upload): a parser that is only iterated uploads nothing */
    bool walk_plain = false;    /* (SQ_FEED_WALKER=plain: every record by the record loop, none by the fast lane) */
    bool plain_copy = false;    /* (SQ_FEED_COPY=plain: memcpy, then the scan over the block -- the way before the fused pass) */
    bool walker_sends = false;  /* the walker issues the early copies, stretch by stretch as the text arrives (else the workers, piece by piece) */
};

namespace {

FeedBlock *open_block(sq_feeder *f) { return f->blocks.empty() || f->blocks.back()->sealed ? nullptr : f->blocks.back(); }

constexpr size_t FEED_PIECE = (size_t)1 << 20;   /* what a worker copies at a time */
constexpr size_t FEED_EARLY_METAS = (size_t)1 << 16;   /* records whose metas go up together */

/* a worker of a feeder with a source: the next stretch of the source into the open block, its newlines noted */
void feed_worker(sq_feeder *f)
{
    if (f->copy_stream) (void)hipSetDevice(f->ctx->device);
    std::unique_lock<std::mutex> lk(f->mu);
    for (;;) {
        FeedBlock *b = nullptr;
        size_t n = 0;
        for (;;) {
            if (f->stop) return;
            b = f->paused ? nullptr : open_block(f);
            if (b && !f->src_failed && f->src_off < f->src_end && b->reserved < b->cap) {
                n = (size_t)std::min<uint64_t>({(uint64_t)FEED_PIECE, (uint64_t)(b->cap - b->reserved), f->src_end - f->src_off});
                break;
            }
            f->cv_work.wait(lk);
        }
        FeedBlock::Piece *pc = new FeedBlock::Piece();
        pc->from = b->reserved;
        pc->to = b->reserved + n;
        b->reserved += n;
        b->pieces.push_back(pc);
        const uint64_t at = f->src_off;
        f->src_off += n;
        f->busy++;
        uint8_t *const d_text = f->walker_sends ? nullptr : b->d_text;
        lk.unlock();
        const double t_work = feed_now();
        uint8_t *dst = b->pin() + pc->from;
        bool ok = true;
        if (f->src_mem && !f->plain_copy) {   /* copy and scan in one pass over the source (streaming stores: nobody reads the block through the cache) */
            pc->nl.reserve(n / 64 + 16);
            pc->after.reserve(n / 64 + 16);
            size_t done = 0;
            while (done < n) {
                uint32_t tmp[2048], high = UINT32_MAX;
                uint8_t tmp_after[2048];
                size_t copied = 0;
                const size_t k = sq_copy_scan_newlines(dst + done, f->src_mem + at + done, n - done, (uint32_t)(pc->from + done), tmp, tmp_after, 2048, &copied, &high);
                pc->nl.insert(pc->nl.end(), tmp, tmp + k);
                pc->after.insert(pc->after.end(), tmp_after, tmp_after + k);
                if (high != UINT32_MAX && pc->first_high == UINT32_MAX) pc->first_high = high;
                done += copied;
                if (!copied) break;
            }
        } else {
            size_t got = f->src_mem ? n : 0;
            if (f->src_mem) memcpy(dst, f->src_mem + at, n);
            while (got < n) {
                const ssize_t r = pread(f->src_fd, dst + got, n - got, (off_t)(at + got));
                if (r <= 0) { ok = false; break; }   /* the file is shorter than it was, or an I/O error */
                got += (size_t)r;
            }
            if (ok) {
                pc->nl.reserve(n / 64 + 16);
                size_t done = 0;
                while (done < n) {
                    uint32_t tmp[2048], high = UINT32_MAX;
                    size_t scanned = 0;
                    const size_t k = sq_scan_newlines(dst + done, n - done, (uint32_t)(pc->from + done), tmp, 2048, &scanned, &high);
                    pc->nl.insert(pc->nl.end(), tmp, tmp + k);
                    if (high != UINT32_MAX && pc->first_high == UINT32_MAX) pc->first_high = high;
                    done += scanned;
                    if (!scanned) break;
                }
                pc->after.resize(pc->nl.size());
                const uint8_t *text = b->pin();
                for (size_t i = 0; i < pc->nl.size(); i++) pc->after[i] = (size_t)pc->nl[i] + 1 < pc->to ? text[pc->nl[i] + 1] : 0;
            }
        }
        bool sent = true;
        if (ok && d_text && hipMemcpyAsync(d_text + pc->from, dst, n, hipMemcpyHostToDevice, b->early_stream) != hipSuccess) {
            (void)hipGetLastError();
            sent = false;
        }
        g_feed_ns[3] += (uint64_t)(1e9 * (feed_now() - t_work));
        lk.lock();
        f->busy--;
        if (!sent) b->early_bad = true;
        if (!ok) f->src_failed = true;
        pc->done = true;
        if (pc->first_high < b->high_min.load(std::memory_order_relaxed)) b->high_min.store(pc->first_high, std::memory_order_relaxed);
        /* the text is there up to the end of the longest run of finished pieces */
        size_t used = b->used;
        for (FeedBlock::Piece *q : b->pieces) {
            if (q->to <= used) continue;
            if (!q->done || q->from != used) break;
            used = q->to;
        }
        b->used = used;
        f->cv_data.notify_all();
    }
}

/* nobody copies into the open block any more until feed_resume(): its `used` and `reserved` agree */
void feed_pause(sq_feeder *f, std::unique_lock<std::mutex> &lk)
{
    f->paused = true;
    while (f->busy) f->cv_data.wait(lk);
}
void feed_resume(sq_feeder *f)
{
    f->paused = false;
    f->cv_work.notify_all();
    f->cv_data.notify_all();   /* the walker waits there */
}

constexpr size_t WALK_CHUNK = 4096;   /* records whose stats are kept together */

/* The walker's fast lane: the well-formed records from *at on whose four newlines lie in ONE piece and in front of `upto`,
   straight from that piece's notes (four offsets and two of the bytes behind them per record; the text is looked at once
   per call).  Returns how many (<= cap) and moves *at behind them.  It stops in front of anything else -- a record that
   crosses into the next piece, a record whose `@`, `+` or lengths are not right, a byte the notes do not hold -- and the
   record loop itself (sq_split_range_indexed) takes that one: every error is still its error. */
size_t walk_fast(const uint8_t *text, const SqNlPiece *pieces, size_t n_pieces, size_t *at, size_t upto, sq_meta *metas, size_t cap, uint64_t *S)
{
    size_t rec = *at, n = 0, k = 0;
    while (k < n_pieces && pieces[k].to <= rec) k++;
    if (k == n_pieces || rec < pieces[k].from || rec >= upto) return 0;
    const SqNlPiece &pc = pieces[k];
    size_t j = (size_t)(std::lower_bound(pc.nl, pc.nl + pc.n_nl, (uint32_t)rec) - pc.nl);
    uint8_t first = text[rec];
    while (n < cap && j + 3 < pc.n_nl) {
        const uint32_t e0 = pc.nl[j], e1 = pc.nl[j + 1], e2 = pc.nl[j + 2], e3 = pc.nl[j + 3];
        if (e3 >= upto || first != '@' || pc.after[j + 1] != '+') break;
        const uint32_t name = (uint32_t)rec + 1, seq = e0 + 1, qual = e2 + 1, L = e1 - seq;
        if (L != e3 - qual) break;
        sq_meta *m = &metas[n++];
        m->record_start = name;
        m->name_length = e0 - name;
        m->sequence_offset = seq - name;
        m->sequence_length = L;
        m->qualities_offset = qual - name;
        m->tags_offset = e3 - name;
        m->tags_length = 0;
        m->accumulated_error_rate = 0.0;
        const uint64_t span = (uint64_t)m->qualities_offset + L;
        S[0] += L;
        if (L > S[1]) S[1] = L;
        if (m->name_length > S[2]) S[2] = m->name_length;
        if (span > S[3]) S[3] = span;
        if (~(uint64_t)L > S[4]) S[4] = ~(uint64_t)L;
        S[5 + (L < (uint32_t)SQ_LEN_BINS - 1 ? L : (uint32_t)SQ_LEN_BINS - 1)]++;
        rec = (size_t)e3 + 1;
        first = pc.after[j + 3];
        j += 4;
        if (first == 0) {   /* behind the piece, or a zero byte: the text knows (as far as it is there) */
            if (rec >= upto) break;
            first = text[rec];
        }
    }
    *at = rec;
    return n;
}

/* the walker of a feeder with a source: the records of the open block as far as its text is there, in the block's meta
   area, by the record loop sq_feeder_next itself uses (sq_split_range_indexed over the workers' notes) */
void feed_walker(sq_feeder *f)
{
    if (f->copy_stream) (void)hipSetDevice(f->ctx->device);
    std::unique_lock<std::mutex> lk(f->mu);
    std::vector<SqNlPiece> idx;
    for (;;) {
        FeedBlock *b = nullptr;
        bool walk = false, send = false;
        for (;;) {
            if (f->stop) return;
            b = f->paused ? nullptr : open_block(f);
            if (b) {
                walk = !b->walk_off && !b->walk_stopped.load(std::memory_order_relaxed) && b->walk_scanned.load(std::memory_order_relaxed) < b->used;
                send = f->walker_sends && b->d_text && !b->early_bad && b->sent_to < b->used;
                if (walk || send) break;
            }
            f->cv_data.wait(lk);
        }
        const size_t upto = b->used;
        idx.clear();
        if (walk)
            for (const FeedBlock::Piece *q : b->pieces)
                if (q->done && q->to > b->walk_next && q->from < upto) idx.push_back(SqNlPiece{q->from, q->to, q->nl.data(), q->after.data(), q->nl.size(), q->first_high});
        uint8_t *const d_text = b->d_text;
        f->busy++;
        lk.unlock();
        if (send) {   /* first: the copy runs while the records are split */
            if (hipMemcpyAsync(d_text + b->sent_to, b->pin() + b->sent_to, upto - b->sent_to, hipMemcpyHostToDevice, b->early_stream) != hipSuccess) {
                (void)hipGetLastError();
                b->early_bad = true;
            }
            b->sent_to = upto;
        }
        if (!walk) {
            lk.lock();
            f->busy--;
            f->cv_data.notify_all();
            continue;
        }
        const double t_walk = feed_now();
        size_t n = b->walk_n.load(std::memory_order_relaxed), at = b->walk_next;
        bool stop = false;
        for (;;) {
            const size_t chunk = n / WALK_CHUNK;
            if (b->chunks.size() <= chunk) b->chunks.resize(chunk + 1, FeedBlock::Chunk{});
            const size_t cap = std::min(b->meta_cap - n, WALK_CHUNK - n % WALK_CHUNK);
            if (cap == 0) { stop = true; break; }   /* the meta area is full: the caller's thread makes a bigger one */
            const size_t got = f->walk_plain ? 0 : walk_fast(b->pin(), idx.data(), idx.size(), &at, upto, b->metas() + n, cap, b->chunks[chunk].s);
            n += got;
            if (got == cap) continue;
            /* the record the fast lane stopped in front of (it crosses pieces, or something is wrong with it), by the record loop */
            size_t consumed = 0;
            const int64_t r = sq_split_range_indexed(b->pin(), at, upto, b->metas() + n, f->walk_plain ? cap : 1, &consumed, b->chunks[chunk].s, (size_t)-1,
                                                     nullptr, idx.data(), idx.size());
            if (r < 0) { stop = true; break; }      /* a malformed record: the caller's thread finds it again, with its window */
            n += (size_t)r;
            at += consumed;
            if (r == 0 || (f->walk_plain && (size_t)r < cap)) break;   /* what follows is not complete yet */
        }
        b->walk_next = at;
        b->walk_n.store(n, std::memory_order_release);
        if (stop) b->walk_stopped.store(true, std::memory_order_release);
        b->walk_scanned.store(upto, std::memory_order_release);
        g_feed_ns[2] += (uint64_t)(1e9 * (feed_now() - t_walk));
        lk.lock();
        f->busy--;
        f->cv_data.notify_all();
    }
}

/* the stats of the block's first n records out of the walker's chunks (the walker is not running) */
void walk_stats(FeedBlock *b, size_t n)
{
    uint64_t *S = b->stats;
    auto fold = [&](const uint64_t *c) {
        S[0] += c[0];
        for (int i = 1; i <= 4; i++) S[i] = std::max(S[i], c[i]);
        for (int i = 5; i < SQ_STATS_N; i++) S[i] += c[i];
    };
    size_t k = 0;
    for (; (k + 1) * WALK_CHUNK <= n && k < b->chunks.size(); k++) fold(b->chunks[k].s);
    for (size_t i = k * WALK_CHUNK; i < n; i++) {
        const sq_meta &m = b->metas()[i];
        const uint64_t L = m.sequence_length, span = (uint64_t)m.qualities_offset + L;
        S[0] += L;
        if (L > S[1]) S[1] = L;
        if (m.name_length > S[2]) S[2] = m.name_length;
        if (span > S[3]) S[3] = span;
        if (~L > S[4]) S[4] = ~L;
        S[5 + (L < SQ_LEN_BINS - 1 ? L : SQ_LEN_BINS - 1)]++;
    }
    b->walked = false;
}

/* from here on the caller's thread splits the open block's windows itself (the walker stays away from the block) */
void walk_leave(sq_feeder *f, FeedBlock *b)
{
    if (b->walk_off) return;
    {
        std::unique_lock<std::mutex> lk(f->mu);
        const bool was = f->paused;
        feed_pause(f, lk);
        b->walk_off = true;
        if (!was) feed_resume(f);
    }
    if (b->walked) walk_stats(b, b->n_records);
}

/* the block's place in HBM (the caller's thread; the workers and the walker see it under the feeder's lock) */
void early_place(sq_feeder *f, FeedBlock *b)
{
    if (b->d_text || b->early_bad) return;
    b->d_text = (uint8_t *)sq_dev_get(f->ctx, b->cap + 64);
    b->d_metas = b->d_text ? (sq_meta *)sq_dev_get(f->ctx, b->meta_cap * sizeof(sq_meta)) : nullptr;   /* neither there: the block goes up when it is sealed */
}

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
    if (f->copy_stream) {
        b->ctx = f->ctx;
        b->early_stream = f->copy_stream;
        if (f->want_early) early_place(f, b);
    }
    f->blocks.push_back(b);
    return b;
}

/* the block's early copies are given up (they have ended: nothing writes the device blocks that go back) */
void drop_early(FeedBlock *b, bool metas_only)
{
    if (!b->d_text && !b->d_metas) return;
    (void)hipStreamSynchronize(b->early_stream);
    if (b->d_metas) sq_dev_put(b->ctx, b->d_metas);
    b->d_metas = nullptr;
    b->metas_sent = 0;
    if (metas_only) return;
    if (b->d_text) sq_dev_put(b->ctx, b->d_text);
    b->d_text = nullptr;
}

void free_block(FeedBlock *b)
{
    drop_early(b, false);
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
    std::unique_lock<std::mutex> lk(f->mu, std::defer_lock);
    if (f->has_source) {   /* the workers leave the block alone while it is closed and its tail moves */
        lk.lock();
        feed_pause(f, lk);
    }

    struct Resume { sq_feeder *f; ~Resume() { if (f->has_source) feed_resume(f); } } resume{f};
    FeedBlock *o = open_block(f);
    const size_t carry = o ? o->used - f->pos : 0;
    if (o) {
        o->sealed = true;
        o->sealed_bytes = f->pos;
        if (o->walked) walk_stats(o, o->n_records);
    }
    FeedBlock *n = new_block(f, carry + room);
    if (!n) { sq_set_error("out of memory for a staging block"); return SQ_ERR_MEMORY; }
    if (carry) memcpy(n->pin(), o->pin() + f->pos, carry);
    n->used = carry;
    if (carry && n->d_text && !f->walker_sends && hipMemcpyAsync(n->d_text, n->pin(), carry, hipMemcpyHostToDevice, n->early_stream) != hipSuccess) {
        (void)hipGetLastError();
        n->early_bad = true;
    }
    if (f->has_source) {
        n->reserved = carry;
        if (carry) {   /* what the workers noted of the carried bytes comes along, as one piece */
            FeedBlock::Piece *pc = new FeedBlock::Piece();
            pc->from = 0;
            pc->to = carry;
            pc->done = true;
            bool look_again = false;
            for (const FeedBlock::Piece *q : o->pieces) {
                if (q->to <= f->pos) continue;
                for (size_t i = 0; i < q->nl.size(); i++)
                    if (q->nl[i] >= f->pos) { pc->nl.push_back(q->nl[i] - (uint32_t)f->pos); pc->after.push_back(q->after[i]); }
                if (q->first_high != UINT32_MAX && pc->first_high == UINT32_MAX) {
                    if (q->first_high >= f->pos) pc->first_high = q->first_high - (uint32_t)f->pos;
                    else look_again = true;   /* the piece's first one stays behind: is there another in what moves? */
                }
            }
            if (look_again && pc->first_high == UINT32_MAX) {
                const int64_t r = sq_first_non_ascii_fast(n->pin(), carry);
                if (r >= 0) pc->first_high = (uint32_t)r;
            } else if (look_again) {
                const int64_t r = sq_first_non_ascii_fast(n->pin(), pc->first_high);
                if (r >= 0) pc->first_high = (uint32_t)r;
            }
            n->high_min.store(pc->first_high, std::memory_order_relaxed);
            n->pieces.push_back(pc);
        }
        if (!f->walk_on) n->walk_off = true;
    } else
        n->walk_off = true;
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
    f->block_bytes = block_bytes ? block_bytes : ((size_t)128 << 20);
    return f;
}

SQ_EXPORT void sq_feeder_free(sq_feeder *f)
{
    if (!f) return;
    if (!f->workers.empty()) {
        { std::lock_guard<std::mutex> g(f->mu); f->stop = true; }
        f->cv_work.notify_all();
        f->cv_data.notify_all();
        for (std::thread &t : f->workers) t.join();
        if (f->walker.joinable()) f->walker.join();
    }
    for (FeedBlock *b : f->blocks) free_block(b);
    if (f->copy_done) (void)hipEventDestroy(f->copy_done);
    delete f;
}

/* The feeder reads its text by itself from here on: `len` bytes at `text` (memory the caller keeps valid and unchanged
 * until sq_feeder_free -- a BytesIO's buffer, a mapping), or the bytes [offset, offset + len) of the regular file `fd`
 * (pread: the descriptor's own position is neither used nor moved).  To be called before the first sq_feeder_next;
 * sq_feeder_next then never answers SQ_FEED_MORE.  The arrays are those of the caller-fed feeder: the same windows of the
 * same bytes. */
static int feeder_start(sq_feeder *f)
{
    if (f->has_source || !f->blocks.empty()) { sq_set_error("sq_feeder_set_source: the parser has started already"); return SQ_ERR_VALUE; }
    if (f->read_in >= ((size_t)1 << 30)) { sq_set_error("sq_feeder_set_source: buffers of 1 GiB and more are read by the caller"); return SQ_ERR_VALUE; }
    f->has_source = true;
    /* with a device: the blocks go up piece by piece as they fill (SQ_FEED_EARLY=0: when they are sealed, as for a
       caller-fed feeder) */
    const char *early = getenv("SQ_FEED_EARLY");
    if (f->ctx && !(early && (early[0] == '0' || early[0] == 0))) {
        /* one stream for the feeders of a context (making one takes a millisecond or two: per parser that showed) */
        if (!f->ctx->feed_stream && hipStreamCreateWithFlags(&f->ctx->feed_stream, hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError();
            f->ctx->feed_stream = nullptr;
        }
        if (f->ctx->feed_stream && hipEventCreateWithFlags(&f->copy_done, hipEventDisableTiming) == hipSuccess)
            f->copy_stream = f->ctx->feed_stream;
        else
            (void)hipGetLastError();
    }
    if (const char *v = getenv("SQ_FEED_COPY")) f->plain_copy = !strcmp(v, "plain");
    const unsigned hc = std::thread::hardware_concurrency();
    unsigned n = std::max(1u, std::min(4u, hc > 1 ? hc - 1 : 1u));
    if (f->src_end - f->src_off < ((uint64_t)4 << 20)) n = 1;   /* a few pieces: one worker (a parser over a few records should not start four threads) */
    if (const char *v = getenv("SQ_FEED_WORKERS")) n = (unsigned)std::max(1, std::min(16, atoi(v)));   /* (experiments) */
    for (unsigned i = 0; i < n; i++) f->workers.emplace_back(feed_worker, f);
    /* and one more thread splits the records as the text arrives (SQ_FEED_WALKER=0: sq_feeder_next does, window by window) */
    const char *w = getenv("SQ_FEED_WALKER");
    f->walk_plain = w && !strcmp(w, "plain");
    if (!(w && (w[0] == '0' || w[0] == 0)) && hc > 2) {
        f->walk_on = true;
        f->walker_sends = f->copy_stream != nullptr;
        f->walker = std::thread(feed_walker, f);
    }
    return SQ_OK;
}
SQ_EXPORT int sq_feeder_set_source_memory(sq_feeder *f, const uint8_t *text, size_t len)
{
    f->src_mem = text; f->src_fd = -1; f->src_off = 0; f->src_end = len;
    return feeder_start(f);
}
SQ_EXPORT int sq_feeder_set_source_fd(sq_feeder *f, int fd, uint64_t offset, uint64_t len)
{
    f->src_mem = nullptr; f->src_fd = fd; f->src_off = offset; f->src_end = offset + len;
    return feeder_start(f);
}

/* Where the next bytes of the file go and how many of them fit (*room >= 1). */
SQ_EXPORT uint8_t *sq_feeder_fill(sq_feeder *f, size_t *room)
{
    if (f->has_source) { sq_set_error("sq_feeder_fill: the feeder reads its source by itself"); *room = 0; return nullptr; }
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
    if (f->has_source) { sq_set_error("sq_feeder_filled: the feeder reads its source by itself"); return SQ_ERR_VALUE; }
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
    if (f->walk_on && (min_records > 1 || max_records < ((size_t)1 << 62))) {   /* (__next__ asks for 1 .. 2^63 - 1 or SIZE_MAX records: all a window holds) */   /* FastqParser.read(n): the caller's thread splits from here on */
        { std::lock_guard<std::mutex> g(f->mu); f->walk_on = false; }
        if (b) walk_leave(f, b);
    }
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
            size_t used_now = f->has_source ? 0 : b->used;   /* (with a source: `used` is the workers', read under the lock below) */
            if (f->has_source && !b->walk_off && b->walk_scanned.load(std::memory_order_acquire) >= f->pos + f->arr_len + want) {
                used_now = b->walk_scanned.load(std::memory_order_acquire);   /* the walker has been through this read's bytes: they are there */
            } else if (f->has_source) {   /* the workers bring the bytes: wait until this read's are there, the source is exhausted or the block is full */
                std::unique_lock<std::mutex> lk(f->mu);
                const size_t upto = f->pos + f->arr_len + want;
                for (;;) {
                    used_now = b->used;
                    if (f->src_failed) {
                        sq_set_error("the file could not be read to its end");
                        f->in_array = false;
                        return SQ_ERR_HIP;
                    }
                    const bool drained = f->src_off >= f->src_end && f->busy == 0 && b->used == b->reserved;
                    if (drained) f->file_eof = true;
                    if (used_now >= upto || drained || (b->reserved >= b->cap && f->busy == 0 && b->used == b->reserved)) break;
                    f->cv_work.notify_all();
                    const double t_wait = feed_now();
                    f->cv_data.wait(lk);
                    g_feed_ns[0] += (uint64_t)(1e9 * (feed_now() - t_wait));
                }
            }
            const size_t have = used_now - (f->pos + f->arr_len);
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
                if (f->has_source) continue;   /* the new block fills by itself: this read again */
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
        int64_t n = 0;
        uint64_t stats[SQ_STATS_N];
        bool from_walker = false;
        if (!b->walk_off) {   /* the walker's records that end inside the window */
            const size_t wend = f->pos + f->arr_len;
            auto there = [&]() { return b->walk_stopped.load(std::memory_order_acquire) || b->walk_scanned.load(std::memory_order_acquire) >= wend; };
            if (!there()) {
                std::unique_lock<std::mutex> lk(f->mu);
                const double t_wait = feed_now();
                while (!there()) f->cv_data.wait(lk);
                g_feed_ns[1] += (uint64_t)(1e9 * (feed_now() - t_wait));
            }
            const bool stopped = b->walk_stopped.load(std::memory_order_acquire);
            const size_t wn = b->walk_n.load(std::memory_order_acquire), lo = b->n_records;
            const sq_meta *M = b->metas();
            auto ends_inside = [&](size_t k) { return M[k].record_start + M[k].tags_offset < wend; };   /* its fourth newline */
            size_t a = lo, z = wn, step = 512;   /* the records in front of a end inside, record z does not (or is not there) */
            while (a < z) {
                const size_t probe = std::min(z - 1, a + step - 1);
                if (ends_inside(probe)) { a = probe + 1; step *= 2; }
                else { z = probe; break; }
            }
            while (a < z) {
                const size_t mid = a + (z - a) / 2;
                if (ends_inside(mid)) a = mid + 1; else z = mid;
            }
            /* what the record loop over this window would do that the walker's records do not tell: raise for a byte >= 0x80
               among the new bytes; look at the record the walker gave up on.  Then it runs (and the block is its from here on) */
            bool leave = fresh_from != (size_t)-1 && b->high_min.load(std::memory_order_relaxed) < wend;
            if (stopped && a == wn && b->walk_next + 2 < wend) leave = true;
            if (leave)
                walk_leave(f, b);
            else {
                n = (int64_t)(a - lo);
                consumed = n ? (size_t)(M[a - 1].record_start + M[a - 1].tags_offset + 1) - f->pos : 0;
                from_walker = true;
            }
        }
        while (!from_walker) {
            const size_t cap = std::min(b->meta_cap - b->n_records, max_records);
            memcpy(stats, b->stats, sizeof stats);
            int64_t bad = -1;
            const double t_split = feed_now();
            if (f->has_source) {
                std::vector<SqNlPiece> idx;
                {
                    std::lock_guard<std::mutex> g(f->mu);
                    for (const FeedBlock::Piece *q : b->pieces)
                        if (q->to > f->pos && q->from < f->pos + f->arr_len) idx.push_back(SqNlPiece{q->from, q->to, q->nl.data(), q->after.data(), q->nl.size(), q->first_high});
                }
                n = sq_split_range_indexed(b->pin(), f->pos, f->pos + f->arr_len, b->metas() + b->n_records, cap, &consumed, stats, fresh_from, &bad,
                                           idx.data(), idx.size());
            } else
            n = sq_split_range_ascii(b->pin(), f->pos, f->pos + f->arr_len, b->metas() + b->n_records, cap, &consumed, stats, fresh_from, &bad);
            g_feed_times[1] += feed_now() - t_split;
            if (non_ascii_error(bad)) return SQ_ERR_VALUE;
            if (n < 0) { f->in_array = false; return (int)n; }
            if ((size_t)n < cap || (size_t)n == max_records) break;
            /* the meta area is full: a bigger one (the block keeps its text) */
            PinBuf bigger = pool_get(2 * b->meta_cap * sizeof(sq_meta), f->ctx != nullptr);
            if (!bigger.p) { sq_set_error("out of memory for the record table"); f->in_array = false; return SQ_ERR_MEMORY; }
            memcpy(bigger.p, b->meta.p, b->n_records * sizeof(sq_meta));
            drop_early(b, true);   /* copies may still read the old table, and the one in HBM is as small */
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
            if (from_walker) b->walked = true; else memcpy(b->stats, stats, sizeof stats);
            out->block_id = b->id;
            out->byte_start = f->pos;
            out->byte_len = f->arr_len;
            out->first_record = b->n_records;
            out->n_records = (uint64_t)n;
            b->n_records += (size_t)n;
            f->logical_end = f->pos + f->arr_len;
            f->pos += consumed;
            f->in_array = false;
            if (b->d_metas && b->n_records - b->metas_sent >= FEED_EARLY_METAS) {   /* the metas of arrays handed out do not change any more */
                if (hipMemcpyAsync(b->d_metas + b->metas_sent, b->metas() + b->metas_sent, (b->n_records - b->metas_sent) * sizeof(sq_meta),
                                   hipMemcpyHostToDevice, b->early_stream) != hipSuccess) {
                    (void)hipGetLastError();
                    b->early_bad = true;
                }
                b->metas_sent = b->n_records;
            }
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

/* The caller will ask for the parser's blocks in HBM (sq_feeder_upload): from the open block on they go up while they fill.
 * Without this call the first upload says so; a parser that is only iterated uploads nothing. */
SQ_EXPORT int sq_feeder_expect_uploads(sq_feeder *f)
{
    if (f->want_early || !f->copy_stream) return SQ_OK;
    std::unique_lock<std::mutex> lk(f->mu);
    f->want_early = true;
    FeedBlock *b = open_block(f);
    if (!b) return SQ_OK;
    feed_pause(f, lk);
    early_place(f, b);
    if (b->d_text && !f->walker_sends && b->used &&
        hipMemcpyAsync(b->d_text, b->pin(), b->used, hipMemcpyHostToDevice, b->early_stream) != hipSuccess) {   /* what the workers brought before */
        (void)hipGetLastError();
        b->early_bad = true;
    }
    feed_resume(f);
    return SQ_OK;
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
    if (!f->want_early) (void)sq_feeder_expect_uploads(f);
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
    /* text and metas from the context's pool (as sq_batch_from_fastq's): a hipMalloc / hipFree pair per block and array was 0.4 ms
       of host time per block, and hipFree waits for the device */
    b->pooled = true;
    const bool early = fb->d_text && !fb->early_bad;
    size_t metas_there = 0;
    if (early) {   /* the text is in HBM but for what is still on its way; so are the metas of all but the last arrays */
        b->d_buf = fb->d_text;
        fb->d_text = nullptr;
        if (fb->d_metas) {
            b->d_metas = fb->d_metas;
            fb->d_metas = nullptr;
            metas_there = fb->metas_sent;
        } else
            b->d_metas = (sq_meta *)sq_dev_get(ctx, (b->n ? b->n : 1) * sizeof(sq_meta));
    } else {
        drop_early(fb, false);
        b->d_buf = (uint8_t *)sq_dev_get(ctx, b->buf_len + 64);
        b->d_metas = b->d_buf ? (sq_meta *)sq_dev_get(ctx, (b->n ? b->n : 1) * sizeof(sq_meta)) : nullptr;
    }
    if (!b->d_buf || !b->d_metas) {
        sq_set_error("sq_feeder_upload: out of device memory");
        if (early) (void)hipStreamSynchronize(fb->early_stream);
        if (b->d_buf) sq_dev_put(ctx, b->d_buf);
        delete b;
        return nullptr;
    }
    auto fail = [&](const char *what, hipError_t e) -> sq_batch * {
        sq_set_error("sq_feeder_upload: %s: %s", what, hipGetErrorString(e));
        if (early) (void)hipStreamSynchronize(fb->early_stream);
        (void)hipStreamSynchronize(ctx->stream);   /* nothing may still be writing the blocks that go back */
        if (b->ready) (void)hipEventDestroy(b->ready);
        sq_dev_put(ctx, b->d_buf);
        sq_dev_put(ctx, b->d_metas);
        delete b;
        return nullptr;
    };
    hipError_t e = hipSuccess;
    if (early) {
        const size_t sent = f->walker_sends ? std::min(fb->sent_to, b->buf_len) : b->buf_len;   /* (the walker is at the next block) */
        if (sent < b->buf_len && (e = hipMemcpyAsync(b->d_buf + sent, fb->pin() + sent, b->buf_len - sent, hipMemcpyHostToDevice, fb->early_stream)) != hipSuccess)
            return fail("text upload", e);
        if (b->n > metas_there && (e = hipMemcpyAsync(b->d_metas + metas_there, fb->metas() + metas_there, (b->n - metas_there) * sizeof(sq_meta),
                                                      hipMemcpyHostToDevice, fb->early_stream)) != hipSuccess) return fail("meta upload", e);
        if ((e = hipEventRecord(f->copy_done, fb->early_stream)) != hipSuccess) return fail("event", e);
        if ((e = hipStreamWaitEvent(ctx->stream, f->copy_done, 0)) != hipSuccess) return fail("event", e);
    } else {
        if (b->buf_len && (e = hipMemcpyAsync(b->d_buf, fb->pin(), b->buf_len, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) return fail("text upload", e);
        if (b->n && (e = hipMemcpyAsync(b->d_metas, fb->metas(), b->n * sizeof(sq_meta), hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) return fail("meta upload", e);
    }
    if ((e = hipMemsetAsync(b->d_buf + b->buf_len, 0, 64, ctx->stream)) != hipSuccess) return fail("padding", e);
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

SQ_EXPORT void sq_feeder_debug_waits(double *out, int reset)
{
    for (int i = 0; i < 4; i++) { out[i] = 1e-9 * (double)g_feed_ns[i].load(); if (reset) g_feed_ns[i] = 0; }
}

SQ_EXPORT void sq_feeder_release(sq_feeder *f, uint64_t block_id)
{
    std::unique_lock<std::mutex> lk(f->mu, std::defer_lock);
    if (f->has_source) lk.lock();   /* the workers look at the list of blocks */
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
