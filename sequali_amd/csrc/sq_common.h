/* sq_common.h -- internals shared by the translation units of libsqgpu.so */
#ifndef SQ_COMMON_H
#define SQ_COMMON_H

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/sqgpu.h"

#define SQ_EXPORT extern "C" __attribute__((visibility("default")))

/* thread-local message of the last failing call */
void sq_set_error(const char *fmt, ...);

#define SQ_HIP(call)                                                                    \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) {                                                         \
            sq_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                         __LINE__);                                                     \
            return SQ_ERR_HIP;                                                          \
        }                                                                               \
    } while (0)

#define SQ_HIP_NULL(call)                                                               \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) {                                                         \
            sq_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                         __LINE__);                                                     \
            return nullptr;                                                             \
        }                                                                               \
    } while (0)

/* Route selection switches, read from the environment ONCE (first use) or when sq_knobs_reload() is called (the tests
 * flip a switch, reload, run, flip it back): the dispatchers consult this struct, never getenv().  Defaults = what
 * production runs.  Every switch names a kernel that is the DEFAULT for some shape of batch (tests/test_gpu_routes.py)
 * and exists so that the tests can send batches of any size through it (round 6 took out the experiment switches):
 *   SQ_SPAN=0           no k_span / k_ptspan / k_isz_span: the kernels behind them (k_wide, k_ring, k_pass, k_ptq, k_seg)
 *   SQ_SPAN_SPLIT=0     k_span with one wave for both streams of a span
 *   SQ_SPAN_SPLIT_QC    0 / 1: QCMetrics alone never / always with a wave per stream (default: from 161 bases on)
 *   SQ_SPAN_SHORT=1     k_span also for batches of one read length of up to 64 bases with adapters (default: k_wide)
 *   SQ_SPAN_SORTED      1 / 0: force / forbid the length-sorted k_span route for ragged batches
 *   SQ_SPAN_W6          0 / 1: adapters of 14 .. 25 characters never / wherever a build exists through k_span
 *   SQ_NO_WIDE, SQ_RING, SQ_NO_RING   k_pass instead of k_wide / k_ring; k_ring also with the automaton
 *   SQ_NO_PTQ           PerTileQuality alone through k_pass
 *   SQ_PT_FUSED         0: PerTileQuality and InsertSizeMetrics in passes of their own; 2: only the tile ids from the pass
 *   SQ_LONG=0, SQ_NO_SEGMENTS   long reads: k_seg instead of k_span<LONG>; stripes of k_pass instead of segments
 *   SQ_DEDUP_SEQUENTIAL DedupEstimator: every piece through the host's sequential loop
 *   SQ_OVERREP_CHAIN    OverrepresentedSequences: k_overrep instead of k_overrep_par */
struct SqKnobs {
    bool span = true, span_split = true;
    int span_sorted = -1;
    int span_w6 = -1;          /* SQ_SPAN_W6: adapters of 14 .. 25 characters through k_span (sq_span_w6.hip) instead of k_wide / k_pass.  -1 (default, measured in round 5, scripts/exp_w6.sh): batches of one read length from 129 bases on (914 / 910 / 1029 against k_wide's 876 / 830 / 901 Gbases/s at 150 / 200 / 224 bases; at 100 bases k_wide's 776 against 694) and every length-sorted batch (the alternative there is k_pass); 0: never; 1: wherever a build exists */
    bool span_short = false;   /* SQ_SPAN_SHORT: k_span also for batches of one read length of up to 64 bases with adapters (default: k_wide, 22 % ahead at 50 bases) */
    int span_split_qc = -1;    /* SQ_SPAN_SPLIT_QC: QCMetrics alone with a wave per stream.  -1 (default, measured in round 5, profiles/r5/exp_split_qc.txt): from 6 windows (161 bases) on, where one wave for both streams holds 8 waves a CU (1218 / 1290 against 1066 / 1145 Gbases/s at 200 / 250 bases; at 100 / 150 bases one wave for both is 2-4 % ahead); 0: never; 1: always */
    bool no_wide = false, ring = false, no_ring = false;
    bool no_ptq = false, no_segments = false;
    int pt_fused = 1;          /* SQ_PT_FUSED: 1 (default since round 5: tests/test_gpu_pair.py is green on a GPU): PerTileQuality rides in QCMetrics' pass on batches of one read length (k_span<PT>, sq_pair.hip); 0: the passes of round 2 (k_tile_parse, k_span, k_ptspan); 2: tile ids from the pass, the table by k_ptspan */
    bool long_spans = true;
    bool dedup_sequential = false;
    bool overrep_chain = false;   /* SQ_OVERREP_CHAIN: k_overrep (a lane's fragments one after the other: reads of more than 10 fragments, fragments of more than 24 bases) for every batch */
};
const SqKnobs &sq_knobs();

/* repr() of an ASCII str, as CPython writes it (unicode_repr): what PyErr_Format's %R puts into the reference's messages
 * (_qcmodule.c:1141-1146: the record's name).  Single quotes unless the text holds one and no double quote; backslash,
 * the quote in use, tab, newline and carriage return escaped; other control characters and DEL as \xNN. */
inline std::string sq_py_repr_ascii(const char *p, size_t n)
{
    bool single = false, dbl = false;
    for (size_t i = 0; i < n; i++) { single |= p[i] == '\''; dbl |= p[i] == '"'; }
    const char quote = single && !dbl ? '"' : '\'';
    std::string out(1, quote);
    static const char hex[] = "0123456789abcdef";
    for (size_t i = 0; i < n; i++) {
        const unsigned char ch = (unsigned char)p[i];
        if (ch == (unsigned char)quote || ch == '\\') { out += '\\'; out += (char)ch; }
        else if (ch == '\t') out += "\\t";
        else if (ch == '\n') out += "\\n";
        else if (ch == '\r') out += "\\r";
        else if (ch < 0x20 || ch == 0x7F) { out += "\\x"; out += hex[ch >> 4]; out += hex[ch & 15]; }
        else out += (char)ch;
    }
    out += quote;
    return out;
}

/* What a worker of the parser's feeder has learnt about a stretch [from, to) of a staging block while it copied it in
 * (sq_feed.hip): the offsets (in the block) of its newlines, ascending, the byte that follows each of them (0: not noted
 * -- it lies behind the stretch --, or it is a zero byte: look at the text), and the offset of its first byte >= 0x80
 * (UINT32_MAX: none).  The record split takes its newlines from these instead of scanning the text again
 * (sq_split_range_indexed), and the two characters it checks per record ('@', '+': each the byte behind a newline) too: the
 * text was written by other cores and every look at it is a cache miss. */
struct SqNlPiece {
    size_t from, to;
    const uint32_t *nl;
    const uint8_t *after;
    size_t n_nl;
    uint32_t first_high;
};

struct sq_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;   /* uploads of FASTQ text (sq_batch_from_fastq): they run beside the counting of the batch before */
    hipEvent_t copied = nullptr;
    hipStream_t feed_stream = nullptr;   /* the feeders' early uploads of their staging blocks (sq_feed.hip; made when the first one asks) */
    hipStream_t prep_stream = nullptr;   /* PerTileQuality's pass over the headers (tile ids, table slots): it runs beside the counting of the batch before */
    int num_cus = 256;
    /* small pinned scratch for scalar read-backs */
    uint64_t *pinned = nullptr; /* 64 words */
    std::string route;   /* the counting kernels launched since sq_route_reset(): "k_span<5,AD,split>+k_ptspan<5>+..." (sq_last_route) */
    uint64_t *pinned_stats = nullptr; /* SQ_STATS_N words: k_batch_stats' read-back */
    /* grow-only device scratch buffers (sorting), reused across batches so that no
       hipFree (a device-wide sync) sits between launches */
    void *scratch[40] = {};       /* 0-5: the fused pass (sorting, carries; 3 holds P.order while a pass runs); 6-13: selections and DedupEstimator; 14-15: k_span over sorted reads; 16: k_isz_span; 18-20: the device-side FASTQ split; 21: k_batch_stats; 22: workgroup shares of k_span<LONG>; 23: its reads per segment; 24-28: the paired pass (sq_pair.hip); 29-31: the DedupEstimator's lower bound (sq_dedup_shard_settle); 32-39: the DedupEstimator's table steps (dedup_process) */
    size_t scratch_bytes[40] = {};
    /* small host arrays an asynchronous copy reads from (segment tables, row starts): the last few calls' copies
       stay alive here, whatever HIP does with pageable sources */
    std::vector<uint8_t> host_keep[8];
    unsigned host_keep_at = 0;
    /* device blocks of batches made by sq_batch_from_fastq (text, metas): kept for the next buffer of the
       same size instead of a hipFree + hipMalloc per buffer (hipFree waits for the whole device) */
    struct DevBlock { void *p; size_t cap; };
    std::vector<DevBlock> pool_free, pool_live;
    size_t pool_free_bytes = 0;
    /* blocks of batches that were freed while kernels on `stream` may still read them: they go back to the pool once the
       event has passed (sq_batch_free used to wait for the stream: 0.25 ms of the caller's thread per staging block) */
    struct Deferred { void *a, *b; hipEvent_t passed; };
    std::vector<Deferred> deferred;
    uint64_t pool_mallocs = 0, pool_frees = 0;   /* (sq_pool_counts) */
    /* FASTQ text uploaded ahead of the call that will split it (sq_batch_from_fastq_ahead) */
    struct Ahead { const uint8_t *host = nullptr; size_t len = 0; uint8_t *dev = nullptr; };
    Ahead ahead;
};

inline void sq_dev_put(sq_ctx *ctx, void *p);
/* the deferred blocks whose event has passed go back to the pool; wait: all of them, waited for */
inline void sq_dev_reclaim(sq_ctx *ctx, bool wait)
{
    size_t kept = 0;
    for (size_t i = 0; i < ctx->deferred.size(); i++) {
        sq_ctx::Deferred d = ctx->deferred[i];
        if (wait) (void)hipEventSynchronize(d.passed);
        else if (hipEventQuery(d.passed) != hipSuccess) {
            (void)hipGetLastError();   /* hipErrorNotReady */
            ctx->deferred[kept++] = d;
            continue;
        }
        (void)hipEventDestroy(d.passed);
        sq_dev_put(ctx, d.a);
        sq_dev_put(ctx, d.b);
    }
    ctx->deferred.resize(kept);
}

/* a device block of at least `bytes` from the context's pool (sq_dev_put hands it back) */
inline void *sq_dev_get(sq_ctx *ctx, size_t bytes)
{
    if (!ctx->deferred.empty()) {
        sq_dev_reclaim(ctx, false);
        bool fits = false;
        for (const sq_ctx::DevBlock &f : ctx->pool_free) fits = fits || (f.cap >= bytes && f.cap <= 2 * bytes + (1 << 20));
        if (!fits && !ctx->deferred.empty()) sq_dev_reclaim(ctx, true);   /* rather than a hipMalloc */
    }
    size_t best = ctx->pool_free.size();
    for (size_t i = 0; i < ctx->pool_free.size(); i++)
        if (ctx->pool_free[i].cap >= bytes && ctx->pool_free[i].cap <= 2 * bytes + (1 << 20) &&
            (best == ctx->pool_free.size() || ctx->pool_free[i].cap < ctx->pool_free[best].cap))
            best = i;
    sq_ctx::DevBlock blk{nullptr, 0};
    if (best < ctx->pool_free.size()) {
        blk = ctx->pool_free[best];
        ctx->pool_free.erase(ctx->pool_free.begin() + (long)best);
        ctx->pool_free_bytes -= blk.cap;
    } else {
        blk.cap = (bytes + (bytes >> 4) + 0xFFFFF) & ~(size_t)0xFFFFF;   /* buffers of one parser differ by a leftover */
        ctx->pool_mallocs++;
        if (hipMalloc(&blk.p, blk.cap) != hipSuccess) {
            /* give the pool's idle blocks back and try once more */
            for (auto &f : ctx->pool_free) (void)hipFree(f.p);
            ctx->pool_free.clear();
            ctx->pool_free_bytes = 0;
            if (hipMalloc(&blk.p, blk.cap) != hipSuccess) return nullptr;
        }
    }
    ctx->pool_live.push_back(blk);
    return blk.p;
}

inline void sq_dev_put(sq_ctx *ctx, void *p)
{
    if (!p) return;
    for (size_t i = 0; i < ctx->pool_live.size(); i++)
        if (ctx->pool_live[i].p == p) {
            const sq_ctx::DevBlock blk = ctx->pool_live[i];
            ctx->pool_live.erase(ctx->pool_live.begin() + (long)i);
            /* the idle blocks that have lain longest make room: a pool full of sizes nobody asks for any more (another
               parser's first small block, metas of arrays of another size) made every put a hipFree and every get a
               hipMalloc -- 18 + 10 of them per pass over 11 staging blocks (profiles/r6/exp_e2e_pool.txt) */
            if (blk.cap > ((size_t)2 << 30)) {
                ctx->pool_frees++;
                (void)hipFree(blk.p);
                return;
            }
            while (!ctx->pool_free.empty() && (ctx->pool_free.size() >= 16 || ctx->pool_free_bytes + blk.cap > ((size_t)4 << 30))) {
                ctx->pool_frees++;
                ctx->pool_free_bytes -= ctx->pool_free.front().cap;
                (void)hipFree(ctx->pool_free.front().p);
                ctx->pool_free.erase(ctx->pool_free.begin());
            }
            ctx->pool_free.push_back(blk);
            ctx->pool_free_bytes += blk.cap;
            return;
        }
    (void)hipFree(p);   /* not from the pool */
}

/* a copy of `bytes` bytes at `src` that outlives the caller's frame (see sq_ctx::host_keep) */
inline const void *sq_host_keep(sq_ctx *ctx, const void *src, size_t bytes)
{
    std::vector<uint8_t> &v = ctx->host_keep[ctx->host_keep_at++ % 8];
    v.assign((const uint8_t *)src, (const uint8_t *)src + bytes);
    return v.data();
}

inline void *sq_scratch(sq_ctx *ctx, int i, size_t bytes)
{
    if (ctx->scratch_bytes[i] < bytes) {
        if (ctx->scratch[i]) {
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipFree(ctx->scratch[i]);
        }
        size_t want = bytes + bytes / 4;
        if (hipMalloc(&ctx->scratch[i], want) != hipSuccess) {
            ctx->scratch[i] = nullptr;
            ctx->scratch_bytes[i] = 0;
            return nullptr;
        }
        ctx->scratch_bytes[i] = want;
    }
    return ctx->scratch[i];
}

/* what is known of a batch's records without reading them again: bases, longest read, longest name, longest
 * record span, ~(shortest read), then how many reads have 0, 1, ..., 255 and 256 or more bases (k_span over a
 * batch of many lengths puts its rows in order with these counts instead of sorting keys) */
constexpr int SQ_LEN_BINS = 257;
constexpr int SQ_STATS_N = 5 + SQ_LEN_BINS;

/* names the kernel a dispatcher has just chosen (tests assert the routes: a build that spills makes a dispatcher
 * fall back to another kernel without a word) */
inline void sq_route(sq_ctx *ctx, const char *fmt, ...)
{
    if (ctx->route.size() > 2000) return;
    char buf[96];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (!ctx->route.empty()) ctx->route += '+';
    ctx->route += buf;
}

inline uint64_t sq_next_batch_id() { static uint64_t next = 0; return ++next; }   /* objects are used from one thread (sqgpu.h) */

struct sq_batch {
    sq_ctx *ctx = nullptr;
    uint64_t id = sq_next_batch_id();   /* identity that survives the address being reused by a later batch */
    uint8_t *d_buf = nullptr;
    sq_meta *d_metas = nullptr;
    size_t buf_len = 0;
    size_t n = 0;
    hipEvent_t ready = nullptr;   /* recorded behind an upload that was still running when the batch was handed out (a kernel on another stream waits for it) */
    bool pooled = false;     /* d_buf and d_metas are blocks of the context's pool (sq_dev_get) */
    bool owns = false;       /* frees d_buf (and d_metas) */
    bool owns_metas = false; /* frees d_metas although d_buf is borrowed */
    bool slack = false;      /* 64 readable bytes follow d_buf + buf_len (the library allocated the text): the kernels whose
                                wide loads run past the last record may take the batch.  True for every batch that owns its
                                text and for a view of such a batch (sq_batch_view) */
    std::vector<uint32_t> len_hist;   /* [SQ_LEN_BINS] reads per length (the last bin: 256 and more); empty: not counted */
    uint64_t total_bases = 0;
    uint64_t max_length = 0;
    uint64_t min_length = 0;
    uint64_t max_name_length = 0;
    uint64_t max_record_span = 0;
    /* host copies kept by sq_batch_upload for rare host-side follow-ups
       (error messages, skipped_reason); empty for wrapped device memory */
    std::vector<uint8_t> h_buf;
    std::vector<sq_meta> h_metas;
};

/* grow a device array of T to at least `want` elements, zero-filling the new
 * tail and preserving the old contents; *cap is the current element count */
template <typename T>
int sq_grow_device(sq_ctx *ctx, T **ptr, size_t *cap, size_t want)
{
    if (want <= *cap) return SQ_OK;
    T *n = nullptr;
    SQ_HIP(hipMalloc((void **)&n, want * sizeof(T)));
    SQ_HIP(hipMemsetAsync(n, 0, want * sizeof(T), ctx->stream));
    if (*ptr && *cap)
        SQ_HIP(hipMemcpyAsync(n, *ptr, *cap * sizeof(T), hipMemcpyDeviceToDevice, ctx->stream));
    if (*ptr) {
        SQ_HIP(hipStreamSynchronize(ctx->stream));
        SQ_HIP(hipFree(*ptr));
    }
    *ptr = n;
    *cap = want;
    return SQ_OK;
}

/* Python's repr() of an ASCII str, as %R prints it in _qcmodule.c:3144 */
inline std::string sq_py_repr(const std::string &s)
{
    bool has_sq = s.find('\'') != std::string::npos, has_dq = s.find('"') != std::string::npos;
    char quote = (has_sq && !has_dq) ? '"' : '\'';
    std::string r(1, quote);
    for (unsigned char c : s) {
        if (c == (unsigned char)quote || c == '\\') { r += '\\'; r += (char)c; }
        else if (c == '\t') r += "\\t";
        else if (c == '\n') r += "\\n";
        else if (c == '\r') r += "\\r";
        else if (c < 0x20 || c == 0x7F) { char t[8]; snprintf(t, sizeof t, "\\x%02x", c); r += t; }
        else r += (char)c;
    }
    r += quote;
    return r;
}

/* device-side helpers -------------------------------------------------------- */
#ifdef __HIPCC__
/* NUCLEOTIDE_TO_INDEX, _qcmodule.c:1748-1763: A/a 0 C/c 1 G/g 2 T/t 3 else 4 */
__device__ __forceinline__ unsigned sq_base_class(unsigned c)
{
    unsigned l = c | 0x20u;
    return l == 'a' ? 0u : l == 'c' ? 1u : l == 'g' ? 2u : l == 't' ? 3u : 4u;
}

__device__ __forceinline__ uint32_t sq_load_u32_unaligned(const uint8_t *p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

__host__ __device__ __forceinline__ uint64_t sq_load_u64_unaligned(const uint8_t *p)
{
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
#endif

#endif
