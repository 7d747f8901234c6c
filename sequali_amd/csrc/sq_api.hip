/* sq_api.hip -- context, batches, host-side record boundary, synthetic FASTQ */
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include "sq_common.h"
#include "sq_synth_core.h"

static thread_local std::string g_last_error;

void sq_set_error(const char *fmt, ...)
{
    char tmp[1024];
    va_list ap, ap2;
    va_start(ap, fmt);
    va_copy(ap2, ap);
    const int need = vsnprintf(tmp, sizeof(tmp), fmt, ap);
    va_end(ap);
    if (need >= 0 && (size_t)need < sizeof(tmp)) {
        g_last_error.assign(tmp, (size_t)need);   /* by length: "%c" of a zero byte of the input is part of the message (sq_last_error_length) */
    } else if (need > 0) {   /* the reference puts whole buffers into its messages (:1076): no cut */
        g_last_error.resize((size_t)need + 1);
        vsnprintf(&g_last_error[0], (size_t)need + 1, fmt, ap2);
        g_last_error.resize((size_t)need);
    }
    va_end(ap2);
}

/* the switches of SqKnobs (sq_common.h) */
static SqKnobs g_knobs;
static bool g_knobs_loaded = false;
static void knobs_load()
{
    SqKnobs k;
    /* (defined but empty or "0" is OFF: `for v in "" 1; do SQ_X=$v ...` compared a kernel with itself twice in round 6) */
    auto flag = [](const char *name) { const char *v = getenv(name); return v != nullptr && *v != 0 && strcmp(v, "0") != 0; };
    auto num = [](const char *name, int unset) { const char *v = getenv(name); return v ? atoi(v) : unset; };
    k.span = num("SQ_SPAN", 1) != 0;
    k.span_split = num("SQ_SPAN_SPLIT", 1) != 0;
    k.span_sorted = num("SQ_SPAN_SORTED", -1);
    k.span_split_qc = (int)num("SQ_SPAN_SPLIT_QC", -1);
    k.span_w6 = num("SQ_SPAN_W6", -1);
    k.span_short = flag("SQ_SPAN_SHORT");
    k.pt_fused = num("SQ_PT_FUSED", 1);
    k.no_wide = flag("SQ_NO_WIDE");
    k.ring = flag("SQ_RING");
    k.no_ring = flag("SQ_NO_RING");
    k.no_ptq = flag("SQ_NO_PTQ");
    k.no_segments = flag("SQ_NO_SEGMENTS");
    k.long_spans = num("SQ_LONG", 1) != 0;
    k.dedup_sequential = flag("SQ_DEDUP_SEQUENTIAL");
    k.overrep_chain = flag("SQ_OVERREP_CHAIN");
    g_knobs = k;
    g_knobs_loaded = true;
}
const SqKnobs &sq_knobs()
{
    if (!g_knobs_loaded) knobs_load();
    return g_knobs;
}
SQ_EXPORT void sq_knobs_reload(void) { knobs_load(); }

SQ_EXPORT int sq_abi_version(void) { return SQ_ABI_VERSION; }
SQ_EXPORT const char *sq_last_error(void) { return g_last_error.c_str(); }
SQ_EXPORT size_t sq_last_error_length(void) { return g_last_error.size(); }

SQ_EXPORT sq_ctx *sq_init(int device)
{
    int count = 0;
    SQ_HIP_NULL(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) {
        sq_set_error("sq_init: device %d not available (%d visible)", device, count);
        return nullptr;
    }
    SQ_HIP_NULL(hipSetDevice(device));
    sq_ctx *ctx = new sq_ctx();
    ctx->device = device;
    hipDeviceProp_t prop;
    SQ_HIP_NULL(hipGetDeviceProperties(&prop, device));
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    SQ_HIP_NULL(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    SQ_HIP_NULL(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    SQ_HIP_NULL(hipStreamCreateWithFlags(&ctx->prep_stream, hipStreamNonBlocking));
    SQ_HIP_NULL(hipEventCreateWithFlags(&ctx->copied, hipEventDisableTiming));
    SQ_HIP_NULL(hipHostMalloc((void **)&ctx->pinned, 64 * sizeof(uint64_t), hipHostMallocDefault));
    SQ_HIP_NULL(hipHostMalloc((void **)&ctx->pinned_stats, SQ_STATS_N * sizeof(uint64_t), hipHostMallocDefault));
    return ctx;
}

SQ_EXPORT void sq_shutdown(sq_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamDestroy(ctx->stream);
    }
    if (ctx->copy_stream) { (void)hipStreamSynchronize(ctx->copy_stream); (void)hipStreamDestroy(ctx->copy_stream); }
    if (ctx->prep_stream) { (void)hipStreamSynchronize(ctx->prep_stream); (void)hipStreamDestroy(ctx->prep_stream); }
    if (ctx->feed_stream) { (void)hipStreamSynchronize(ctx->feed_stream); (void)hipStreamDestroy(ctx->feed_stream); }
    if (ctx->copied) (void)hipEventDestroy(ctx->copied);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->pinned_stats) (void)hipHostFree(ctx->pinned_stats);
    for (void *p : ctx->scratch)
        if (p) (void)hipFree(p);
    if (ctx->ahead.dev) sq_dev_put(ctx, ctx->ahead.dev);
    sq_dev_reclaim(ctx, true);
    for (auto &f : ctx->pool_free) (void)hipFree(f.p);
    for (auto &f : ctx->pool_live) (void)hipFree(f.p);   /* batches that outlive their context hold dangling blocks: as before */
    delete ctx;
}

SQ_EXPORT const char *sq_last_route(sq_ctx *ctx) { return ctx->route.c_str(); }
/* out[0..3]: hipMalloc and hipFree calls of the context's pool of device blocks so far, blocks idle in it, blocks waiting for their event */
SQ_EXPORT void sq_pool_counts(sq_ctx *ctx, uint64_t *out)
{
    out[0] = ctx->pool_mallocs;
    out[1] = ctx->pool_frees;
    out[2] = ctx->pool_free.size();
    out[3] = ctx->deferred.size();
}
SQ_EXPORT void sq_route_reset(sq_ctx *ctx) { ctx->route.clear(); }

SQ_EXPORT int sq_synchronize(sq_ctx *ctx)
{
    SQ_HIP(hipStreamSynchronize(ctx->stream));
    return SQ_OK;
}

SQ_EXPORT void *sq_stream_handle(sq_ctx *ctx) { return (void *)ctx->stream; }

/* ---- host-side record boundary ------------------------------------------- */

size_t sq_scan_newlines(const uint8_t *p, size_t n, uint32_t base, uint32_t *out, size_t cap, size_t *scanned, uint32_t *high);   /* sq_hostsimd.cpp */
int64_t sq_first_non_ascii_fast(const uint8_t *p, size_t n);
int64_t sq_split_range_ascii(const uint8_t *base, size_t start, size_t end_off, sq_meta *metas, size_t cap, size_t *consumed,
                             uint64_t stats[SQ_STATS_N], size_t ascii_from, int64_t *non_ascii);
static int64_t split_core(const uint8_t *base, size_t start, size_t end_off, sq_meta *metas, size_t cap, size_t *consumed,
                          uint64_t stats[SQ_STATS_N], size_t ascii_from, int64_t *first_high);

/* FastqParser_create_record_array, the record loop _qcmodule.c:1093-1171, over bytes
 * [start, end) of `base`; record_start is relative to `base`.  stats (may be NULL): bases,
 * longest read, longest name, longest record span, ~(shortest read) of the records so far.
 * The reference finds the four line ends of a record with four memchr calls; here the newline
 * positions of the range come from one vectorised scan (sq_scan_newlines), 1 K of them at a time,
 * and the record loop takes them as it needs them: same records, same errors in the same order. */
int64_t sq_split_range(const uint8_t *base, size_t start, size_t end_off, sq_meta *metas, size_t cap, size_t *consumed,
                       uint64_t stats[SQ_STATS_N])
{
    return sq_split_range_ascii(base, start, end_off, metas, cap, consumed, stats, (size_t)-1, nullptr);
}

/* ascii_from (offset in base, or (size_t)-1: no check): the bytes from there to the end of the range
 * are new (FastqParser_create_record_array checks what it has just read for ASCII before it looks
 * at records, :1055-1067): *non_ascii = offset of the first byte >= 0x80 among them, or -1.  The
 * check rides on the newline scan; bytes the scan did not reach (the record loop ended early)
 * are looked at separately.  When *non_ascii >= 0 the return value and the metas are to be
 * ignored: the reference raises before it parses. */
int64_t sq_split_range_ascii(const uint8_t *base, size_t start, size_t end_off, sq_meta *metas, size_t cap, size_t *consumed,
                             uint64_t stats[SQ_STATS_N], size_t ascii_from, int64_t *non_ascii)
{
    int64_t first_high = -1;
    const int64_t n = split_core(base, start, end_off, metas, cap, consumed, stats, ascii_from, &first_high);
    if (non_ascii) *non_ascii = first_high;
    return n;
}

/* The record loop itself (:1093-1171).  next_newline(from, &after): the first newline at or behind `from` inside the range,
 * or NULL -- the loop asks strictly left to right; `after`: the byte that follows that newline where the provider knows it,
 * else 0 (the loop then looks at the text under the conditions it always did); finish_ascii(): called once, whichever
 * way the loop ends. */
template <class NextNewline, class FinishAscii>
static int64_t record_loop(const uint8_t *base, size_t start, size_t end_off, sq_meta *metas, size_t cap, size_t *consumed,
                           uint64_t stats[SQ_STATS_N], NextNewline &&next_newline, FinishAscii &&finish_ascii)
{
    const uint8_t *end = base + end_off;
    const uint8_t *rec = base + start;
    int64_t n = 0;
    uint8_t rec_first = 0, after = 0;   /* rec[0] where the newline in front of it brought it along */
    while ((size_t)n < cap) {
        if (rec + 2 >= end) break; /* :1094 */
        if ((rec_first ? rec_first : rec[0]) != '@') {
            sq_set_error("Record does not start with @ but with %c", rec[0]);
            finish_ascii();
            return SQ_ERR_VALUE;
        }
        const uint8_t *name = rec + 1;
        const uint8_t *name_end = next_newline(name, after);
        if (!name_end) break;
        const uint8_t *seq = name_end + 1;
        const uint8_t *seq_end = next_newline(seq, after);
        if (!seq_end) break;
        const uint8_t *plus = seq_end + 1;
        if (plus < end && (after ? after : plus[0]) != '+') {
            sq_set_error("Record second header does not start with + but with %c", plus[0]);
            finish_ascii();
            return SQ_ERR_VALUE;
        }
        const uint8_t *plus_end = next_newline(plus, after);
        if (!plus_end) break;
        const uint8_t *qual = plus_end + 1;
        const uint8_t *qual_end = next_newline(qual, after);
        if (!qual_end) break;
        rec_first = after;
        if (seq_end - seq != qual_end - qual) {
            sq_set_error("Record sequence and qualities do not have equal length, %s", sq_py_repr_ascii((const char *)name, name_end - name).c_str());   /* :1141-1146: %R */
            finish_ascii();
            return SQ_ERR_VALUE;
        }
        if ((uint64_t)(qual_end - name) > UINT32_MAX) {
            sq_set_error("Total length of FASTQ record exceeds 4 GiB");
            finish_ascii();
            return SQ_ERR_OVERFLOW;
        }
        sq_meta *m = &metas[n++];
        m->record_start = (uint64_t)(name - base);
        m->name_length = (uint32_t)(name_end - name);
        m->sequence_offset = (uint32_t)(seq - name);
        m->sequence_length = (uint32_t)(seq_end - seq);
        m->qualities_offset = (uint32_t)(qual - name);
        m->tags_offset = (uint32_t)(qual_end - name);
        m->tags_length = 0;
        m->accumulated_error_rate = 0.0;
        if (stats) {
            const uint64_t L = m->sequence_length, span = (uint64_t)m->qualities_offset + L;
            stats[0] += L;
            if (L > stats[1]) stats[1] = L;
            if (m->name_length > stats[2]) stats[2] = m->name_length;
            if (span > stats[3]) stats[3] = span;
            if (~L > stats[4]) stats[4] = ~L;
            stats[5 + (L < SQ_LEN_BINS - 1 ? L : SQ_LEN_BINS - 1)]++;
        }
        rec = qual_end + 1;
    }
    if (consumed) *consumed = (size_t)(rec - (base + start));
    finish_ascii();
    return n;
}

static int64_t split_core(const uint8_t *base, size_t start, size_t end_off, sq_meta *metas, size_t cap, size_t *consumed,
                          uint64_t stats[SQ_STATS_N], size_t ascii_from, int64_t *first_high)
{
    const uint8_t *end = base + end_off;
    /* the newlines of the range, in order: every call hands out the next one (NULL: there is none).
       The record loop asks for them strictly left to right -- `from` is always one behind the
       newline it got last -- so no position has to be compared */
    uint32_t nl[2048];
    size_t nl_count = 0, nl_next = 0;
    const uint8_t *scan = base + start;    /* everything in front of it has been scanned */
    const uint8_t *chunk0 = scan;          /* nl[] holds offsets from here */
    const uint8_t *ascii_lo = ascii_from == (size_t)-1 ? nullptr : base + ascii_from;
    const uint8_t *high_at = nullptr;   /* the first byte >= 0x80 at or behind ascii_lo met so far */
    auto refill = [&]() -> bool {       /* false: the range has no more newlines */
        while (scan < end) {
            size_t scanned = 0;
            uint32_t high = UINT32_MAX;
            chunk0 = scan;
            nl_next = 0;
            nl_count = sq_scan_newlines(scan, std::min<size_t>((size_t)(end - scan), (size_t)1 << 30), 0, nl, 2048, &scanned, &high);
            if (high != UINT32_MAX && high_at == nullptr && ascii_lo) {
                const uint8_t *h = scan + high;
                if (h < ascii_lo) {   /* in front of the new bytes (checked by an earlier call): look again from there */
                    const int64_t again = ascii_lo < scan + scanned ? sq_first_non_ascii_fast(ascii_lo, (size_t)(scan + scanned - ascii_lo)) : -1;
                    h = again >= 0 ? ascii_lo + again : nullptr;
                }
                if (h) high_at = h;
            }
            scan += scanned;
            if (nl_count) return true;
        }
        return false;
    };
    auto next_newline = [&](const uint8_t *, uint8_t &after) -> const uint8_t * {
        after = 0;
        if (nl_next == nl_count && !refill()) return nullptr;
        return chunk0 + nl[nl_next++];
    };
    /* what of the new bytes the scan has not seen (the loop ended on max_records, or on an error) */
    auto finish_ascii = [&]() {
        if (!ascii_lo || !first_high) return;
        if (!high_at) {
            const uint8_t *from = std::max(scan, ascii_lo);
            if (from < end) {
                const int64_t r = sq_first_non_ascii_fast(from, (size_t)(end - from));
                if (r >= 0) high_at = from + r;
            }
        }
        *first_high = high_at ? (int64_t)(high_at - base) : -1;
    };
    return record_loop(base, start, end_off, metas, cap, consumed, stats, next_newline, finish_ascii);
}

/* sq_split_range_ascii over a range whose newlines are known already: `pieces` (sorted, back to back) cover [start,
 * end_off) of `base` -- what the feeder's workers noted while they copied the text in (SqNlPiece, sq_feed.hip).  The same
 * records, the same errors in the same order; the text itself is only looked at where a record is checked. */
int64_t sq_split_range_indexed(const uint8_t *base, size_t start, size_t end_off, sq_meta *metas, size_t cap, size_t *consumed,
                               uint64_t stats[SQ_STATS_N], size_t ascii_from, int64_t *non_ascii, const SqNlPiece *pieces, size_t n_pieces)
{
    size_t k = 0, j = 0;   /* the next newline to hand out: entry j of piece k */
    while (k < n_pieces && pieces[k].to <= start) k++;
    if (k < n_pieces) j = (size_t)(std::lower_bound(pieces[k].nl, pieces[k].nl + pieces[k].n_nl, (uint32_t)start) - pieces[k].nl);
    auto next_newline = [&](const uint8_t *, uint8_t &after) -> const uint8_t * {
        after = 0;
        while (k < n_pieces) {
            if (j < pieces[k].n_nl) {
                const uint32_t at = pieces[k].nl[j];
                if (at >= end_off) return nullptr;
                if ((size_t)at + 1 < end_off) after = pieces[k].after[j];   /* (behind the range's end nothing is looked at) */
                j++;
                return base + at;
            }
            k++;
            j = 0;
        }
        return nullptr;
    };
    /* :1055-1067: the first byte >= 0x80 among the bytes [ascii_from, end_off) */
    auto finish_ascii = [&]() {
        if (!non_ascii) return;
        *non_ascii = -1;
        if (ascii_from == (size_t)-1) return;
        for (size_t q = 0; q < n_pieces; q++) {
            const SqNlPiece &pc = pieces[q];
            if (pc.to <= ascii_from || pc.first_high == UINT32_MAX) continue;
            if (pc.from >= end_off) break;
            size_t at = pc.first_high;
            if (at < ascii_from) {   /* the piece's first one lies in front of the new bytes: look again behind them */
                const size_t hi = std::min(pc.to, end_off);
                const int64_t r = ascii_from < hi ? sq_first_non_ascii_fast(base + ascii_from, hi - ascii_from) : -1;
                if (r < 0) continue;
                at = ascii_from + (size_t)r;
            }
            if (at < end_off) *non_ascii = (int64_t)at;
            break;
        }
    };
    return record_loop(base, start, end_off, metas, cap, consumed, stats, next_newline, finish_ascii);
}

SQ_EXPORT int64_t sq_fastq_split(const uint8_t *buf, size_t len, sq_meta *metas, size_t cap,
                                 size_t *consumed)
{
    return sq_split_range(buf, 0, len, metas, cap, consumed, nullptr);
}

SQ_EXPORT int64_t sq_first_non_ascii(const uint8_t *buf, size_t len)
{
    return sq_first_non_ascii_fast(buf, len);
}

/* fastq_names_are_mates, _qcmodule.c:777-800 */
static bool names_are_mates(const uint8_t *n1, size_t l1, const uint8_t *n2, size_t l2)
{
    size_t id = 0;
    while (id < l1 && n1[id] != ' ' && n1[id] != '\t') id++;
    if (l2 < id) return false;
    if (l2 > id && !(n2[id] == ' ' || n2[id] == '\t')) return false;
    if (id > 0) {
        uint8_t a = n1[id - 1], b = n2[id - 1];
        if ((a == '1' || a == '2') && (b == '1' || b == '2')) id -= 1;
    }
    return memcmp(n1, n2, id) == 0;
}

SQ_EXPORT int sq_names_are_mates(const uint8_t *buf1, const sq_meta *metas1, const uint8_t *buf2,
                                 const sq_meta *metas2, size_t n)
{
    for (size_t i = 0; i < n; i++)
        if (!names_are_mates(buf1 + metas1[i].record_start, metas1[i].name_length,
                             buf2 + metas2[i].record_start, metas2[i].name_length))
            return 0;
    return 1;
}

/* ---- batches ----------------------------------------------------------------- */

/* per-batch statistics on the device: [0] total bases [1] max length
 * [2] max name length [3] max record span (name start .. quality end)
 * [4] ~min length (kept as a max so that one memset(0) initialises all) */
__global__ void k_batch_stats(const sq_meta *metas, size_t n, unsigned long long *out)
{
    __shared__ uint32_t l_len[SQ_LEN_BINS];   /* reads per length (out[5 ..]) */
    for (int i = threadIdx.x; i < SQ_LEN_BINS; i += blockDim.x) l_len[i] = 0;
    __syncthreads();
    unsigned long long bases = 0, maxlen = 0, maxname = 0, maxspan = 0, minlen_inv = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        sq_meta m = metas[i];
        bases += m.sequence_length;
        if (m.sequence_length > maxlen) maxlen = m.sequence_length;
        if (m.name_length > maxname) maxname = m.name_length;
        unsigned long long span = (unsigned long long)m.qualities_offset + m.sequence_length;
        if (span > maxspan) maxspan = span;
        unsigned long long inv = ~(unsigned long long)m.sequence_length;
        if (inv > minlen_inv) minlen_inv = inv;
        /* one read length in the whole wave (a file of untrimmed reads): one lane counts for all, 64 atomics on
           one LDS address would take their turns */
        const uint32_t bin = m.sequence_length < SQ_LEN_BINS - 1 ? m.sequence_length : SQ_LEN_BINS - 1;
        const unsigned long long active = __ballot(1), same = __ballot(bin == (uint32_t)__shfl(bin, __ffsll((long long)active) - 1));
        if (same == active) {
            if ((int)(threadIdx.x & 63) == __ffsll((long long)active) - 1) atomicAdd(&l_len[bin], (uint32_t)__popcll(active));
        } else {
            atomicAdd(&l_len[bin], 1u);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        bases += __shfl_down(bases, off);
        unsigned long long a = __shfl_down(maxlen, off), b = __shfl_down(maxname, off),
                           c = __shfl_down(maxspan, off), d = __shfl_down(minlen_inv, off);
        maxlen = a > maxlen ? a : maxlen;
        maxname = b > maxname ? b : maxname;
        maxspan = c > maxspan ? c : maxspan;
        minlen_inv = d > minlen_inv ? d : minlen_inv;
    }
    /* one set of global atomics per workgroup (they all go to the same five addresses) */
    __shared__ unsigned long long part[5][16];
    const int wave = threadIdx.x >> 6, nwaves = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) {
        part[0][wave] = bases; part[1][wave] = maxlen; part[2][wave] = maxname;
        part[3][wave] = maxspan; part[4][wave] = minlen_inv;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < nwaves; w++) {
            bases += part[0][w];
            maxlen = part[1][w] > maxlen ? part[1][w] : maxlen;
            maxname = part[2][w] > maxname ? part[2][w] : maxname;
            maxspan = part[3][w] > maxspan ? part[3][w] : maxspan;
            minlen_inv = part[4][w] > minlen_inv ? part[4][w] : minlen_inv;
        }
        atomicAdd(&out[0], bases);
        atomicMax(&out[1], maxlen);
        atomicMax(&out[2], maxname);
        atomicMax(&out[3], maxspan);
        atomicMax(&out[4], minlen_inv);
    }
    for (int i = threadIdx.x; i < SQ_LEN_BINS; i += blockDim.x)
        if (l_len[i]) atomicAdd(&out[5 + i], (unsigned long long)l_len[i]);
}

/* k_batch_stats over a batch's metas, queued on the context's stream; stats_take() behind the next
 * synchronisation of that stream puts the numbers into the batch */
static int stats_issue(sq_ctx *ctx, const sq_meta *d_metas, size_t n)
{
    unsigned long long *d_out = (unsigned long long *)sq_scratch(ctx, 21, SQ_STATS_N * 8);
    if (!d_out) { sq_set_error("out of device memory"); return SQ_ERR_MEMORY; }
    SQ_HIP(hipMemsetAsync(d_out, 0, SQ_STATS_N * 8, ctx->stream));
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_batch_stats, dim3(blocks ? blocks : 1), dim3(256), 0, ctx->stream, d_metas, n, d_out);
    SQ_HIP(hipMemcpyAsync(ctx->pinned_stats, d_out, SQ_STATS_N * 8, hipMemcpyDeviceToHost, ctx->stream));
    return SQ_OK;
}

static void stats_take(sq_ctx *ctx, sq_batch *b)
{
    const uint64_t *h = ctx->pinned_stats;
    b->total_bases = h[0];
    b->max_length = h[1];
    b->max_name_length = h[2];
    b->max_record_span = h[3];
    b->min_length = ~h[4];
    b->len_hist.resize(SQ_LEN_BINS);
    for (int i = 0; i < SQ_LEN_BINS; i++) b->len_hist[i] = (uint32_t)h[5 + i];
}

SQ_EXPORT sq_batch *sq_batch_upload(sq_ctx *ctx, const uint8_t *buf, size_t buf_len,
                                    const sq_meta *metas, size_t n)
{
    sq_batch *b = new sq_batch();
    b->ctx = ctx;
    b->buf_len = buf_len;
    b->n = n;
    b->owns = b->slack = true;
    b->len_hist.assign(SQ_LEN_BINS, 0);
    for (size_t i = 0; i < n; i++) {
        const sq_meta &m = metas[i];
        uint64_t span = (uint64_t)m.qualities_offset + m.sequence_length;
        uint64_t seq_end = (uint64_t)m.sequence_offset + m.sequence_length;
        if (m.record_start + span > buf_len || m.record_start + seq_end > buf_len ||
            m.record_start + m.name_length > buf_len) {
            sq_set_error("sq_batch_upload: record %zu lies outside the buffer", i);
            delete b;
            return nullptr;
        }
        b->total_bases += m.sequence_length;
        b->len_hist[m.sequence_length < SQ_LEN_BINS - 1 ? m.sequence_length : SQ_LEN_BINS - 1]++;
        if (m.sequence_length > b->max_length) b->max_length = m.sequence_length;
        if (i == 0 || m.sequence_length < b->min_length) b->min_length = m.sequence_length;
        if (m.name_length > b->max_name_length) b->max_name_length = m.name_length;
        if (span > b->max_record_span) b->max_record_span = span;
    }
    /* 64 spare bytes so that wide loads near the end stay inside the allocation */
    if (hipMalloc((void **)&b->d_buf, buf_len + 64) != hipSuccess ||
        hipMalloc((void **)&b->d_metas, (n ? n : 1) * sizeof(sq_meta)) != hipSuccess) {
        sq_set_error("sq_batch_upload: out of device memory");
        if (b->d_buf) (void)hipFree(b->d_buf);
        delete b;
        return nullptr;
    }
    b->h_buf.assign(buf, buf + buf_len);
    b->h_metas.assign(metas, metas + n);
    if (buf_len) SQ_HIP_NULL(hipMemcpyAsync(b->d_buf, b->h_buf.data(), buf_len, hipMemcpyHostToDevice, ctx->stream));
    SQ_HIP_NULL(hipMemsetAsync(b->d_buf + buf_len, 0, 64, ctx->stream));
    if (n) SQ_HIP_NULL(hipMemcpyAsync(b->d_metas, b->h_metas.data(), n * sizeof(sq_meta), hipMemcpyHostToDevice, ctx->stream));
    SQ_HIP_NULL(hipEventCreateWithFlags(&b->ready, hipEventDisableTiming));
    SQ_HIP_NULL(hipEventRecord(b->ready, ctx->stream));
    return b;
}

SQ_EXPORT sq_batch *sq_batch_wrap_device(sq_ctx *ctx, const void *d_buf, size_t buf_len,
                                         void *d_metas, size_t n)
{
    sq_batch *b = new sq_batch();
    b->ctx = ctx;
    b->d_buf = (uint8_t *)d_buf;
    b->d_metas = (sq_meta *)d_metas;
    b->buf_len = buf_len;
    b->n = n;
    b->owns = false;
    if (n) {
        if (stats_issue(ctx, b->d_metas, n) != SQ_OK) { delete b; return nullptr; }
        SQ_HIP_NULL(hipStreamSynchronize(ctx->stream));
        stats_take(ctx, b);
    }
    return b;
}

SQ_EXPORT sq_batch *sq_batch_view(sq_batch *parent, size_t first, size_t n)
{
    if (!parent || first > parent->n || n > parent->n - first) { sq_set_error("sq_batch_view: records out of range"); return nullptr; }
    sq_batch *b = sq_batch_wrap_device(parent->ctx, parent->d_buf, parent->buf_len, parent->d_metas + first, n);
    if (b) b->slack = parent->slack;    /* the same text: the same spare bytes behind it */
    return b;
}

/* ---- FASTQ record split on the device ----------------------------------------------
 * pass 1: newlines per 16 KiB block (+ first non-ASCII byte), exclusive scan of the
 * block counts; pass 2: every thread rescans its 64 bytes and writes the positions of
 * its newlines at (block prefix + in-block prefix); pass 3: one thread per record turns
 * four consecutive newline positions into a FastqMeta and checks '@', '+' and the
 * equal-length rule. */
namespace {

constexpr uint32_t SPLIT_THREADS = 256, SPLIT_BYTES_PER_THREAD = 64;
constexpr uint64_t SPLIT_BLOCK_BYTES = (uint64_t)SPLIT_THREADS * SPLIT_BYTES_PER_THREAD;

/* bit 8k+7 set for every byte k of w that equals `c` (exact for any byte value) */
__device__ __forceinline__ uint32_t eq_bytes(uint32_t w, uint32_t c4)
{
    const uint32_t x = w ^ c4;
    return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);
}

__device__ __forceinline__ void load_segment(const uint8_t *text, uint64_t len, uint64_t off, uint32_t w[16])
{
    if (off + 64 <= len) {
        const uint4 *p = (const uint4 *)(text + off);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint4 v;
            __builtin_memcpy(&v, p + i, 16);
            w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            uint32_t v = 0;
            for (int k = 0; k < 4; k++) {
                const uint64_t at = off + 4 * i + k;
                if (at < len) v |= (uint32_t)text[at] << (8 * k);
            }
            w[i] = v;
        }
    }
}

__global__ void __launch_bounds__(SPLIT_THREADS)
k_split_count(const uint8_t *text, uint64_t len, unsigned long long *block_counts,
              unsigned long long *first_non_ascii)
{
    typedef hipcub::BlockReduce<uint32_t, SPLIT_THREADS> Reduce;
    __shared__ typename Reduce::TempStorage tmp;
    const uint64_t off = (uint64_t)blockIdx.x * SPLIT_BLOCK_BYTES + (uint64_t)threadIdx.x * SPLIT_BYTES_PER_THREAD;
    uint32_t count = 0;
    if (off < len) {
        uint32_t w[16];
        load_segment(text, len, off, w);
        uint32_t high = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            count += __popc(eq_bytes(w[i], 0x0A0A0A0Au));
            high |= w[i];
        }
        if (high & 0x80808080u) { /* string_is_ascii :203-237 */
            for (int i = 0; i < 16; i++)
                for (int k = 0; k < 4; k++)
                    if ((w[i] >> (8 * k)) & 0x80) {
                        atomicMin(first_non_ascii, (unsigned long long)(off + 4 * i + k));
                        i = 16;
                        break;
                    }
        }
    }
    const uint32_t total = Reduce(tmp).Sum(count);
    if (threadIdx.x == 0) block_counts[blockIdx.x] = total;
}

__global__ void __launch_bounds__(SPLIT_THREADS)
k_split_positions(const uint8_t *text, uint64_t len, const unsigned long long *block_prefix,
                  unsigned long long *positions)
{
    typedef hipcub::BlockScan<uint32_t, SPLIT_THREADS> Scan;
    __shared__ typename Scan::TempStorage tmp;
    const uint64_t off = (uint64_t)blockIdx.x * SPLIT_BLOCK_BYTES + (uint64_t)threadIdx.x * SPLIT_BYTES_PER_THREAD;
    uint32_t w[16], count = 0;
    if (off < len) {
        load_segment(text, len, off, w);
#pragma unroll
        for (int i = 0; i < 16; i++) count += __popc(eq_bytes(w[i], 0x0A0A0A0Au));
    }
    uint32_t before = 0;
    Scan(tmp).ExclusiveSum(count, before);
    if (count) {
        unsigned long long *out = positions + block_prefix[blockIdx.x] + before;
        for (int i = 0; i < 16; i++) {
            uint32_t m = eq_bytes(w[i], 0x0A0A0A0Au);
            while (m) {
                const int bit = __ffs((int)m) - 1;
                m &= m - 1;
                *out++ = off + 4 * i + (bit >> 3);
            }
        }
    }
}

enum { SPLIT_OK = 0, SPLIT_NO_AT = 1, SPLIT_NO_PLUS = 2, SPLIT_LENGTHS = 3, SPLIT_TOO_LONG = 4 };

__global__ void k_split_metas(const uint8_t *text, const unsigned long long *nl, uint64_t n_records,
                              sq_meta *metas, unsigned long long *first_bad)
{
    for (uint64_t r = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; r < n_records;
         r += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t start = r ? nl[4 * r - 1] + 1 : 0;
        const uint64_t e1 = nl[4 * r], e2 = nl[4 * r + 1], e3 = nl[4 * r + 2], e4 = nl[4 * r + 3];
        const uint64_t name = start + 1, seq = e1 + 1, plus = e2 + 1, qual = e3 + 1;
        unsigned long long bad = 0;
        if (text[start] != '@') bad = SPLIT_NO_AT;                 /* :1097-1103 */
        else if (e1 < name) bad = SPLIT_NO_AT;
        else if (text[plus] != '+') bad = SPLIT_NO_PLUS;            /* :1119-1127 */
        else if (e2 - seq != e4 - qual) bad = SPLIT_LENGTHS;        /* :1140-1150 */
        else if (e4 - name > 0xFFFFFFFFull) bad = SPLIT_TOO_LONG;
        if (bad) atomicMin(first_bad, (unsigned long long)((r << 3) | bad));
        sq_meta m;
        m.record_start = name;
        m.name_length = (uint32_t)(e1 - name);
        m.sequence_offset = (uint32_t)(seq - name);
        m.sequence_length = (uint32_t)(e2 - seq);
        m.qualities_offset = (uint32_t)(qual - name);
        m.tags_offset = (uint32_t)(e4 - name);
        m.tags_length = 0;
        m.accumulated_error_rate = 0.0;
        metas[r] = m;
    }
}

sq_batch *split_on_device(sq_ctx *ctx, uint8_t *d_text, bool owns_text, const uint8_t *h_text, size_t len,
                          size_t *consumed)
{
    /* d_text (when owned) and the metas are blocks of the context's pool, the temporaries live in its
       scratch slots: no hipMalloc / hipFree per buffer (round 3; a 64 MiB buffer spent more time in those
       than in its kernels) */
    auto fail = [&](sq_batch *b) -> sq_batch * {
        if (owns_text && d_text && !b) sq_dev_put(ctx, d_text);
        if (b) sq_batch_free(b);
        return nullptr;
    };
    if (consumed) *consumed = 0;
    const uint64_t n_blocks = (len + SPLIT_BLOCK_BYTES - 1) / SPLIT_BLOCK_BYTES;
    unsigned long long *d_counts = nullptr, *d_prefix = nullptr, *d_flags = nullptr, *d_nl = nullptr;
    sq_batch *b = new sq_batch();
    b->ctx = ctx;
    b->d_buf = d_text;
    b->buf_len = len;
    b->owns = b->slack = owns_text;
    b->pooled = true;
    if (len == 0 || n_blocks == 0) {
        if (!(b->d_metas = (sq_meta *)sq_dev_get(ctx, sizeof(sq_meta)))) return fail(b);
        b->owns_metas = true;
        return b;
    }
    d_counts = (unsigned long long *)sq_scratch(ctx, 18, (2 * (n_blocks + 1) + 2) * 8);
    if (!d_counts) {
        sq_set_error("sq_batch_from_fastq: out of device memory");
        return fail(b);
    }
    d_prefix = d_counts + n_blocks + 1;
    d_flags = d_prefix + n_blocks + 1;
    (void)hipMemsetAsync(d_flags, 0xFF, 16, ctx->stream);
    (void)hipMemsetAsync(d_counts + n_blocks, 0, 8, ctx->stream);
    hipLaunchKernelGGL(k_split_count, dim3((unsigned)n_blocks), dim3(SPLIT_THREADS), 0, ctx->stream, d_text,
                       (uint64_t)len, d_counts, d_flags);
    size_t temp_bytes = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, temp_bytes, d_counts, d_prefix, (int)(n_blocks + 1), ctx->stream);
    void *d_temp = sq_scratch(ctx, 19, temp_bytes ? temp_bytes : 8);
    if (!d_temp) { sq_set_error("sq_batch_from_fastq: out of device memory"); return fail(b); }
    (void)hipcub::DeviceScan::ExclusiveSum(d_temp, temp_bytes, d_counts, d_prefix, (int)(n_blocks + 1), ctx->stream);
    unsigned long long h[3] = {0, 0, 0};
    (void)hipMemcpyAsync(&ctx->pinned[40], d_prefix + n_blocks, 8, hipMemcpyDeviceToHost, ctx->stream);
    (void)hipMemcpyAsync(&ctx->pinned[41], d_flags, 8, hipMemcpyDeviceToHost, ctx->stream);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) { sq_set_error("FASTQ split failed on the device"); return fail(b); }
    h[0] = ctx->pinned[40];
    h[1] = ctx->pinned[41];
    if (h[1] != ~0ULL) { /* :1055-1067 */
        uint8_t c = 0;
        if (h_text) c = h_text[h[1]];
        else (void)hipMemcpy(&c, d_text + h[1], 1, hipMemcpyDeviceToHost);
        sq_set_error("Found non-ASCII character in file: %c", (char)c);
        return fail(b);
    }
    const uint64_t n_newlines = h[0], n_records = n_newlines / 4;
    b->n = n_records;
    b->owns_metas = true;
    b->d_metas = (sq_meta *)sq_dev_get(ctx, (n_records ? n_records : 1) * sizeof(sq_meta));
    d_nl = (unsigned long long *)sq_scratch(ctx, 20, (n_newlines ? n_newlines : 1) * 8);
    if (!b->d_metas || !d_nl) {
        sq_set_error("sq_batch_from_fastq: out of device memory");
        return fail(b);
    }
    auto byte_at = [&](uint64_t at) -> uint8_t {
        uint8_t c = 0;
        if (h_text) return h_text[at];
        (void)hipMemcpy(&c, d_text + at, 1, hipMemcpyDeviceToHost);
        return c;
    };
    if (n_newlines)
        hipLaunchKernelGGL(k_split_positions, dim3((unsigned)n_blocks), dim3(SPLIT_THREADS), 0, ctx->stream,
                           d_text, (uint64_t)len, d_prefix, d_nl);
    uint64_t tail_start = 0;
    if (n_records) {
        const int blocks = (int)((n_records + 255) / 256 > 65535 ? 65535 : (n_records + 255) / 256);
        hipLaunchKernelGGL(k_split_metas, dim3(blocks), dim3(256), 0, ctx->stream, d_text, d_nl, n_records,
                           b->d_metas, d_flags + 1);
        if (stats_issue(ctx, b->d_metas, (size_t)n_records) != SQ_OK) return fail(b);
        (void)hipMemcpyAsync(&ctx->pinned[42], d_flags + 1, 8, hipMemcpyDeviceToHost, ctx->stream);
        (void)hipMemcpyAsync(&ctx->pinned[43], d_nl + 4 * n_records - 1, 8, hipMemcpyDeviceToHost, ctx->stream);
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) {
            sq_set_error("FASTQ split failed on the device");
            return fail(b);
        }
        stats_take(ctx, b);
        tail_start = ctx->pinned[43] + 1;
        const unsigned long long bad = ctx->pinned[42];
        if (bad != ~0ULL) {
            const uint64_t r = bad >> 3;
            sq_meta m;
            (void)hipMemcpy(&m, b->d_metas + r, sizeof(sq_meta), hipMemcpyDeviceToHost);
            switch (bad & 7) {
                case SPLIT_NO_AT:
                    sq_set_error("Record does not start with @ but with %c", (char)byte_at(m.record_start - 1));
                    break;
                case SPLIT_NO_PLUS:
                    sq_set_error("Record second header does not start with + but with %c",
                                 (char)byte_at(m.record_start + m.sequence_offset + m.sequence_length + 1));
                    break;
                case SPLIT_LENGTHS: {
                    std::string name(m.name_length, ' ');
                    if (m.name_length)
                        (void)hipMemcpy(&name[0], d_text + m.record_start, m.name_length, hipMemcpyDeviceToHost);
                    sq_set_error("Record sequence and qualities do not have equal length, %s", sq_py_repr_ascii(name.data(), name.size()).c_str());   /* :1141-1146: %R */
                    break;
                }
                default:
                    sq_set_error("Total length of FASTQ record exceeds 4 GiB");
            }
            return fail(b);
        }
    }
    /* the incomplete record after the last complete one: the reference looks at its
       '@' (:1097) and, when two of its lines are complete, at its '+' (:1119) before it
       finds out that the record is incomplete */
    if (tail_start + 2 < len) {
        const uint8_t c = byte_at(tail_start);
        if (c != '@') {
            sq_set_error("Record does not start with @ but with %c", (char)c);
            return fail(b);
        }
        if (n_newlines % 4 >= 2) {
            unsigned long long second = 0;
            (void)hipMemcpyAsync(&second, d_nl + 4 * n_records + 1, 8, hipMemcpyDeviceToHost, ctx->stream);
            (void)hipStreamSynchronize(ctx->stream);
            if (second + 1 < len && byte_at(second + 1) != '+') {
                sq_set_error("Record second header does not start with + but with %c", (char)byte_at(second + 1));
                return fail(b);
            }
        }
    }
    if (consumed) *consumed = (size_t)tail_start;
    return b;
}

} // namespace

namespace {
/* dst[0, n) = src[0, n), both anywhere: 16 bytes per thread and step where dst is aligned */
__global__ void __launch_bounds__(256) k_copy_bytes(uint8_t *dst, const uint8_t *src, uint64_t n)
{
    const uint64_t lead = (16 - ((uintptr_t)dst & 15)) & 15, first = lead < n ? lead : n;
    const uint64_t chunks = (n - first) / 16, rest = first + 16 * chunks;
    const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = tid; i < chunks; i += stride) {
        uint4 v;
        __builtin_memcpy(&v, src + first + 16 * i, 16);
        *(uint4 *)(dst + first + 16 * i) = v;
    }
    if (tid < first) dst[tid] = src[tid];
    if (tid < n - rest) dst[rest + tid] = src[rest + tid];
}
} // namespace

SQ_EXPORT sq_batch *sq_batch_from_fastq_ahead(sq_ctx *ctx, const uint8_t *text, size_t len, size_t *consumed,
                                              const uint8_t *ahead, size_t ahead_len)
{
    uint8_t *d_text = (uint8_t *)sq_dev_get(ctx, len + 64);
    if (!d_text) {
        sq_set_error("sq_batch_from_fastq: out of device memory");
        return nullptr;
    }
    /* the upload runs on a stream of its own: the passes over the batch before (queued on
       ctx->stream by the caller, who is back here for the next buffer) go on beside it; the split
       waits for it.  What an earlier call has sent ahead is already in HBM: a copy within the device. */
    sq_ctx::Ahead &A = ctx->ahead;
    uint8_t *spent = nullptr;
    /* wherever this call fails: the block of this buffer (and the one sent ahead, once taken off the context)
       go back to the pool, behind whatever is still reading or writing them */
    auto fail = [&]() -> sq_batch * {
        (void)hipStreamSynchronize(ctx->copy_stream);
        (void)hipStreamSynchronize(ctx->stream);
        sq_dev_put(ctx, d_text);
        if (spent) sq_dev_put(ctx, spent);
        return nullptr;
    };
#define SQ_HIP_FAIL(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { sq_set_error("%s: %s", #call, hipGetErrorString(e_)); return fail(); } } while (0)
    /* what was sent ahead may begin in front of this buffer and reach beyond it (the caller did not know
       its leftover then): the overlap is copied inside the device, the rest comes from the host */
    const uint8_t *lo = A.dev ? std::max(A.host, text) : nullptr, *hi = A.dev ? std::min(A.host + A.len, text + len) : nullptr;
    const bool covered = A.dev && lo < hi;
    if (covered) {
        const size_t head = (size_t)(lo - text), tail = (size_t)(text + len - hi);
        if (head) SQ_HIP_FAIL(hipMemcpyAsync(d_text, text, head, hipMemcpyHostToDevice, ctx->copy_stream));
        if (tail) SQ_HIP_FAIL(hipMemcpyAsync(d_text + (hi - text), hi, tail, hipMemcpyHostToDevice, ctx->copy_stream));
    } else if (len) {
        SQ_HIP_FAIL(hipMemcpyAsync(d_text, text, len, hipMemcpyHostToDevice, ctx->copy_stream));
    }
    SQ_HIP_FAIL(hipEventRecord(ctx->copied, ctx->copy_stream));
    SQ_HIP_FAIL(hipStreamWaitEvent(ctx->stream, ctx->copied, 0));   /* behind the upload sent ahead too: same stream */
    SQ_HIP_FAIL(hipMemsetAsync(d_text + len, 0, 64, ctx->stream));
    if (covered)   /* a kernel, not hipMemcpyDeviceToDevice: that one goes through the copy engines at ~60 GB/s */
        hipLaunchKernelGGL(k_copy_bytes, dim3(2048), dim3(256), 0, ctx->stream, d_text + (lo - text),
                           (const uint8_t *)A.dev + (lo - A.host), (uint64_t)(hi - lo));
    spent = A.dev;   /* used, or sent for a buffer that never came: back to the pool when the copy kernel has read it */
    A = sq_ctx::Ahead();
    if (ahead && ahead_len) {
        /* the caller's next buffer: on its way while this one is split and counted (into a block of its
           own, so the upload depends on nothing that is queued) */
        A.dev = (uint8_t *)sq_dev_get(ctx, ahead_len);
        if (A.dev) {
            if (hipMemcpyAsync(A.dev, ahead, ahead_len, hipMemcpyHostToDevice, ctx->copy_stream) == hipSuccess) {
                A.host = ahead;
                A.len = ahead_len;
            } else {
                sq_dev_put(ctx, A.dev);
                A = sq_ctx::Ahead();
            }
        }
    }
    sq_batch *b = split_on_device(ctx, d_text, true, text, len, consumed);
    if (spent) {
        (void)hipStreamSynchronize(ctx->stream);   /* the split has done that already unless it failed early */
        sq_dev_put(ctx, spent);
    }
    return b;
#undef SQ_HIP_FAIL
}

/* What sq_batch_from_fastq_ahead has sent ahead is forgotten: the caller's next buffer will not be the one it
 * named (a parser that ends, fails or is rewound; its pages may be handed to somebody else).  Without this a later
 * call whose text happens to lie at the same host address would be assembled from the stale copy in HBM. */
SQ_EXPORT void sq_ahead_drop(sq_ctx *ctx)
{
    if (!ctx || !ctx->ahead.dev) return;
    (void)hipStreamSynchronize(ctx->copy_stream);   /* the upload may still be writing the block */
    sq_dev_put(ctx, ctx->ahead.dev);
    ctx->ahead = sq_ctx::Ahead();
}

SQ_EXPORT sq_batch *sq_batch_from_fastq(sq_ctx *ctx, const uint8_t *text, size_t len, size_t *consumed)
{
    return sq_batch_from_fastq_ahead(ctx, text, len, consumed, nullptr, 0);
}

SQ_EXPORT sq_batch *sq_batch_from_fastq_device(sq_ctx *ctx, const void *d_text, size_t len, size_t *consumed)
{
    return split_on_device(ctx, (uint8_t *)d_text, false, nullptr, len, consumed);
}

/* ---- BAM records ------------------------------------------------------
 * The record walk of BamParser__next__ (_qcmodule.c:1601-1681) is a pointer chase through
 * the block_size fields: it stays on the host and yields the offsets of the records to
 * decode.  The decode is data parallel: sizes per record, exclusive scan, then one wave
 * per record writes name | sequence | qualities | tags. */
SQ_EXPORT int64_t sq_bam_scan(const uint8_t *bam, size_t len, uint64_t *offsets, size_t cap, size_t *consumed,
                              uint64_t *skipped)
{
    const uint8_t *rec = bam, *end = bam + len;
    int64_t n = 0;
    uint64_t skip = 0;
    for (;;) {
        if (rec + 4 >= end) break; /* :1602 */
        uint32_t block_size;
        memcpy(&block_size, rec, 4);
        const uint8_t *rec_end = rec + 4 + block_size;
        if (rec_end > end) break;
        if (block_size < 32) {
            sq_set_error("BAM record of %u bytes is shorter than its fixed fields", block_size);
            return SQ_ERR_VALUE;
        }
        uint16_t flag;
        memcpy(&flag, rec + 18, 2);
        if (flag & (0x100 | 0x800)) { /* BAM_FSECONDARY | BAM_FSUPPLEMENTARY, :1262,1611 */
            rec = rec_end;
            skip++;
            continue;
        }
        if (offsets) {
            if ((size_t)n == cap) break;
            offsets[n] = (uint64_t)(rec - bam);
        }
        n++;
        rec = rec_end;
    }
    if (consumed) *consumed = (size_t)(rec - bam);
    if (skipped) *skipped = skip;
    return n;
}

namespace {

__device__ __forceinline__ uint32_t bam_le32(const uint8_t *p)
{
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

struct BamFields {
    uint32_t name_len, l_seq, tags_len;
    uint64_t name, seq, qual, tags; /* offsets into the BAM bytes */
};

__device__ BamFields bam_fields(const uint8_t *bam, uint64_t off)
{
    const uint8_t *rec = bam + off;
    BamFields f;
    const uint32_t block_size = bam_le32(rec), l_read_name = rec[12];
    const uint32_t n_cigar = (uint32_t)rec[16] | ((uint32_t)rec[17] << 8);
    f.l_seq = bam_le32(rec + 20);
    f.name = off + 36;
    f.seq = f.name + l_read_name + 4ull * n_cigar;
    f.qual = f.seq + (f.l_seq + 1) / 2;
    f.tags = f.qual + f.l_seq;
    f.tags_len = (uint32_t)(off + 4 + block_size - f.tags);
    f.name_len = l_read_name ? l_read_name - 1 : 0; /* without the terminating NUL, :1633 */
    return f;
}

__global__ void k_bam_sizes(const uint8_t *bam, const unsigned long long *offsets, uint64_t n,
                            unsigned long long *sizes)
{
    for (uint64_t r = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; r < n;
         r += (uint64_t)gridDim.x * blockDim.x) {
        const BamFields f = bam_fields(bam, offsets[r]);
        sizes[r] = (unsigned long long)f.name_len + 2ull * f.l_seq + f.tags_len;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) sizes[n] = 0;
}

/* one wave per record; lanes take consecutive output bytes */
__global__ void k_bam_decode(const uint8_t *bam, const unsigned long long *offsets,
                             const unsigned long long *starts, uint64_t n, uint8_t *out, sq_meta *metas)
{
    const uint64_t waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    for (uint64_t r = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; r < n; r += waves) {
        const BamFields f = bam_fields(bam, offsets[r]);
        uint8_t *o = out + starts[r];
        for (uint32_t i = lane; i < f.name_len; i += 64) o[i] = bam[f.name + i];
        o += f.name_len;
        /* decode_bam_sequence :1264-1290: "=ACMGRSVTWYHKDBN"[nibble], high nibble first */
        const unsigned long long lut_lo = 0x565352474D43413DULL, lut_hi = 0x4E42444B48595754ULL;
        for (uint32_t i = lane; i < f.l_seq; i += 64) {
            const uint32_t b = bam[f.seq + (i >> 1)];
            const uint32_t code = (i & 1) ? (b & 15u) : (b >> 4);
            o[i] = (uint8_t)(((code & 8u) ? lut_hi : lut_lo) >> (8 * (code & 7u)));
        }
        o += f.l_seq;
        /* :1642-1650: missing qualities (0xff) become phred 0 */
        const bool missing = f.l_seq && bam[f.qual] == 0xff;
        /* A quality of 95 .. 222 would become a byte >= 128: no phred character in the reference either
           (`q > PHRED_MAX`, :2073-2075), but the round-1 kernels index 136-entry tables with the raw byte.
           Such a byte is stored as 0x7F, the smallest character that is none too: same ValueError, same
           tables behind it (every invalid character counts in bin 11); only the character the
           message prints differs. */
        for (uint32_t i = lane; i < f.l_seq; i += 64) {
            const uint8_t v = (uint8_t)(bam[f.qual + i] + 33);
            o[i] = missing ? (uint8_t)33 : v >= 128 ? (uint8_t)0x7F : v;
        }
        o += f.l_seq;
        for (uint32_t i = lane; i < f.tags_len; i += 64) o[i] = bam[f.tags + i];
        if (lane == 0) {
            sq_meta m;
            m.record_start = starts[r];
            m.name_length = f.name_len;
            m.sequence_offset = f.name_len;
            m.sequence_length = f.l_seq;
            m.qualities_offset = f.name_len + f.l_seq;
            m.tags_offset = f.name_len + 2 * f.l_seq;
            m.tags_length = f.tags_len;
            m.accumulated_error_rate = 0.0;
            metas[r] = m;
        }
    }
}

} // namespace

/* decoded batch of the BAM records at bam + offsets[i] (from sq_bam_scan); host pointers */
SQ_EXPORT sq_batch *sq_batch_from_bam(sq_ctx *ctx, const uint8_t *bam, size_t len, const uint64_t *offsets, size_t n)
{
    sq_batch *b = new sq_batch();
    b->ctx = ctx;
    b->owns = b->slack = true;
    b->n = n;
    uint8_t *d_bam = nullptr;
    unsigned long long *d_off = nullptr, *d_sizes = nullptr, *d_starts = nullptr;
    void *d_temp = nullptr;
    auto fail = [&](const char *what) -> sq_batch * {
        if (what) sq_set_error("%s", what);
        (void)hipStreamSynchronize(ctx->stream);
        for (void *p : {(void *)d_bam, (void *)d_off, (void *)d_sizes, (void *)d_starts, d_temp})
            if (p) (void)hipFree(p);
        sq_batch_free(b);
        return nullptr;
    };
    if (hipMalloc((void **)&b->d_metas, (n ? n : 1) * sizeof(sq_meta)) != hipSuccess) return fail("out of device memory");
    if (n == 0) {
        if (hipMalloc((void **)&b->d_buf, 64) != hipSuccess) return fail("out of device memory");
        return b;
    }
    if (hipMalloc((void **)&d_bam, len + 64) != hipSuccess || hipMalloc((void **)&d_off, n * 8) != hipSuccess ||
        hipMalloc((void **)&d_sizes, (n + 1) * 8) != hipSuccess || hipMalloc((void **)&d_starts, (n + 1) * 8) != hipSuccess)
        return fail("out of device memory");
    if (hipMemcpyAsync(d_bam, bam, len, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(d_off, offsets, n * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        return fail("copy to the device failed");
    const int blocks = (int)std::min<uint64_t>((n + 255) / 256, 8192);
    hipLaunchKernelGGL(k_bam_sizes, dim3(blocks), dim3(256), 0, ctx->stream, d_bam, d_off, (uint64_t)n, d_sizes);
    size_t temp_bytes = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, temp_bytes, d_sizes, d_starts, (int)(n + 1), ctx->stream);
    if (hipMalloc(&d_temp, temp_bytes ? temp_bytes : 8) != hipSuccess) return fail("out of device memory");
    (void)hipcub::DeviceScan::ExclusiveSum(d_temp, temp_bytes, d_sizes, d_starts, (int)(n + 1), ctx->stream);
    (void)hipMemcpyAsync(&ctx->pinned[44], d_starts + n, 8, hipMemcpyDeviceToHost, ctx->stream);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return fail("BAM decode failed on the device");
    b->buf_len = (size_t)ctx->pinned[44];
    if (hipMalloc((void **)&b->d_buf, b->buf_len + 64) != hipSuccess) return fail("out of device memory");
    const int wblocks = (int)std::min<uint64_t>((n + 3) / 4, 16384);
    hipLaunchKernelGGL(k_bam_decode, dim3(wblocks), dim3(256), 0, ctx->stream, d_bam, d_off, d_starts, (uint64_t)n,
                       b->d_buf, b->d_metas);
    if (stats_issue(ctx, b->d_metas, (size_t)n) != SQ_OK) return fail("out of device memory");
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return fail("BAM decode failed on the device");
    stats_take(ctx, b);
    (void)hipFree(d_bam); (void)hipFree(d_off); (void)hipFree(d_sizes); (void)hipFree(d_starts); (void)hipFree(d_temp);
    return b;
}

SQ_EXPORT void sq_batch_free(sq_batch *b)
{
    if (!b) return;
    if (b->pooled && (b->owns || b->owns_metas)) {   /* the blocks go back behind what is queued on the stream, without waiting for it */
        sq_ctx::Deferred d{b->owns ? (void *)b->d_buf : nullptr, (void *)b->d_metas, nullptr};
        if (hipEventCreateWithFlags(&d.passed, hipEventDisableTiming) == hipSuccess) {
            if (hipEventRecord(d.passed, b->ctx->stream) == hipSuccess) {
                b->ctx->deferred.push_back(d);
                sq_dev_reclaim(b->ctx, false);
                if (b->ready) (void)hipEventDestroy(b->ready);
                delete b;
                return;
            }
            (void)hipEventDestroy(d.passed);
        }
        (void)hipGetLastError();
    }
    if (b->owns || b->owns_metas) (void)hipStreamSynchronize(b->ctx->stream);
    if (b->ready) (void)hipEventDestroy(b->ready);
    if (b->pooled) {
        if (b->owns && b->d_buf) sq_dev_put(b->ctx, b->d_buf);
        if ((b->owns || b->owns_metas) && b->d_metas) sq_dev_put(b->ctx, b->d_metas);
    } else {
        if (b->owns && b->d_buf) (void)hipFree(b->d_buf);
        if ((b->owns || b->owns_metas) && b->d_metas) (void)hipFree(b->d_metas);
    }
    delete b;
}

SQ_EXPORT uint64_t sq_batch_size(const sq_batch *b) { return b->n; }
SQ_EXPORT uint64_t sq_batch_total_bases(const sq_batch *b) { return b->total_bases; }
SQ_EXPORT int sq_batch_length_counts(const sq_batch *b, uint32_t *counts)
{
    if (b->len_hist.size() != SQ_LEN_BINS) return 0;
    memcpy(counts, b->len_hist.data(), SQ_LEN_BINS * sizeof(uint32_t));
    return 1;
}
SQ_EXPORT uint64_t sq_batch_max_length(const sq_batch *b) { return b->max_length; }

SQ_EXPORT uint64_t sq_batch_bytes(const sq_batch *b) { return b->buf_len; }
SQ_EXPORT void *sq_batch_device_text(const sq_batch *b) { return b->d_buf; }
SQ_EXPORT void *sq_batch_device_metas(const sq_batch *b) { return b->d_metas; }

SQ_EXPORT int sq_batch_download(sq_batch *b, uint8_t *buf, size_t buf_cap, sq_meta *metas, size_t meta_cap)
{
    if ((buf && buf_cap < b->buf_len) || meta_cap < b->n) {
        sq_set_error("sq_batch_download: destination too small");
        return SQ_ERR_VALUE;
    }
    SQ_HIP(hipStreamSynchronize(b->ctx->stream));
    if (buf && b->buf_len) SQ_HIP(hipMemcpy(buf, b->d_buf, b->buf_len, hipMemcpyDeviceToHost));
    if (b->n) SQ_HIP(hipMemcpy(metas, b->d_metas, b->n * sizeof(sq_meta), hipMemcpyDeviceToHost));
    return SQ_OK;
}

SQ_EXPORT int sq_batch_error_rates(sq_batch *b, double *out, size_t n)
{
    if (n > b->n) n = b->n;
    if (!n) return SQ_OK;
    /* strided 8 of every 40 bytes */
    SQ_HIP(hipMemcpy2DAsync(out, sizeof(double),
                            (const uint8_t *)b->d_metas + offsetof(sq_meta, accumulated_error_rate),
                            sizeof(sq_meta), sizeof(double), n, hipMemcpyDeviceToHost,
                            b->ctx->stream));
    SQ_HIP(hipStreamSynchronize(b->ctx->stream));
    return SQ_OK;
}

/* ---- synthetic FASTQ ------------------------------------------------------------ */

__host__ __device__ static inline void synth_write_record(int kind, uint64_t seed, uint64_t i,
                                                          uint8_t *dst, uint32_t lane,
                                                          uint32_t nlanes)
{
    /* lanes stride over the bytes of one record */
    const int mate = sqs_kind_mate(kind);
    const bool nanopore = sqs_base_kind(kind) == SQ_SYNTH_NANOPORE;
    uint32_t nl = sqs_name_length(kind);
    uint32_t L = sqs_read_length(kind, seed, i);
    uint64_t src = 0;
    uint32_t flen = 0;
    if (!nanopore) {
        src = sqs_source_pair(seed, i);
        flen = sqs_fragment_length(seed, src);
    }
    if (lane == 0) {
        dst[0] = '@';
        if (nanopore) sqs_nanopore_name(seed, i, dst + 1);
        else sqs_illumina_name(seed, i, mate, dst + 1, sqs_kind_by_tile(kind));
        dst[1 + nl] = '\n';
        dst[2 + nl + L] = '\n';
        dst[3 + nl + L] = '+';
        dst[4 + nl + L] = '\n';
        dst[5 + nl + 2 * (uint64_t)L] = '\n';
    }
    uint8_t *seq = dst + 2 + nl;
    uint8_t *qual = dst + 5 + nl + L;
    for (uint32_t p = lane; p < L; p += nlanes) {
        if (nanopore) {
            seq[p] = sqs_nanopore_base(seed, i, p);
            qual[p] = sqs_nanopore_qual(seed, i, p);
        } else {
            seq[p] = sqs_illumina_base(seed, i, src, flen, mate, p);
            qual[p] = sqs_illumina_qual(seed, i, mate, p, L);
        }
    }
}

__host__ __device__ static inline void synth_fill_meta(int kind, uint64_t seed, uint64_t i,
                                                       uint64_t rec_off, sq_meta *m)
{
    uint32_t nl = sqs_name_length(kind), L = sqs_read_length(kind, seed, i);
    m->record_start = rec_off + 1;
    m->name_length = nl;
    m->sequence_offset = nl + 1;
    m->sequence_length = L;
    m->qualities_offset = nl + 1 + L + 3;
    m->tags_offset = nl + 1 + L + 3 + L;
    m->tags_length = 0;
    m->accumulated_error_rate = 0.0;
}

SQ_EXPORT uint64_t sq_synth_bytes(int kind, uint64_t seed, uint64_t first, uint64_t n)
{
    if (sqs_base_kind(kind) != SQ_SYNTH_NANOPORE) return n * sqs_record_bytes(kind, seed, 0);
    uint64_t t = 0;
    for (uint64_t i = 0; i < n; i++) t += sqs_record_bytes(kind, seed, first + i);
    return t;
}

SQ_EXPORT int sq_synth_host(int kind, uint64_t seed, uint64_t first, uint64_t n, uint8_t *buf,
                            size_t buf_cap, sq_meta *metas)
{
    uint64_t off = 0;
    for (uint64_t k = 0; k < n; k++) {
        uint64_t sz = sqs_record_bytes(kind, seed, first + k);
        if (off + sz > buf_cap) {
            sq_set_error("sq_synth_host: buffer too small");
            return SQ_ERR_VALUE;
        }
        synth_write_record(kind, seed, first + k, buf + off, 0, 1);
        if (metas) synth_fill_meta(kind, seed, first + k, off, &metas[k]);
        off += sz;
    }
    return SQ_OK;
}

__global__ void k_synth_sizes(int kind, uint64_t seed, uint64_t first, uint64_t n, uint64_t *sizes)
{
    uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (k < n) sizes[k] = sqs_record_bytes(kind, seed, first + k);
}

/* one wave per record; offs[k] = byte offset of record k (NULL: fixed size) */
__global__ void k_synth_fill(int kind, uint64_t seed, uint64_t first, uint64_t n,
                             const uint64_t *offs, uint64_t fixed, uint8_t *buf, sq_meta *metas)
{
    uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    uint32_t lane = threadIdx.x & 63;
    for (uint64_t k = wave; k < n; k += nwaves) {
        uint64_t off = offs ? offs[k] : k * fixed;
        synth_write_record(kind, seed, first + k, buf + off, lane, 64);
        if (lane == 0) synth_fill_meta(kind, seed, first + k, off, &metas[k]);
    }
}

SQ_EXPORT sq_batch *sq_synth_device(sq_ctx *ctx, int kind, uint64_t seed, uint64_t first, uint64_t n)
{
    sq_batch *b = new sq_batch();
    b->ctx = ctx;
    b->n = n;
    b->owns = b->slack = true;
    uint64_t *d_offs = nullptr;
    uint64_t fixed = 0, total = 0;
    if (sqs_base_kind(kind) == SQ_SYNTH_NANOPORE) {
        /* sizes on the device, exclusive scan on the host (n is ~1e6) */
        std::vector<uint64_t> sizes(n);
        SQ_HIP_NULL(hipMalloc((void **)&d_offs, (n ? n : 1) * sizeof(uint64_t)));
        if (n) {
            hipLaunchKernelGGL(k_synth_sizes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                               ctx->stream, kind, seed, first, n, d_offs);
            SQ_HIP_NULL(hipMemcpyAsync(sizes.data(), d_offs, n * 8, hipMemcpyDeviceToHost, ctx->stream));
            SQ_HIP_NULL(hipStreamSynchronize(ctx->stream));
        }
        for (uint64_t k = 0; k < n; k++) {
            uint64_t s = sizes[k];
            sizes[k] = total;
            total += s;
        }
        if (n) SQ_HIP_NULL(hipMemcpyAsync(d_offs, sizes.data(), n * 8, hipMemcpyHostToDevice, ctx->stream));
        SQ_HIP_NULL(hipStreamSynchronize(ctx->stream));
    } else {
        fixed = sqs_record_bytes(kind, seed, 0);
        total = fixed * n;
    }
    b->buf_len = total;
    if (hipMalloc((void **)&b->d_buf, total + 64) != hipSuccess ||
        hipMalloc((void **)&b->d_metas, (n ? n : 1) * sizeof(sq_meta)) != hipSuccess) {
        sq_set_error("sq_synth_device: out of device memory (%llu bytes)", (unsigned long long)total);
        if (b->d_buf) (void)hipFree(b->d_buf);
        if (d_offs) (void)hipFree(d_offs);
        delete b;
        return nullptr;
    }
    SQ_HIP_NULL(hipMemsetAsync(b->d_buf + total, 0, 64, ctx->stream));
    if (n) {
        uint64_t waves = n < 65536 ? n : 65536;
        unsigned blocks = (unsigned)((waves + 3) / 4);
        hipLaunchKernelGGL(k_synth_fill, dim3(blocks), dim3(256), 0, ctx->stream, kind, seed, first,
                           n, d_offs, fixed, b->d_buf, b->d_metas);
        if (stats_issue(ctx, b->d_metas, n) != SQ_OK) return nullptr;
        SQ_HIP_NULL(hipStreamSynchronize(ctx->stream));
        stats_take(ctx, b);
    }
    if (d_offs) (void)hipFree(d_offs);
    return b;
}

namespace {
__global__ void k_synth_trim(sq_meta *metas, uint64_t n, uint64_t seed, uint32_t lo)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t L = metas[i].sequence_length;
        if (L <= lo) continue;
        uint64_t h = (seed ^ i) * 0x9E3779B97F4A7C15ULL;
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ULL; h ^= h >> 32;
        metas[i].sequence_length = lo + (uint32_t)(h % (L - lo + 1));
    }
}
} // namespace

SQ_EXPORT int sq_synth_trim(sq_batch *b, uint64_t seed, uint32_t lo)
{
    if (!b->n) return SQ_OK;
    sq_ctx *ctx = b->ctx;
    hipLaunchKernelGGL(k_synth_trim, dim3((unsigned)std::min<uint64_t>((b->n + 255) / 256, 4096)), dim3(256), 0, ctx->stream,
                       b->d_metas, (uint64_t)b->n, seed, lo);
    int rc = stats_issue(ctx, b->d_metas, b->n);
    if (rc != SQ_OK) return rc;
    SQ_HIP(hipStreamSynchronize(ctx->stream));
    stats_take(ctx, b);
    b->h_metas.clear(); /* the host copy, if any, no longer describes the batch */
    b->h_buf.clear();
    return SQ_OK;
}
