/* sq_span.h -- LDS layout of k_span (sq_span.hip), shared by the kernel and its launcher */
#ifndef SQ_SPAN_H
#define SQ_SPAN_H

#include <cstddef>
#include <cstdint>
#include <vector>

constexpr int SPAN_NW_MAX = 8;               /* reads of up to 256 bases */
constexpr uint32_t SPAN_DFA_ROW = 216;          /* bytes three states of the two-character automaton share: 36 entries each */
constexpr uint32_t SPAN_DFA_MAX_STATES = 336;   /* 24 KB of LDS */
constexpr int SPAN_NW_AD = 5;   /* windows of 32 positions with the automaton in the pass (one wave for both streams) */
constexpr int SPAN_NW_AD_SPLIT = 8;   /* the same with a wave per stream: the whole range (batches of ONE read length of 225-256 bases go to k_wide all the same: sq_span_launch) */
constexpr uint32_t SPAN_ERR_N = 264;      /* error rates by raw quality byte, NaN for what is no phred character ... */
constexpr uint32_t SPAN_ERR_PAD = 256;    /* ... and one entry that no byte reaches: +0.0, for chain steps that do not exist */
constexpr uint32_t SPAN_BIN_OFF = SPAN_ERR_N * 8;
constexpr uint32_t SPAN_META_BYTES = 16 * 40;   /* the metas of a span */
constexpr uint32_t SPAN_META_LDS = 16 * 32;     /* what k_span keeps of them: the first 32 bytes of each (SEG: 16 bytes per row) */

struct SpanLds {
    uint32_t thr, gc, ps, prog, dfa, out, adlen, hist, first, rows, dma, meta, ends, slots;
    size_t total;
};

/* nw: 32-position windows per read; U: read length; states / n_ad / ad_lds: the automaton, its
 * adapters, how many of them are counted in LDS (0 without AdapterCounter); waves per workgroup */
__host__ __device__ inline SpanLds span_lds_layout(int nw, uint32_t U, uint32_t states, uint32_t n_ad,
                                                   uint32_t ad_lds, int waves, bool seg = false, bool split = false, bool lng = false, bool ends = false)
{
    SpanLds L;
    const uint32_t hs = (U + 31u) & ~31u;
    uint32_t o = SPAN_ERR_N * 8;        /* error rates by quality byte, at LDS address 0 */
    o += 256 * 2;                        /* SPAN_BIN_OFF: phred histogram row by quality byte */
    L.thr = o; o += 96 * 8;
    L.gc = o; o += 104 * 4;
    L.ps = o; o += 96 * 4;
    L.prog = o; o += 64;                 /* spans done, per wave (a wave per stream: the two waves of a pair stay within a span of each other) */
    L.dfa = o; o += ((states + 2) / 3 * SPAN_DFA_ROW + 15u) & ~15u; /* three states to a row */
    L.out = o; o += states * 16;   /* adapters ending on the second / the first character of a step */
    L.adlen = o; o += states ? 64 : 0;
    L.hist = o; o += hs * (5 + 12 + (seg ? 1 : 0)) * 4 + ad_lds * hs * 4; /* seg: a 13th phred row takes the qualities of filler rows */
    L.first = o; o += (uint32_t)waves * 16 * n_ad * 4;
    L.rows = o; o += (uint32_t)waves * (seg ? 64 : 32) * 4;
    /* split: a wave holds one stream of a span at a time (sequence or qualities): rows of 2 nw + 1 pieces */
    const uint32_t pr = (split ? 2 : 4) * (uint32_t)nw + 1 + (lng ? 2 : 0);   /* long: a piece in front of the segment (and one to stay odd) */
    L.dma = o; o += ((16u * pr + 63) / 64) * 64 * 4;
    o = (o + 15u) & ~15u;
    L.meta = o; o += (uint32_t)waves * (seg ? 256 : SPAN_META_LDS);   /* one buffer: the metas of span k + 2 land where those of k + 1 were read */
    L.ends = o; o += ends ? (uint32_t)waves * 512 + 128 : 0;   /* k_span<PAIR = 2>: the ends of read 2 of the span being counted, and 16 more parameters */
    L.slots = o;
    L.total = (size_t)o + (size_t)waves * 2 * 16 * 16 * (size_t)pr; /* two slots of 16 rows of pr pieces */
    return L;
}

/* k_span over a batch of many read lengths: the records are sorted by length and cut into spans
 * of 16 reads of one length (the last span of a length is filled up with copies of its last
 * read, which the kernel turns into padding).  A launch takes the lengths of one window count. */
struct SpanSeg {           /* one read length */
    uint32_t span0;        /* its first span (of the launch) */
    uint32_t nspans;
    uint32_t U;            /* the length */
    uint32_t last_rows;    /* reads in its last span (1 .. 16) */
    uint32_t first;        /* its first read among the sorted rows */
    uint32_t pos_base;     /* k_span<LONG>: where the segment starts in its reads */
    uint32_t pad[2];
};
struct SpanRow {           /* one row of a span: 16 bytes */
    uint64_t seq;          /* offset of the sequence in the batch's buffer */
    uint32_t qual_delta;   /* qualities start that many bytes behind the sequence */
    uint32_t record;       /* index of the record in the batch */
};

/* k_isz_span: the overlap scan of InsertSizeMetrics (calculate_insert_size, _qcmodule.c:5667-5707)
 * on pairs of one read length each, read 1 streamed through LDS */
struct sq_meta;
struct IszSpanParams {
    const uint8_t *buf1, *buf2;
    const sq_meta *metas1, *metas2;
    uint64_t n;
    uint32_t L1, L2;                     /* the read lengths (both >= 16, L1 <= 256) */
    unsigned long long *insert_sizes;    /* [>= L1 + L2 + 17] */
    uint32_t lds_sizes;                  /* entries of it a workgroup counts in LDS first */
    unsigned long long *max_insert;
    uint32_t *results;                   /* [n]: the insert size where the pair leaves an adapter remainder, else 0 */
};

struct sq_ctx;
struct PassParams;
int sq_span_launch(sq_ctx *ctx, const PassParams &P, bool ad, uint32_t n_ad, uint64_t *done);
int sq_isz_span_launch(sq_ctx *ctx, const IszSpanParams &P, uint64_t *done);
int sq_ptspan_launch(sq_ctx *ctx, const PassParams &P, uint32_t nslots, uint64_t *done);
int sq_span_launch_sorted(sq_ctx *ctx, const PassParams &P, bool ad, uint32_t n_ad, uint32_t min_len, uint32_t max_len, const uint32_t *len_hist, uint64_t *done);
/* the pass over the segments of long reads and what follows it (the G/C bins, the end-anchored tables) */
int sq_span_launch_long(sq_ctx *ctx, const PassParams &P, bool ad, uint32_t n_ad, uint32_t max_len, uint64_t *done);
bool sq_span_long_takes(const PassParams &P, bool ad, uint32_t n_ad, uint32_t max_len);
/* sq_span_w6.hip: the builds of k_span for adapters of 14 .. 25 characters (the automaton restarted six dwords in front of a quarter) */
bool sq_span_w6_exists(int nw, bool seg, bool split);
bool sq_span_w6_spills(int nw, bool seg, bool split);
int sq_span_launch_w6(int nw, bool seg, bool split, sq_ctx *ctx, const PassParams &P, uint32_t n_ad, int waves, size_t lds, int grid);
/* sq_pair.hip: QCMetrics' pass with PerTileQuality riding along (k_span<PT>), and what folds its staged runs into the tables */
struct PtRun;
int sq_span_launch_pt(sq_ctx *ctx, const PassParams &P, int pair, uint64_t *done);   /* pair: 0, 1 (read 2: writes the ends), 2 (read 1: the overlap scan) */
int sq_pt_runs_assign(sq_ctx *ctx, PtRun *runs, uint32_t n_runs, long long *keys, int *vals, int *n_slots, int *overflow);
/* sq_ends.hip: InsertSizeMetrics behind an overlap scan that ran inside read 1's pass (sq_paired_add_batches, sq_qc.hip) */
struct sq_insertsize;
struct sq_batch;
int sq_insertsize_add_batch_pair_scanned(sq_insertsize *z, sq_batch *b1, sq_batch *b2, const uint32_t *scanned, uint64_t scanned_pairs);
uint32_t *sq_insertsize_scan_results(sq_insertsize *z, uint64_t n);
int sq_insertsize_reserve_for(sq_insertsize *z, sq_batch *b1, sq_batch *b2);
int sq_pt_fold(sq_ctx *ctx, const PtRun *runs, const double *sums, uint32_t n_runs, uint32_t U, double *errors, unsigned long long *len_counts, uint64_t cap);

#endif
