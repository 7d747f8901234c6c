/*
 * sqgpu.h -- C ABI of libsqgpu.so: sequali's per-read QC accumulators on
 * MI355X (gfx950).  Plain pointers and sizes only; no torch / HIP types.
 *
 * Every entry point replaces a function of the reference's `_qc` extension
 * (paths relative to src/sequali/ of rhpvorderman/sequali v1.0.2).  The
 * reference hands each module a FastqRecordArrayView = one bytes buffer + an
 * array of 40-byte FastqMeta structs (_qcmodule.c:337-355, 575-579); here the
 * same pair crosses the boundary as (buf, buf_len, sq_meta*, n) with the
 * record_start pointer turned into a byte offset into buf.
 *
 * Conventions (the reference's C convention, _qcmodule.c:2183-2204 etc.):
 * functions returning int give 0 on success and a negative SQ_ERR_* code on
 * failure; sq_last_error() then holds the message the reference would have
 * put into the Python exception.  Work is enqueued on the context's HIP
 * stream and is asynchronous: a deferred error (e.g. an invalid phred byte)
 * surfaces at the module's *_flush() or at its first getter, which flush.
 * Objects are not thread-safe (neither are the reference's).
 */
#ifndef SQGPU_H
#define SQGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SQ_ABI_VERSION 1

#define SQ_OK 0
#define SQ_ERR_HIP (-1)           /* a HIP runtime call failed (RuntimeError)            */
#define SQ_ERR_VALUE (-2)         /* ValueError, e.g. "Not a valid phred character: %c"   */
#define SQ_ERR_MEMORY (-3)        /* MemoryError                                          */
#define SQ_ERR_TYPE (-4)          /* TypeError                                            */
#define SQ_ERR_EOF (-5)           /* EOFError: incomplete record at the end of the file   */
#define SQ_ERR_OVERFLOW (-6)      /* OverflowError                                        */
#define SQ_ERR_RUNTIME (-7)       /* RuntimeError raised by the reference itself          */
#define SQ_ERR_SYSTEM (-8)        /* the reference returns NULL without an exception set  */

#define SQ_NUMBER_OF_NUCS 5       /* NUC_TABLE_SIZE   _qcmodule.c:1766 */
#define SQ_NUMBER_OF_PHREDS 12    /* PHRED_TABLE_SIZE _qcmodule.c:1768 */
#define SQ_PHRED_MAX 93           /* _qcmodule.c:101  */
#define SQ_MAX_SEQUENCE_SIZE 64   /* _qcmodule.c:2393 */
#define SQ_ADAPTER_STORE_SIZE 31  /* INSERT_SIZE_MAX_ADAPTER_STORE_SIZE _qcmodule.c:5457 */

/* struct FastqMeta, _qcmodule.c:337-355, same size and field order; the
 * pointer is an offset so that the array is position independent. */
typedef struct sq_meta {
    uint64_t record_start;        /* offset of the first name byte in buf */
    uint32_t name_length;
    uint32_t sequence_offset;     /* relative to record_start */
    uint32_t sequence_length;
    uint32_t qualities_offset;    /* relative to record_start */
    uint32_t tags_offset;
    uint32_t tags_length;
    double accumulated_error_rate; /* written by QCMetrics (_qcmodule.c:2126) */
} sq_meta;

typedef struct sq_ctx sq_ctx;
typedef struct sq_batch sq_batch;
typedef struct sq_qcmetrics sq_qcmetrics;
typedef struct sq_adaptercounter sq_adaptercounter;
typedef struct sq_pertile sq_pertile;
typedef struct sq_overrep sq_overrep;
typedef struct sq_dedup sq_dedup;
typedef struct sq_insertsize sq_insertsize;
typedef struct sq_feeder sq_feeder;

/* ---- library / context ------------------------------------------------ */
int sq_abi_version(void);
const char *sq_last_error(void);
size_t sq_last_error_length(void);   /* the message may hold a zero byte ("... but with %c" of the input's byte, _qcmodule.c:1076) */
/* One context per process and GPU: selects `device`, creates the stream all
 * work of its modules is ordered on.  NULL on failure. */
sq_ctx *sq_init(int device);
void sq_shutdown(sq_ctx *ctx);
int sq_synchronize(sq_ctx *ctx);
/* hipStream_t of the context, for event timing / interop by the caller. */
void *sq_stream_handle(sq_ctx *ctx);
/* The library reads its SQ_* route / experiment switches from the environment once, at first
 * use; call this after changing one (the tests do).  No reference counterpart: sequali has no
 * kernel routes to choose from. */
void sq_knobs_reload(void);

/* ---- record boundary, host side (no GPU involved) ---------------------- */
/* The record loop of FastqParser_create_record_array, _qcmodule.c:1093-1171:
 * split FASTQ text into records.  Writes up to `cap` metas (offsets relative
 * to buf), stores the number of bytes consumed by complete records in
 * *consumed.  Returns the record count or a negative SQ_ERR_* (bad '@'/'+',
 * unequal sequence/quality length). */
int64_t sq_fastq_split(const uint8_t *buf, size_t len, sq_meta *metas, size_t cap,
                       size_t *consumed);
/* string_is_ascii, _qcmodule.c:203-237: index of the first byte >= 0x80 or -1 */
int64_t sq_first_non_ascii(const uint8_t *buf, size_t len);
/* FastqRecordArrayView_is_mate, _qcmodule.c:814-850 (fastq_names_are_mates
 * :777-800): 1 if every pair of names matches, else 0. */
int sq_names_are_mates(const uint8_t *buf1, const sq_meta *metas1, const uint8_t *buf2,
                       const sq_meta *metas2, size_t n);

/* ---- FastqParser's buffer logic over pinned staging blocks (host side) ---- */
/* FastqParser (_qcmodule.c:889-1244) hands out record arrays that are windows of the file of
 * `initial_buffersize` bytes.  sq_feeder keeps that chunking (same arrays, same errors) while the
 * file's text is read once into pinned blocks of up to `block_bytes` (0: 64 MiB) that go to HBM
 * with one asynchronous copy each: see csrc/sq_feed.hip.  ctx may be NULL (host parsing only). */
#define SQ_FEED_MORE 1
typedef struct sq_feed_array {
    uint64_t block_id;      /* the staging block the array lies in */
    uint64_t byte_start;    /* its window of the block: what the reference's buffer object holds */
    uint64_t byte_len;
    uint64_t first_record;  /* its records among the block's */
    uint64_t n_records;     /* 0: the file is exhausted (FastqParser__next__ :1201 stops) */
} sq_feed_array;
/* FastqParser__new__ :905-945 */
sq_feeder *sq_feeder_new(sq_ctx *ctx, size_t read_in_size, size_t block_bytes);
void sq_feeder_free(sq_feeder *f);
/* where the caller's file.readinto() puts the next bytes, and how many fit (:1022-1030) */
/* A source the feeder reads by itself instead of being fed (to be set before the first sq_feeder_next, which then never
 * answers SQ_FEED_MORE): worker threads copy it into the staging blocks and note the line ends on their way, so that neither
 * the read (readinto of FastqParser_create_record_array, :1040-1051) nor the newline search (the four memchr of a record,
 * :1101-1139) run on the caller's thread.  `text` must stay valid and unchanged until sq_feeder_free; `fd` is a regular file,
 * read with pread. */
int sq_feeder_set_source_memory(sq_feeder *f, const uint8_t *text, size_t len);
int sq_feeder_set_source_fd(sq_feeder *f, int fd, uint64_t offset, uint64_t len);
uint8_t *sq_feeder_fill(sq_feeder *f, size_t *room);
int sq_feeder_filled(sq_feeder *f, size_t n);
/* FastqParser_create_record_array :964-1184 (min 1, max SIZE_MAX: __next__; n, n: read(n)).
 * SQ_FEED_MORE: fill and call again; SQ_OK: *out is the array; < 0: the reference's exception. */
int sq_feeder_next(sq_feeder *f, size_t min_records, size_t max_records, sq_feed_array *out);
/* closes the open block behind its last array, so that it can be uploaded */
int sq_feeder_seal(sq_feeder *f);
const uint8_t *sq_feeder_block_text(sq_feeder *f, uint64_t block_id);
const sq_meta *sq_feeder_block_metas(sq_feeder *f, uint64_t block_id);
uint64_t sq_feeder_block_records(sq_feeder *f, uint64_t block_id);
/* bytes of the block that hold text: every array window of the block lies inside [0, bytes) */
uint64_t sq_feeder_block_bytes(sq_feeder *f, uint64_t block_id);
int sq_feeder_block_is_open(sq_feeder *f, uint64_t block_id);
/* a sealed block as a record array in HBM (one async copy from pinned memory); NULL on failure */
sq_batch *sq_feeder_upload(sq_feeder *f, uint64_t block_id);
/* a feeder that reads its source by itself: the blocks will be asked for in HBM -- from the open one on they go up while they
 * fill (without the call the first sq_feeder_upload says so; a parser that is only iterated uploads nothing) */
int sq_feeder_expect_uploads(sq_feeder *f);
/* the host copy of a sealed block is no longer needed */
void sq_feeder_release(sq_feeder *f, uint64_t block_id);

/* Test hook (host only): how k_span over the segments of long reads shares its spans among `grid` workgroups -- bounds
 * [grid + 1] = first span of every workgroup's stretch of consecutive spans -- when a span costs 1 and every segment a
 * workgroup meets `cost` (nspans[j] = spans of segment j). */
void sq_span_cost_shares(const uint32_t *nspans, size_t n, int grid, int cost, uint32_t *bounds);

/* Diagnostics: the counting kernels the dispatchers have launched on this context since sq_route_reset(), joined by
 * '+' ("k_span<5,AD,split>+k_span_scatter+k_span<4,AD,sorted>+..."): which kernel takes a batch is decided per batch
 * (DESIGN.md 4.1), and a build that no longer fits its registers makes a dispatcher take another one silently;
 * tests/test_gpu_routes.py pins the routes of the benchmark's shapes. */
const char *sq_last_route(sq_ctx *ctx);
void sq_route_reset(sq_ctx *ctx);

/* Diagnostics: seconds the process's feeders have spent moving to a new block, in the record split and in fresh
 * allocations of the page-locked pool, and the number of those allocations (out[0..3]); reset != 0: start again. */
void sq_feeder_debug_times(double *out, int reset);
/* ... of feeders that read their source by themselves: seconds the caller's thread waited for text and for the walker,
 * seconds the walker and (summed) the workers were at work (out[0..3]). */
void sq_feeder_debug_waits(double *out, int reset);
/* Diagnostics: the context's pool of device blocks (texts and metas of record arrays): hipMalloc calls, hipFree calls,
 * blocks idle in the pool, blocks of freed arrays still waiting for the kernels queued when they were freed (out[0..3]). */
void sq_pool_counts(sq_ctx *ctx, uint64_t *out);

/* page-locked host memory (uploads from it run at the bus rate); plain memory without a device */
void *sq_host_alloc(size_t bytes, int *pinned);
void sq_host_free(void *p, int pinned);

/* ---- batches: a record array resident in HBM --------------------------- */
/* Copies buf and metas to the device (the reference's modules borrow the
 * array for the duration of the call, _qcmodule.c:575-607; a deferred GPU
 * pass must own a copy). */
sq_batch *sq_batch_upload(sq_ctx *ctx, const uint8_t *buf, size_t buf_len,
                          const sq_meta *metas, size_t n);
/* Wraps memory that already lives on the device (borrowed, not freed). */
sq_batch *sq_batch_wrap_device(sq_ctx *ctx, const void *d_buf, size_t buf_len, void *d_metas,
                               size_t n);
/* Records [first, first + n) of a batch as a batch of their own (borrowed: `parent` must outlive it).  Unlike
 * sq_batch_wrap_device over the same pointers it keeps what the library knows about the parent's memory (the spare
 * bytes behind the text), so the view takes the same kernels as the parent.  The reference's counterpart is a
 * FastqRecordArrayView made from a slice of records of one buffer (_qcmodule.c:690-719). */
sq_batch *sq_batch_view(sq_batch *parent, size_t first, size_t n);
/* FastqParser's record split on the GPU (FastqParser_create_record_array, the loop at
 * _qcmodule.c:1093-1171, and the ASCII check :203-237, :1055-1067): finds every
 * complete 4-line record of `text`, builds the metas in HBM and returns the batch.
 * *consumed = bytes covered by complete records (the tail is the caller's leftover).
 * NULL + sq_last_error() for the reference's ValueErrors ("Record does not start
 * with @ ...", "... second header does not start with + ...", unequal sequence /
 * quality lengths, non-ASCII byte).  The text is copied to the device (host variant)
 * or borrowed (device variant). */
sq_batch *sq_batch_from_fastq(sq_ctx *ctx, const uint8_t *text, size_t len, size_t *consumed);
sq_batch *sq_batch_from_fastq_device(sq_ctx *ctx, const void *d_text, size_t len, size_t *consumed);
/* sq_batch_from_fastq for a caller who knows the bytes of its NEXT buffer already (text that lies in
 * page-locked memory as a whole: `ahead` must stay valid and unchanged until a later call has covered it):
 * their upload is started behind this buffer's and runs while this buffer is split and counted; a later
 * call whose text contains `ahead` finds them in HBM and uploads only what lies in front of them (the
 * leftover of the buffer before) and behind them; bytes sent ahead that the next call does not cover are
 * dropped. */
sq_batch *sq_batch_from_fastq_ahead(sq_ctx *ctx, const uint8_t *text, size_t len, size_t *consumed,
                                    const uint8_t *ahead, size_t ahead_len);
/* forgets what was sent ahead (the parser has ended, failed or been rewound: the next buffer will not be the one
 * it named; a later text at the same host address must not meet the stale copy) */
void sq_ahead_drop(sq_ctx *ctx);
/* BAM input.  sq_bam_scan is the record walk of BamParser__next__
 * (_qcmodule.c:1601-1681) on the host: offsets of the complete records of an uncompressed
 * BAM record stream that are not secondary / supplementary (:1262,1611), the bytes they
 * cover and how many were skipped; with offsets == NULL it only counts.
 * sq_batch_from_bam decodes those records on the GPU into name | sequence | qualities |
 * tags (4-bit bases -> ASCII :1264-1290, qualities + 33 :1350-1358, 0xff -> '!' :1642-1650,
 * name without its NUL :1633). */
int64_t sq_bam_scan(const uint8_t *bam, size_t len, uint64_t *offsets, size_t cap, size_t *consumed,
                    uint64_t *skipped);
sq_batch *sq_batch_from_bam(sq_ctx *ctx, const uint8_t *bam, size_t len, const uint64_t *offsets, size_t n);
void sq_batch_free(sq_batch *b);
uint64_t sq_batch_size(const sq_batch *b);
uint64_t sq_batch_total_bases(const sq_batch *b);
uint64_t sq_batch_max_length(const sq_batch *b);
/* counts[L] = records of L bases for L < 256, counts[256] = records of 256 and more (257 entries): counted when
 * the batch is made (with its number of bases and longest read); k_span over a batch of many read lengths puts
 * its rows in order with them instead of sorting.  Returns 0 when the batch has none. */
int sq_batch_length_counts(const sq_batch *b, uint32_t *counts);
uint64_t sq_batch_bytes(const sq_batch *b);
/* Device addresses of the batch's text and of its 40-byte metas (interop: torch, RCCL). */
void *sq_batch_device_text(const sq_batch *b);
void *sq_batch_device_metas(const sq_batch *b);
/* Copies the batch back to the host: buf_len bytes (skipped when buf is NULL) and n metas. */
int sq_batch_download(sq_batch *b, uint8_t *buf, size_t buf_cap, sq_meta *metas, size_t meta_cap);
/* metas[i].accumulated_error_rate of every record, after QCMetrics ran. */
int sq_batch_error_rates(sq_batch *b, double *out, size_t n);

/* ---- QCMetrics, _qcmodule.c:1786-2385 ---------------------------------- */
sq_qcmetrics *sq_qcmetrics_new(sq_ctx *ctx, uint64_t end_anchor_length); /* QCMetrics__new__ :1821 */
void sq_qcmetrics_free(sq_qcmetrics *m);
/* QCMetrics_add_record_array :2183, from host memory; also writes
 * accumulated_error_rate into metas (synchronises). */
int sq_qcmetrics_add(sq_qcmetrics *m, const uint8_t *buf, size_t buf_len, sq_meta *metas, size_t n);
int sq_qcmetrics_add_batch(sq_qcmetrics *m, sq_batch *b);
int sq_qcmetrics_flush(sq_qcmetrics *m);
/* Without waiting: the first call arms a poll (0); later calls: 0 = the passes queued when it was armed still run, 1 = they
 * have ended and no invalid phred character is flagged (the batches handed in before the arming call are counted for good:
 * the caller may free them; disarmed), -1 = a character is flagged (sq_qcmetrics_flush sees to it; disarmed). */
int sq_qcmetrics_poll(sq_qcmetrics *m);
/* Behind a flush that returned SQ_ERR_VALUE (an invalid phred character, _qcmodule.c:2102-2105:
 * the passes run whole batches and only flag the read).  A batch may hold the records of several
 * calls (small arrays are staged together); for every call's stretch [start, end) of it:
 * sq_batch_first_invalid_phred gives the index of the first offending record (-1: none),
 * sq_qcmetrics_uncount_tail(first, end, kept_max_length) takes back what the pass counted for
 * the records behind it up to `end` and for the part of that record the reference never
 * reached, so that tables, number_of_reads and max_length (kept_max_length: the longest read of
 * the batch that stays counted) are the reference's behind its ValueError.  The flag is rearmed
 * by the failing flush: the object stays usable. */
int64_t sq_batch_first_invalid_phred(sq_batch *b, uint64_t start, uint64_t end);
int sq_qcmetrics_uncount_tail(sq_qcmetrics *m, sq_batch *b, uint64_t first, uint64_t end, uint64_t kept_max_length);
uint64_t sq_qcmetrics_number_of_reads(sq_qcmetrics *m);    /* members :2361-2369 */
uint64_t sq_qcmetrics_max_length(sq_qcmetrics *m);
uint64_t sq_qcmetrics_end_anchor_length(sq_qcmetrics *m);
/* getters :2215-2334; each returns the element count (or <0) and fills `out`
 * when cap is large enough: max_length*5, max_length*12, end_anchor*5,
 * end_anchor*12, 101, 94 */
int64_t sq_qcmetrics_base_count_table(sq_qcmetrics *m, uint64_t *out, size_t cap);
int64_t sq_qcmetrics_phred_count_table(sq_qcmetrics *m, uint64_t *out, size_t cap);
int64_t sq_qcmetrics_end_anchored_base_count_table(sq_qcmetrics *m, uint64_t *out, size_t cap);
int64_t sq_qcmetrics_end_anchored_phred_count_table(sq_qcmetrics *m, uint64_t *out, size_t cap);
int64_t sq_qcmetrics_gc_content(sq_qcmetrics *m, uint64_t *out, size_t cap);
int64_t sq_qcmetrics_phred_scores(sq_qcmetrics *m, uint64_t *out, size_t cap);

/* ---- AdapterCounter, _qcmodule.c:2391-2969 ------------------------------ */
/* Test hook (host only): the automaton tables the kernels walk in place of the shift-AND words of
 * populate_bitmask (_qcmodule.c:2451-2609) for the first <= 64 adapters: one character per step
 * (dfa [states][8]: next << 4 | reports; out [states]: adapters ending there; states >=
 * *accept_first report) and two characters per step (dfa2 [states2][36] indexed first class +
 * 6 * second class, classes A C G T other padding; out2 [states2][2]: adapters ending on the
 * second / on the first character).  Returns the number of states, -1 if a capacity is too small. */
int64_t sq_adapter_automaton_tables(const char *const *adapters, const size_t *lengths, size_t n,
                                    uint16_t *dfa, uint64_t *out, size_t cap, uint32_t *accept_first,
                                    uint16_t *dfa2, uint64_t *out2, size_t cap2, uint32_t *states2);

/* Test hook (host only): thresholds[k] = the largest average error rate whose phred_scores bin
 * (floor(-10 log10(avg)), _qcmodule.c:2127-2136, host libm) is >= k, and what k_span compares the SUM of a
 * read's error rates with instead of dividing it by the length (1..256): sums[k] = the largest double S with
 * S / length <= thresholds[k] in IEEE double division.  94 entries each. */
void sq_phred_sum_thresholds(uint32_t length, double *thresholds, double *sums);

/* AdapterCounter__new__ :2464: n ASCII adapters, each at most 64 bytes. */
sq_adaptercounter *sq_adaptercounter_new(sq_ctx *ctx, const char *const *adapters,
                                         const size_t *lengths, size_t n);
void sq_adaptercounter_free(sq_adaptercounter *a);
int sq_adaptercounter_add(sq_adaptercounter *a, const uint8_t *buf, size_t buf_len,
                          const sq_meta *metas, size_t n);          /* :2867 */
int sq_adaptercounter_add_batch(sq_adaptercounter *a, sq_batch *b);
int sq_adaptercounter_flush(sq_adaptercounter *a);
uint64_t sq_adaptercounter_number_of_sequences(sq_adaptercounter *a);
uint64_t sq_adaptercounter_max_length(sq_adaptercounter *a);
uint64_t sq_adaptercounter_number_of_adapters(sq_adaptercounter *a);
/* get_counts :2902: forward and reverse arrays of adapter i, max_length each */
int64_t sq_adaptercounter_get_counts(sq_adaptercounter *a, size_t i, uint64_t *forward,
                                     uint64_t *reverse, size_t cap);

/* ---- PerTileQuality, _qcmodule.c:2975-3397 ------------------------------ */
sq_pertile *sq_pertile_new(sq_ctx *ctx);
void sq_pertile_free(sq_pertile *p);
int sq_pertile_add(sq_pertile *p, const uint8_t *buf, size_t buf_len, const sq_meta *metas, size_t n); /* :3268 */
int sq_pertile_add_batch(sq_pertile *p, sq_batch *b);
int sq_pertile_flush(sq_pertile *p);
uint64_t sq_pertile_number_of_reads(sq_pertile *p);
uint64_t sq_pertile_max_length(sq_pertile *p);
/* skipped_reason :3377: NULL while the module is active, else
 * "Can not parse header: <repr of the header>" (:3143) */
const char *sq_pertile_skipped_reason(sq_pertile *p);
uint64_t sq_pertile_number_of_tiles(sq_pertile *p);
/* get_tile_counts :3307: ascending tile ids; errors and counts are
 * [tiles][max_length], counts already reverse-cumulated */
int64_t sq_pertile_get_tile_counts(sq_pertile *p, int64_t *tile_ids, double *errors,
                                   uint64_t *counts, size_t cap_tiles, size_t cap_len);

/* ---- one fused pass over a batch for the three per-base modules --------- */
/* Any of the three may be NULL.  Same result as calling the three
 * *_add_batch functions one after the other, with one read of the batch. */
int sq_fused_add_batch(sq_batch *b, sq_qcmetrics *m, sq_adaptercounter *a, sq_pertile *p);

/* ---- OverrepresentedSequences, _qcmodule.c:3435-4236 --------------------- */
sq_overrep *sq_overrep_new(sq_ctx *ctx, int64_t max_unique_fragments, int64_t fragment_length,
                           int64_t sample_every, int64_t bases_from_start,
                           int64_t bases_from_end);                  /* :3464 */
void sq_overrep_free(sq_overrep *o);
int sq_overrep_add(sq_overrep *o, const uint8_t *buf, size_t buf_len, const sq_meta *metas, size_t n); /* :3987 */
int sq_overrep_add_batch(sq_overrep *o, sq_batch *b);
int sq_overrep_flush(sq_overrep *o);
uint64_t sq_overrep_number_of_sequences(sq_overrep *o);          /* members :4199-4220 */
uint64_t sq_overrep_sampled_sequences(sq_overrep *o);
uint64_t sq_overrep_collected_unique_fragments(sq_overrep *o);
uint64_t sq_overrep_total_fragments(sq_overrep *o);
/* Number of sampled reads that held a byte outside ACGTN (the reference
 * raises one UserWarning per such read, :3931) and the index (counted over
 * every record ever added) of the most recent one, or -1. */
uint64_t sq_overrep_warning_count(sq_overrep *o);
int64_t sq_overrep_last_warning_record(sq_overrep *o);
/* sequence_counts :4020: k-mers (2 bits per base, first base most
 * significant; already un-hashed) and their counts, unordered. */
int64_t sq_overrep_get_counts(sq_overrep *o, uint64_t *kmers, uint64_t *counts, size_t cap);

/* ---- DedupEstimator, _qcmodule.c:4270-4802 -------------------------------- */
sq_dedup *sq_dedup_new(sq_ctx *ctx, int64_t max_stored_fingerprints, int64_t front_sequence_length,
                       int64_t back_sequence_length, int64_t front_sequence_offset,
                       int64_t back_sequence_offset);                /* :4301 */
void sq_dedup_free(sq_dedup *d);
int sq_dedup_add(sq_dedup *d, const uint8_t *buf, size_t buf_len, const sq_meta *metas, size_t n); /* :4531 */
int sq_dedup_add_batch(sq_dedup *d, sq_batch *b);
int sq_dedup_add_pair(sq_dedup *d, const uint8_t *buf1, size_t len1, const sq_meta *metas1,
                      const uint8_t *buf2, size_t len2, const sq_meta *metas2, size_t n); /* :4572 */
int sq_dedup_add_batch_pair(sq_dedup *d, sq_batch *b1, sq_batch *b2);
int sq_dedup_flush(sq_dedup *d);
uint64_t sq_dedup_modulo_bits(sq_dedup *d);                     /* members :4770-4786 */
uint64_t sq_dedup_hash_table_size(sq_dedup *d);
uint64_t sq_dedup_tracked_sequences(sq_dedup *d);
/* duplication_counts :4721: the non-zero counts in slot order */
int64_t sq_dedup_duplication_counts(sq_dedup *d, uint64_t *out, size_t cap);
/* The table (EstimatorEntry[], :4283-4299) lives in HBM: lookups, inserts (DedupEstimator_add_fingerprint :4426-4460)
 * and rebuilds (DedupEstimator_increment_modulo :4383-4423) run as parallel steps over pieces of the stream of hashes
 * and leave the reference's table, slot for slot.  A piece the steps must not take (see csrc/sq_ends.hip) goes through
 * the reference's loop on a host copy; these two count the pieces either way (tests, bench.py). */
uint64_t sq_dedup_device_pieces(sq_dedup *d);
uint64_t sq_dedup_host_pieces(sq_dedup *d);

/* ---- InsertSizeMetrics, _qcmodule.c:5456-5982 ------------------------------ */
sq_insertsize *sq_insertsize_new(sq_ctx *ctx, int64_t max_adapters);  /* :5505 */
void sq_insertsize_free(sq_insertsize *z);
int sq_insertsize_add_pair(sq_insertsize *z, const uint8_t *buf1, size_t len1, const sq_meta *metas1,
                           const uint8_t *buf2, size_t len2, const sq_meta *metas2, size_t n); /* :5808 */
int sq_insertsize_add_batch_pair(sq_insertsize *z, sq_batch *b1, sq_batch *b2);
/* What the reference's driver does with a pair of arrays (__main__.py:279-306) in one call: QCMetrics_add_record_array +
 * PerTileQuality_add_record_array on read 1 and on read 2 (_qcmodule.c:2141-2165, :3224-3248), InsertSizeMetrics_
 * add_record_array_pair on both (:5827-5872) -- the same results as those five calls in that order.  Batches of one
 * read length each take two passes over the records instead of seven (csrc/sq_pair.hip; the default, SQ_PT_FUSED=0: the
 * five calls).  Any module may be NULL.  After a non-zero return the modules hold some prefix of the work (read 2's pass
 * runs first): partial, as the reference's state is behind an error in the middle of an array. */
int sq_paired_add_batches(sq_batch *b1, sq_batch *b2, sq_qcmetrics *m1, sq_pertile *p1, sq_qcmetrics *m2,
                          sq_pertile *p2, sq_insertsize *z);
int sq_insertsize_flush(sq_insertsize *z);
uint64_t sq_insertsize_total_reads(sq_insertsize *z);           /* members :5492-5502 */
uint64_t sq_insertsize_number_of_adapters_read1(sq_insertsize *z);
uint64_t sq_insertsize_number_of_adapters_read2(sq_insertsize *z);
int64_t sq_insertsize_insert_sizes(sq_insertsize *z, uint64_t *out, size_t cap);  /* :5876 */
/* adapters_read1/2 :5923-5944, slot order: bytes is [n][31] zero padded */
int64_t sq_insertsize_adapters(sq_insertsize *z, int read2, uint8_t *bytes, uint8_t *lengths,
                               uint64_t *counts, size_t cap);

/* ---- NanoStats, _qcmodule.c:4804-5430 ------------------------- */
/* struct NanoInfo :4808-4815 */
typedef struct sq_nanoinfo {
    int64_t start_time;           /* unix UTC seconds */
    float duration;
    int32_t channel_id;
    uint32_t length;
    uint32_t pad_;
    double cumulative_error_rate; /* FastqMeta.accumulated_error_rate of the read (:5314) */
    uint64_t parent_id_hash;
} sq_nanoinfo;
typedef struct sq_nanostats sq_nanostats;
sq_nanostats *sq_nanostats_new(sq_ctx *ctx);
void sq_nanostats_free(sq_nanostats *s);
/* add_record_array :5357; SQ_ERR_VALUE / SQ_ERR_RUNTIME / SQ_ERR_SYSTEM with the reference's
 * message when a record's tags are malformed (records in front of it stay counted); a header
 * without ch= / start_time= stops the module for good without an error (:5302-5312).
 * Run it after the QCMetrics pass of the same batch: it reads accumulated_error_rate. */
int sq_nanostats_add(sq_nanostats *s, const uint8_t *buf, size_t buf_len, const sq_meta *metas, size_t n);
int sq_nanostats_add_batch(sq_nanostats *s, sq_batch *b);
uint64_t sq_nanostats_number_of_reads(sq_nanostats *s);     /* members :5416-5428 */
int64_t sq_nanostats_minimum_time(sq_nanostats *s);
int64_t sq_nanostats_maximum_time(sq_nanostats *s);
const char *sq_nanostats_skipped_reason(sq_nanostats *s);
/* nano_info_iterator :5403: the NanoInfo of every counted read, in read order */
int64_t sq_nanostats_infos(sq_nanostats *s, sq_nanoinfo *out, size_t cap);
/* the "pi tag should have a valid uuid4 format" warnings (:5247) of the last add: the
 * counted length of each */
int64_t sq_nanostats_last_warnings(sq_nanostats *s, uint64_t *lengths, size_t cap);

/* ---- multi-GPU: the order-dependent modules across shards ------ */
/* One rank owns a contiguous range of the job's records.  What the reference makes
 * depend on the order of the reads is decided exactly as in one sequential run:
 *
 * OverrepresentedSequences (first-come cap, :3553): in shard mode the table is uncapped
 * and every key keeps the rank (sampled read of the job, staging slot) of its first
 * occurrence; sampling follows the job-wide record index (:3833).  Merge = all-gather
 * the shards' candidates -> select -> per-shard lookup -> sum -> install. */
int sq_overrep_set_shard(sq_overrep *o, uint64_t first_record_index);
/* first min(distinct, max_unique_fragments) keys by rank; DEVICE arrays; returns the
 * number (only the number when d_hashes is NULL or cap is too small) */
int64_t sq_overrep_shard_candidates(sq_overrep *o, uint64_t *d_hashes, uint64_t *d_ranks, size_t cap);
/* first max_unique_fragments distinct hashes of the concatenated candidates by rank */
int64_t sq_overrep_shard_select(sq_overrep *o, const uint64_t *d_hashes, const uint64_t *d_ranks,
                                size_t n, uint64_t *d_selected, size_t cap);
int sq_overrep_shard_lookup(sq_overrep *o, const uint64_t *d_hashes, size_t n, uint64_t *d_counts);
/* totals = {number_of_sequences, sampled_sequences, total_fragments, warning_count,
 * last_warning_record}; afterwards the object holds the job's state, not a shard's */
int sq_overrep_shard_install(sq_overrep *o, const uint64_t *d_hashes, const uint64_t *d_counts, size_t n,
                             const uint64_t *totals);

/* DedupEstimator (insertion order decides the modulo bits, :4430-4459): deferred mode
 * only hashes (the hashes stay in HBM); the insertion tail runs shard after shard:
 * import the state of the shard in front, resolve, export. */
int sq_dedup_set_deferred(sq_dedup *d, int on);
uint64_t sq_dedup_pending(sq_dedup *d);
int sq_dedup_resolve(sq_dedup *d);
uint64_t sq_dedup_state_bytes(sq_dedup *d);
int sq_dedup_export_state(sq_dedup *d, void *out, size_t cap);
int sq_dedup_import_state(sq_dedup *d, const void *in, size_t len);
/* The same estimator by gathering instead of relaying (sq_ends.hip "by gathering" has the argument): every shard
 * settles (the hashes of short pairs at its start finished with the store of the shards in front; lower_bound = the
 * least b with at most max_stored distinct resident hashes of b trailing zero bits, counted on the device), hands the
 * head its hashes that pass a mask of B bits (B = the largest lower bound of the shards in front of it, read order,
 * HOST array) and drops its stream; the head resolves its own shard and feeds the others' hashes in shard order.
 * sq_dedup_feed_hashes answers SQ_DEDUP_FEED_TOO_STRICT, with nothing done, when the estimator has fewer than
 * filtered_bits bits (hashes whose low bits are not spread): finish with the relay from that shard on. */
#define SQ_DEDUP_FEED_TOO_STRICT 1
uint64_t sq_dedup_lower_bound_of(const uint64_t *hist65, uint64_t max_stored);
int64_t sq_dedup_shard_store(sq_dedup *d, int head, uint8_t *bytes, uint8_t *known, size_t cap); /* -> fingerprint length */
int sq_dedup_shard_settle(sq_dedup *d, const uint8_t *store_in, size_t len, uint64_t *lower_bound);
int64_t sq_dedup_shard_passing(sq_dedup *d, uint64_t bits, uint64_t *out, size_t cap);
int sq_dedup_shard_drop(sq_dedup *d);
int sq_dedup_feed_hashes(sq_dedup *d, const uint64_t *hashes, size_t n, uint64_t filtered_bits,
                         const uint8_t *store_after, size_t store_len);

/* InsertSizeMetrics adapter tables (first-come cap, :5583,5599): ranks count the pairs
 * of the whole job and the shard's tables (2^table_bits slots) never close.  Keys are
 * 32 bytes {length, bytes[31]}; HOST arrays.  The histogram is a plain sum. */
int sq_insertsize_set_shard(sq_insertsize *z, uint64_t first_pair_index, uint32_t table_bits);
int64_t sq_insertsize_shard_candidates(sq_insertsize *z, int read2, uint8_t *keys, uint64_t *ranks, size_t cap);
int64_t sq_insertsize_shard_select(sq_insertsize *z, const uint8_t *keys, const uint64_t *ranks, size_t n,
                                   uint8_t *out_keys, uint64_t *out_ranks, size_t cap);
int sq_insertsize_shard_lookup(sq_insertsize *z, int read2, const uint8_t *keys, size_t n, uint64_t *counts);
int sq_insertsize_shard_install(sq_insertsize *z, int read2, const uint8_t *keys, const uint64_t *ranks,
                                const uint64_t *counts, size_t n, uint64_t n_events);
int sq_insertsize_shard_set_totals(sq_insertsize *z, uint64_t total_reads, const uint64_t *insert_sizes,
                                   size_t len);

/* PerTileQuality: sums per tile over the union of the shards' tiles; a shard behind the
 * job's first unparsable header contributes nothing (:3126). */
int64_t sq_pertile_first_unparsable(sq_pertile *p);
int sq_pertile_install(sq_pertile *p, const int64_t *tile_ids, size_t n_tiles, const double *errors,
                       const uint64_t *length_counts, size_t len, uint64_t number_of_reads,
                       const char *skipped_reason);

/* ---- multi-GPU: raw count tables for an all-reduce over RCCL --------------- */
/* Exposes the device arrays of the additive tables (u64 counts; f64 for the
 * per-tile error sums) so that the caller's collective can sum them in place
 * across ranks (torch.distributed / RCCL).  Fills up to `cap` entries of
 * (device pointer, element count); returns the number of arrays. */
int64_t sq_qcmetrics_device_tables(sq_qcmetrics *m, void **ptrs, uint64_t *counts, size_t cap);
int64_t sq_adaptercounter_device_tables(sq_adaptercounter *a, void **ptrs, uint64_t *counts, size_t cap);
/* Pads the tables to `length` rows (after an all-reduce(max) of max_length)
 * and, after the sum, sets the scalar members that are kept on the host. */
int sq_qcmetrics_reserve(sq_qcmetrics *m, uint64_t length);
int sq_qcmetrics_set_totals(sq_qcmetrics *m, uint64_t number_of_reads, uint64_t max_length);
int sq_adaptercounter_reserve(sq_adaptercounter *a, uint64_t length);
int sq_adaptercounter_set_totals(sq_adaptercounter *a, uint64_t number_of_sequences, uint64_t max_length);
/* Row length (in counters) of the forward / reverse tables, and a regrow to exactly `row_length`
 * (>= the current one): sq_adaptercounter_reserve grows geometrically, so ranks with different
 * batch histories hold different row lengths; the merge agrees on the largest first. */
uint64_t sq_adaptercounter_row_length(sq_adaptercounter *a);
int sq_adaptercounter_set_row_length(sq_adaptercounter *a, uint64_t row_length);

/* ---- multi-GPU without torch: RCCL through the C ABI (csrc/sq_dist.hip) ---------- */
/* The job's one exchange step (the reference has no counterpart: it is one process) for a host that binds
 * this library directly.  One process per GPU; librccl.so is opened on first use.  Rank 0: sq_rccl_unique_id, hand
 * the 128 bytes to every rank; all ranks: comm = sq_rccl_comm_init(ctx, n_ranks, id, rank); after the pass over the
 * rank's shard: sq_qcmetrics_allreduce / sq_adaptercounter_allreduce (the ranks agree on the longest read and the row
 * length, pad, sum in place over xGMI, set the totals: every rank then holds the job's tables and the getters of
 * _qc.pyi:60-75 answer for the whole job).  sq_rccl_allreduce_tables / sq_rccl_allgather_bytes are the collectives
 * themselves, for the tables of sq_*_device_tables and the candidate lists of the sq_*_shard_* entry points.
 * op: 0 sum of u64, 1 sum of f64, 2 max of u64.  Not yet run with more than one rank: no such node was available. */
int sq_rccl_available(void);
int sq_rccl_unique_id(uint8_t *out128);
void *sq_rccl_comm_init(sq_ctx *ctx, int n_ranks, const uint8_t *id128, int rank);
void sq_rccl_comm_destroy(void *comm);
int sq_rccl_allreduce_tables(sq_ctx *ctx, void *comm, void *const *ptrs, const uint64_t *counts, size_t n, int op);
int sq_rccl_allgather_bytes(sq_ctx *ctx, void *comm, const void *d_send, void *d_recv, size_t bytes);
int sq_qcmetrics_allreduce(sq_qcmetrics *m, void *comm);
int sq_adaptercounter_allreduce(sq_adaptercounter *a, void *comm);

/* test hook: the header parse of k_span<PT> (csrc/sq_pair.hip) compiled for the host; `name` has 64 readable bytes */
int64_t sq_test_tile_of_header(const uint8_t *name, uint32_t n);
int64_t sq_test_tile_of_header_quad(const uint8_t *name, uint32_t n);   /* the parse shared by the four lanes of a quad, emulated */

/* ---- synthetic FASTQ (bench / tests): counter-based, host == device bytes -- */
#define SQ_SYNTH_ILLUMINA 0       /* 150 bp single end / R1          */
#define SQ_SYNTH_ILLUMINA_R2 1    /* the mate of read i              */
#define SQ_SYNTH_NANOPORE 2       /* variable length, ~10 kb          */
#define SQ_SYNTH_ILLUMINA_BY_TILE 3 /* kind 0 with the reads ordered by tile */
#define SQ_SYNTH_ILLUMINA_R2_BY_TILE 4 /* kind 1 with the reads ordered by tile: the mates of kind 3 */
/* Illumina kinds: `kind | (L << 8)` makes reads of L bases instead of 150 (1 <= L <= 65535) */
#define SQ_SYNTH_KIND_LEN(kind, L) ((kind) | ((L) << 8))
/* Size in bytes of records [first, first+n) and the generators themselves.
 * Host version writes FASTQ text + metas into caller memory; device version
 * allocates a batch in HBM and fills it with a kernel. */
uint64_t sq_synth_bytes(int kind, uint64_t seed, uint64_t first, uint64_t n);
int sq_synth_host(int kind, uint64_t seed, uint64_t first, uint64_t n, uint8_t *buf,
                  size_t buf_cap, sq_meta *metas);
sq_batch *sq_synth_device(sq_ctx *ctx, int kind, uint64_t seed, uint64_t first, uint64_t n);
/* Cuts every read of a batch in HBM to a length in [lo, its length], by a hash of (seed, record
 * index): what adapter trimming leaves of a file of one read length (bench / tests: the ragged
 * variant of the synthetic workload).  Only sequence_length changes; the statistics of the batch
 * are recomputed. */
int sq_synth_trim(sq_batch *b, uint64_t seed, uint32_t lo);

#ifdef __cplusplus
}
#endif
#endif /* SQGPU_H */
