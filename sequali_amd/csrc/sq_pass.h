/* sq_pass.h -- what the kernels of the per-base pass share (sq_qc.hip, sq_span.hip): the launch
 * parameters, LDS histogram geometry, class codes of a dword of bases, LDS addressing helpers. */
#ifndef SQ_PASS_H
#define SQ_PASS_H

#include "sq_common.h"
#include "sq_error_table.h"

/* parameters of one launch of the per-base pass (global scope: sq_span.hip takes them from sq_qc.hip) */
struct sq_carry;
struct PassParams {
    const uint8_t *buf;
    uint64_t buf_len;
    sq_meta *metas;
    uint64_t n;
    uint64_t first_read_index; /* index of record 0 over everything the module saw */
    uint32_t lds_len;          /* positions covered by the LDS histograms */
    /* stripes (reads longer than the LDS histograms): one launch covers positions
       [pos_base, pos_end) of the reads that are longer than pos_base; what a read carries from
       one stripe to the next (the f64 chains, the automaton, its base counts) lives in `carry` */
    uint32_t pos_base, pos_end;
    struct sq_carry *carry;    /* [records in processing order], NULL: one stripe holds every read */
    const uint32_t *order;     /* processing order of the records (NULL: as stored) */
    uint32_t blocked;          /* != 0: a wave takes a contiguous run of groups (tile-sorted order:
                                  concurrent waves then sit in different tiles) */
    /* QCMetrics */
    unsigned long long *qc_base, *qc_phred, *qc_ea_base, *qc_ea_phred, *qc_gc, *qc_ps;
    uint32_t ea_len;
    uint32_t ea_in_lds;
    uint32_t uniform_len;     /* != 0: every record of the batch has this length (<= lds_len) */
    const double *thresholds; /* [94], see phred_thresholds() */
    const uint32_t *span_bounds;   /* k_span over sorted rows: first span of every workgroup's stretch ([grid + 1]); NULL: equal shares */
    const double *thr_sum;    /* [257][96]: row U = the same thresholds for the SUM of the error rates of a read of U bases (phred_sum_thresholds()) */
    unsigned long long *qc_first_bad;
    /* AdapterCounter */
    const uint16_t *dfa;      /* [states][8] */
    uint32_t dfa_states;
    uint32_t dfa_accept;      /* states >= this one are the ones some adapter ends in */
    const unsigned long long *dfa_out; /* [states] adapters ending in that state */
    const uint16_t *dfa2;     /* [dfa2_states][36]: the automaton two characters at a time (build_pair_dfa) */
    uint32_t dfa2_states;     /* states >= dfa_accept report, as in the one-character automaton */
    const unsigned long long *dfa2_out; /* [dfa2_states][2] adapters ending on the second / the first character of the step */
    const uint8_t *ad_len;    /* [n_adapters] */
    unsigned long long *ad_fwd, *ad_rev; /* [n_adapters][ad_cap] */
    uint64_t ad_cap;
    uint32_t ad_lds;          /* != 0: number of adapters whose hits a workgroup counts in LDS first
                                 (batches of one read length: the reverse table is derived at the merge) */
    uint32_t ad_maxlen;       /* longest adapter of this automaton */
    /* PerTileQuality */
    const int32_t *pt_slot;   /* per record, from k_tile_prepass */
    unsigned long long *pt_len_counts;
    double *pt_errors;
    uint64_t pt_cap;          /* row length of the two tables */
    uint64_t pt_first_bad;    /* records >= this are ignored */
    /* k_span over reads sorted by length (sq_span.h): spans of 16 reads of ONE length */
    const struct SpanSeg *span_segs; /* the length classes of this launch, in span order */
    uint32_t span_nsegs;
    uint32_t span_total;       /* spans of the launch */
    const struct SpanRow *span_rows; /* the batch's reads sorted by length, longest first */
    unsigned int *long_first;  /* k_span<LONG>: [records][n_adapters] start of the first occurrence found so far, ~0: none */
    unsigned int *long_gc;     /* k_span<LONG>: [records][2] G/C bases and A/C/G/T bases of the read, summed segment by segment */
    /* k_span<PT> (sq_pair.hip): PerTileQuality rides in the QCMetrics pass.  The tile id is parsed from the header
       bytes (the lines the pass fetches anyway); a wave sums the error rates of consecutive reads of ONE tile in
       registers (a sequencer writes its reads tile by tile) and stages every such run -- tile, reads, a sum per
       position -- for k_pt_fold, which knows the table rows.  Nothing reaches PerTileQuality's tables from the pass
       itself: a batch with a header that does not parse, or with more runs than fit, is counted by the older route */
    long long *pt_tiles;            /* [n] tile id of every record (< 0: the header does not parse); NULL: not wanted */
    unsigned long long *pt_bad;     /* atomicMin: pt_first_index + index of the first record whose header does not parse */
    uint64_t pt_first_index;
    struct PtRun *pt_runs;          /* [pt_runs_cap]; NULL: tiles only */
    double *pt_run_sums;            /* [pt_runs_cap][uniform_len] */
    unsigned int *pt_nruns;         /* runs asked for so far (more than pt_runs_cap: the rest was not stored) */
    uint32_t pt_runs_cap;
    /* k_span<PAIR> (sq_pair.hip): the overlap scan of InsertSizeMetrics (calculate_insert_size, :5667-5707) split over
       the passes of the two mates.  PAIR = 1 (read 2): the first and the last 16 bases of every read go to pair_ends
       ([n][32] bytes: head, tail) on the way.  PAIR = 2 (read 1): a span's 512 bytes of ends come in by one LDS-DMA and
       the scan runs on the sequences the pass holds in LDS anyway; pair_results[pair] = its insert size (0: none) */
    uint8_t *pair_ends;
    uint32_t *pair_results;
    uint32_t pair_L2;               /* PAIR = 2: the length of the reads of read 2 (>= 16) */
};
struct PtRun { long long tile; uint32_t reads; uint32_t pad; };

namespace {

constexpr int WG_THREADS = 256;
constexpr int WAVES = WG_THREADS / 64;
constexpr uint32_t PAD4 = 0x80808080u;   /* quality tile: past the end of the read */
constexpr uint32_t LDS_HIST_MAX = 512;  /* positions kept in LDS histograms */
constexpr uint32_t STRIPE = 512;        /* positions per launch when reads are longer than that (256: more launches, slower) */
constexpr uint32_t LDS_EA_MAX = 256;    /* end-anchor rows kept in LDS */
constexpr uint32_t DFA_LDS_MAX_STATES = 1024; /* 16 KB of LDS */
/* LDS histograms are class-major, [class][position] with the position stride rounded
 * up to 32 words: the lanes of one atomic instruction hold consecutive positions, so
 * whatever their classes they fall into 32 different banks */
constexpr uint32_t BASE_COLS = 5, PHRED_COLS = 12;
__host__ __device__ inline uint32_t hist_stride(uint32_t rows) { return (rows + 31u) & ~31u; }
constexpr int64_t TILE_EMPTY = -1;
constexpr uint32_t TILE_MAP_SIZE = 1u << 16;

/* SCORE_TO_ERROR_RATE as bit patterns (score_to_error_rate.h:4-99) */
__constant__ unsigned long long c_error_rate_bits[94] = {SQ_ERROR_RATE_BITS_LIST};


__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
    for (int off = 32; off > 0; off >>= 1) {
        uint32_t o = __shfl_xor(v, off);
        v = o > v ? o : v;
    }
    return v;
}

/* 16 bytes from an arbitrary byte address; falls back to byte loads when the
 * 16-byte window would leave the buffer */
__device__ __forceinline__ uint4 load16(const uint8_t *buf, uint64_t off, uint64_t buf_len)
{
    uint4 v;
    if (off + 16 <= buf_len) {
        __builtin_memcpy(&v, buf + off, 16);
    } else {
        uint32_t w[4] = {PAD4, PAD4, PAD4, PAD4};
        for (int k = 0; k < 16; k++) {
            if (off + k < buf_len) {
                uint32_t sh = 8 * (k & 3);
                w[k >> 2] = (w[k >> 2] & ~(0xFFu << sh)) | ((uint32_t)buf[off + k] << sh);
            }
        }
        v = make_uint4(w[0], w[1], w[2], w[3]);
    }
    return v;
}

/* keep the first `nvalid` (0..4) bytes of w, the rest become `pad` */
__device__ __forceinline__ uint32_t pad_tail(uint32_t w, int nvalid, uint32_t pad)
{
    if (nvalid >= 4) return w;
    if (nvalid <= 0) return pad;
    uint32_t keep = (1u << (8 * nvalid)) - 1u;
    return (w & keep) | (pad & ~keep);
}

/* Four sequence bytes -> four class codes, doubled (A 0, C 2, G 4, T 6, other 8),
 * without a table in memory.  The low three bits of A/C/G/T (either case) are
 * 1/3/7/4 (the observation the reference cites from fastp, _qcmodule.c:1731-1739)
 * and index two 8-byte LUTs held in registers (v_perm_b32): the class and the
 * upper-case letter that class requires.  A byte whose upper-cased value is not
 * that letter is class 4 like in NUCLEOTIDE_TO_INDEX (:1748-1763).  Bytes must be
 * 7-bit ASCII (the parser guarantees it, :1055). */
constexpr uint32_t CLS2_PAD4 = 0x0E0E0E0Eu; /* class 7 = past the end of the read */
__device__ __forceinline__ uint32_t cls2_of_dword(uint32_t w)
{
    const uint32_t idx = w & 0x07070707u;
    const uint32_t lut = __builtin_amdgcn_perm(0x04080806u, 0x02080008u, idx);
    const uint32_t want = __builtin_amdgcn_perm(0x47000054u, 0x43004100u, idx);
    const uint32_t d = (w & 0xDFDFDFDFu) ^ want;
    const uint32_t ne4 = ((d + 0x7F7F7F7Fu) & 0x80808080u) >> 5; /* 4 where it is not that letter */
    /* a third v_perm picks, byte by byte, the class (selectors 4-7: bytes of lut) or 8 (selectors
       0-3: bytes of the constant) */
    return __builtin_amdgcn_perm(lut, 0x08080808u, 0x07060504u - ne4);
}

/* &hist[row * stride] with a full-rate 24-bit multiply-add (v_mad_u32_u24): a plain 32-bit
 * multiply is a quarter-rate instruction and there are two of these per base */
__device__ __forceinline__ uint32_t *hist_row(uint32_t *base, uint32_t row, uint32_t stride_bytes)
{
    return (uint32_t *)((uint8_t *)base + __umul24(row, stride_bytes));
}

/* A wave walks its 64 reads in chunks of CW positions. */
constexpr uint32_t CW = 32;               /* positions per chunk */
constexpr uint32_t ROW_WORDS = CW / 4;    /* dwords per read and chunk in a tile */
constexpr uint32_t TILE_WORDS = 64 * ROW_WORDS;
constexpr uint32_t WAVE_WORDS = 2 * TILE_WORDS + 128 + 128 + 64;
constexpr uint32_t FIXED_BYTES = 136 * 8 + 96 * 8 + 104 * 4 + 96 * 4;

/* tile address of dword d of row r: rows are ROW_WORDS = 8 dwords, the dword index
 * is XOR-ed with bits of the row so that "lane = row, same dword" (phase S),
 * "lane = position, one or two rows" (phase H) and the staging writes all spread
 * over the 32 banks */
__device__ __forceinline__ uint32_t tile_idx(uint32_t row, uint32_t d)
{
    return row * ROW_WORDS + (d ^ ((row >> 2) & 7));
}

/* k_pass keeps a wave's two tiles interleaved in blocks of 8 rows (256 B of sequence classes,
 * then the 256 B of qualities of the same rows): the quality word of a tile word is always
 * 64 words further on, which one ds_read2_b32 reaches, and rows 8d .. 8d+7 (what one step of
 * the fused loop touches in phase H) are block d.  Banks are those of tile_idx. */
constexpr uint32_t QUAL_WORDS = 64;       /* w_qual = w_seq + QUAL_WORDS */
__device__ __forceinline__ uint32_t ptile_idx(uint32_t row, uint32_t d)
{
    return (row >> 3) * 128 + (row & 7) * ROW_WORDS + (d ^ ((row >> 2) & 7));
}
/* phase H: dword h_dw of row 2 * rp + half */
__device__ __forceinline__ uint32_t ptile_idx_h(uint32_t rp, uint32_t half, uint32_t h_dw)
{
    return (rp >> 2) * 128 + (2 * (rp & 3) + half) * ROW_WORDS + (h_dw ^ ((rp >> 1) & 7));
}

/* LDS through 32-bit addresses: the fused loop computes them with one instruction each */
#define SQ_LDS __attribute__((address_space(3)))
__device__ __forceinline__ uint32_t lds_addr(const void *p) { return (uint32_t)(uintptr_t)(SQ_LDS const uint8_t *)p; }
__device__ __forceinline__ uint32_t lds_u32(uint32_t a) { return *(SQ_LDS const uint32_t *)(uintptr_t)a; }
__device__ __forceinline__ uint32_t lds_u16(uint32_t a) { return *(SQ_LDS const uint16_t *)(uintptr_t)a; }
__device__ __forceinline__ double lds_f64(uint32_t a) { return *(SQ_LDS const double *)(uintptr_t)a; }
/* a | byte J of w, and byte J of w << 3, in one instruction each (SDWA operand selects) */
template <int J> __device__ __forceinline__ uint32_t or_byte(uint32_t a, uint32_t w)
{
    uint32_t r;
    if (J == 0) asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(a), "v"(w));
    if (J == 1) asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(a), "v"(w));
    if (J == 2) asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(a), "v"(w));
    if (J == 3) asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(a), "v"(w));
    return r;
}
template <int J> __device__ __forceinline__ uint32_t shl3_byte(uint32_t w)
{
    uint32_t r;
    if (J == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(3u), "v"(w));
    if (J == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(3u), "v"(w));
    if (J == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(3u), "v"(w));
    if (J == 3) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(3u), "v"(w));
    return r;
}
__device__ __forceinline__ void lds_inc(uint32_t a)
{
    __hip_atomic_fetch_add((SQ_LDS uint32_t *)(uintptr_t)a, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
/* phred bins of two quality bytes at once (packed 16-bit math): byte `sel`-selected of qa in
 * the low half, of qb in the high half; min(q - 33, 47) >> 2 like the scalar form */
typedef unsigned short sq_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t bins_of_two(uint32_t qa, uint32_t qb, uint32_t sel)
{
    sq_us2 v = __builtin_bit_cast(sq_us2, __builtin_amdgcn_perm(qb, qa, sel));
    v = __builtin_elementwise_min(v - (unsigned short)33, (sq_us2)(unsigned short)47) >> (unsigned short)2;
    return __builtin_bit_cast(uint32_t, v);
}
/* base + (low / high half of packed) * stride: v_mad_u32_u16, the half picked by op_sel */
template <int HI> __device__ __forceinline__ uint32_t mad_half(uint32_t packed, uint32_t stride, uint32_t base)
{
    uint32_t r;
    if (HI) asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(r) : "v"(packed), "s"(stride), "v"(base));
    else asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(r) : "v"(packed), "s"(stride), "v"(base));
    return r;
}
/* (x ^ s) + b */
__device__ __forceinline__ uint32_t xor_add(uint32_t x, uint32_t s, uint32_t b)
{
    uint32_t r;
    asm("v_xad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "s"(s), "v"(b));
    return r;
}

/* illumina_header_to_tile_id, _qcmodule.c:3088-3121 (+ :159-180): the decimal number
 * between the 4th and the 5th ':' of the header.  Colons are found eight bytes at a
 * time (headers are 7-bit ASCII, so x + 0x7F sets bit 7 of every non-zero byte). */
__host__ __device__ inline long long tile_id_of(const uint8_t *name, uint32_t n)
{
    uint32_t colons = 0, c4 = n, c5 = n;
    for (uint32_t off = 0; off < n && c5 == n; off += 8) {
        /* the sequence follows the name inside the same buffer: reading 8 bytes is safe */
        uint64_t w = sq_load_u64_unaligned(name + off);
        if (n - off < 8) w |= ~0ULL << (8 * (n - off)); /* bytes past the name never match */
        const uint64_t x = w ^ 0x3A3A3A3A3A3A3A3AULL;
        uint64_t m = ~((x & 0x7F7F7F7F7F7F7F7FULL) + 0x7F7F7F7F7F7F7F7FULL | x) & 0x8080808080808080ULL;
        while (m) {
            const uint32_t at = off + ((uint32_t)__builtin_ffsll((long long)m) - 1) / 8;
            m &= m - 1;
            colons++;
            if (colons == 4) c4 = at;
            else if (colons == 5) { c5 = at; break; }
        }
    }
    if (c5 == n) return -1;           /* fewer than five colons */
    const uint32_t start = c4 + 1, len = c5 - start;
    if (len < 1 || len > 18) return -1;
    unsigned long long v = 0;
    for (uint32_t k = start; k < c5; k++) {
        const uint32_t d = (uint32_t)name[k] - '0';
        if (d > 9) return -1;
        v = v * 10 + d;
    }
    return (long long)v;
}

/* The same for a header of at most 8 NWORDS bytes whose bytes were fetched with independent
 * 16-byte loads (tile_id_of asks memory for 8 bytes at a time, each load waiting for the scan
 * of the one before: the slowest way to gather).  w[] holds the bytes little endian.  Returns
 * -2 when the tile field is longer than 8 digits: the caller falls back to tile_id_of. */
template <int NWORDS>
__host__ __device__ inline long long tile_id_of_words(const uint64_t (&w)[NWORDS], uint32_t n)
{
    uint32_t colons = 0, c4 = n, c5 = n;
#pragma unroll
    for (uint32_t k = 0; k < (uint32_t)NWORDS; k++) {
        const uint32_t off = 8 * k;
        if (off < n && c5 == n) {
            uint64_t x = w[k];
            if (n - off < 8) x |= ~0ULL << (8 * (n - off)); /* bytes past the name never match */
            x ^= 0x3A3A3A3A3A3A3A3AULL;
            uint64_t m = ~((x & 0x7F7F7F7F7F7F7F7FULL) + 0x7F7F7F7F7F7F7F7FULL | x) & 0x8080808080808080ULL;
            while (m) {
                const uint32_t at = off + ((uint32_t)__builtin_ffsll((long long)m) - 1) / 8;
                m &= m - 1;
                colons++;
                if (colons == 4) c4 = at;
                else if (colons == 5) { c5 = at; break; }
            }
        }
    }
    if (c5 == n) return -1;
    const uint32_t start = c4 + 1, len = c5 - start;
    if (len < 1 || len > 18) return -1;
    if (len > 8) return -2;
    /* the 8 bytes from `start` on: a funnel over two neighbouring words */
    const uint32_t wi = start >> 3, sh = 8 * (start & 7);
    uint64_t lo = w[0], hi = w[1];
#pragma unroll
    for (uint32_t k = 1; k < (uint32_t)NWORDS; k++) {
        if (wi == k) { lo = w[k]; hi = k + 1 < (uint32_t)NWORDS ? w[k + 1] : 0; }
    }
    const uint64_t win = sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
    unsigned long long v = 0;
    for (uint32_t k = 0; k < len; k++) {
        const uint32_t d = (uint32_t)((win >> (8 * k)) & 0xFF) - '0';
        if (d > 9) return -1;
        v = v * 10 + d;
    }
    return (long long)v;
}



/* ---- the same parse by the four lanes of a read's quad (k_span<PT>: lane c holds bytes 16 c .. 16 c + 15 of the header)
 * tile_id_of_words runs on one lane in four with loops that diverge between rows; here every lane works on its own 16
 * bytes and the quad exchanges five small values.  Pure per-lane functions (compiled for the host too and checked
 * there, quad_tile_id_host below); the kernel does the exchanges by DPP.  Anything out of the ordinary -- fewer than five
 * colons inside the first 64 bytes of a longer header, a tile field of more than 8 or of no digits -- is left to the
 * caller's byte-by-byte parse (QUAD_TILE_SLOW). */
constexpr long long QUAD_TILE_SLOW = -2;
/* bit i set: byte i of the lane's 16 is a ':' and lies inside the name */
__host__ __device__ inline uint32_t quad_colon_mask(const uint32_t (&w)[4], uint32_t c, uint32_t nlen)
{
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t x = w[k] ^ 0x3A3A3A3Au;
        const uint32_t z = ((((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u) ^ 0x80808080u;   /* bit 7 of the bytes that are ':' */
        m |= ((((z >> 7) * 0x00204081u) >> 21) & 0xFu) << (4 * k);                                  /* bits 7, 15, 23, 31 -> 4 bits */
    }
    const uint32_t first = 16 * c, valid = nlen > first ? (nlen - first < 16 ? nlen - first : 16) : 0;
    return m & ((1u << valid) - 1u);   /* valid <= 16: no shift by 32 */
}
/* position in the header of the k-th colon (k = 1 ..) if it lies in this lane's bytes (`before` colons lie in the lanes in
   front, `m` is this lane's mask), else 0xFF */
__host__ __device__ inline uint32_t quad_kth_colon(uint32_t m, uint32_t before, uint32_t k, uint32_t c)
{
    const uint32_t cnt = (uint32_t)__builtin_popcount(m);
    if (!(before < k && k <= before + cnt)) return 0xFFu;
    uint32_t r = k - before;   /* 1 .. 5 */
#pragma unroll
    for (int i = 1; i < 5; i++)
        if ((uint32_t)i < r) m &= m - 1;
    return 16 * c + (uint32_t)__builtin_ctz(m);
}
/* this lane's share of the 8 bytes of the header from byte s0 on (bytes of other lanes: 0), little endian.  (Scalars, no
   arrays: a local array that is indexed by anything but a literal lands in scratch memory on the device.) */
__host__ __device__ inline uint64_t quad_window8(const uint32_t (&w)[4], uint32_t c, uint32_t s0)
{
    const int d = (int)s0 - (int)(16 * c);
    const bool none = d >= 16 || d <= -8;
    const uint32_t off = (uint32_t)(8 + (none ? 0 : d)), sh = off & 3;   /* the window starts `off` bytes into [8 zero bytes | the 16 | 8 zero bytes] */
    const bool s4 = (off & 16) != 0, s2 = (off & 8) != 0, s1 = (off & 4) != 0;   /* its first dword: off >> 2 = 4 s4 + 2 s2 + s1 (<= 5) */
    /* A = 0 0 w0 w1 w2 w3 0 0; B[k] = s4 ? A[k + 4] : A[k] (k < 6) */
    const uint32_t b0 = s4 ? w[2] : 0u, b1 = s4 ? w[3] : 0u, b2 = s4 ? 0u : w[0], b3 = s4 ? 0u : w[1], b4 = s4 ? 0u : w[2], b5 = s4 ? 0u : w[3];
    /* C[k] = s2 ? B[k + 2] : B[k] (k < 4) */
    const uint32_t c0 = s2 ? b2 : b0, c1 = s2 ? b3 : b1, c2 = s2 ? b4 : b2, c3 = s2 ? b5 : b3;
    /* D[k] = s1 ? C[k + 1] : C[k] (k < 3) */
    const uint32_t d0 = s1 ? c1 : c0, d1 = s1 ? c2 : c1, d2 = s1 ? c3 : c2;
    const uint64_t lo = (((uint64_t)d1 << 32) | d0) >> (8 * sh), hi = (((uint64_t)d2 << 32) | d1) >> (8 * sh);
    const uint64_t v = (uint64_t)(uint32_t)lo | ((uint64_t)(uint32_t)hi << 32);
    return none ? 0 : v;
}
/* unsigned_decimal_integer_from_string (:159-180) of the first `len` (1 .. 8) bytes of d8; -1 for a byte that is no digit */
__host__ __device__ inline long long quad_digits_value(uint64_t d8, uint32_t len)
{
    uint32_t v = 0;
    bool bad = false;
#pragma unroll
    for (uint32_t i = 0; i < 8; i++) {
        const uint32_t dgt = (uint32_t)((d8 >> (8 * i)) & 0xFF) - '0';
        if (i < len) {
            bad |= dgt > 9;
            v = v * 10 + dgt;
        }
    }
    return bad ? -1 : (long long)v;
}
/* what the quad makes of it once the exchanges are done: c4 / c5 = positions of the 4th / 5th colon (0xFF: none inside the
   bytes looked at), nlen = the name's length, d8 = the 8 bytes behind the 4th colon */
__host__ __device__ inline long long quad_tile_value(uint32_t c4, uint32_t c5, uint32_t nlen, uint64_t d8)
{
    if (c5 == 0xFFu) return nlen > 64 ? QUAD_TILE_SLOW : -1;   /* fewer than five colons (:3116-3120) -- as far as the 64 bytes show */
    const uint32_t len = c5 - c4 - 1;
    if (len < 1 || len > 18) return -1;                         /* :164-166 */
    if (len > 8) return QUAD_TILE_SLOW;
    return quad_digits_value(d8, len);
}
/* the whole parse on the host, lane by lane and exchange by exchange as the kernel does it (tests) */
inline long long quad_tile_id_host(const uint8_t *name64, uint32_t nlen)
{
    uint32_t w[4][4], m[4], before[4], c4 = 0xFFu, c5 = 0xFFu;
    for (uint32_t c = 0; c < 4; c++) {
        memcpy(w[c], name64 + 16 * c, 16);
        m[c] = quad_colon_mask(w[c], c, nlen);
    }
    for (uint32_t c = 0; c < 4; c++) {
        before[c] = 0;
        for (uint32_t j = 0; j < c; j++) before[c] += (uint32_t)__builtin_popcount(m[j]);
    }
    for (uint32_t c = 0; c < 4; c++) {
        c4 = std::min(c4, quad_kth_colon(m[c], before[c], 4, c));
        c5 = std::min(c5, quad_kth_colon(m[c], before[c], 5, c));
    }
    uint64_t d8 = 0;
    for (uint32_t c = 0; c < 4; c++) d8 |= quad_window8(w[c], c, c4 == 0xFFu ? 0 : c4 + 1);
    return quad_tile_value(c4, c5, nlen, d8);
}

} // namespace

#endif
