/*
 * sq_span.hip -- k_span: QCMetrics (+ AdapterCounter) on batches of one read length, the
 * records streamed through LDS by LDS-DMA.
 *
 * What the reference computes per read (QCMetrics_add_meta, _qcmodule.c:1966-2139;
 * AdapterCounter_add_meta, :2786-2823) is restated as in k_pass / k_wide; what differs is how the
 * bytes reach the lanes.  k_wide gathers 64 bytes per row and visit through registers and is
 * bound by that gather.  Here a wave owns *spans* of 16 consecutive records:
 *
 *   - DMA: `global_load_lds_dwordx4` copies every record's sequence and quality bytes
 *     straight into an LDS slot, 16 bytes per lane, the source address of a lane being
 *     `sequence start + 16 k` or `qualities start + 16 k` at whatever byte alignment that has:
 *     in LDS every row starts 16-byte aligned (row r: sequence at r * ROWB, qualities at
 *     r * ROWB + SB; SB = 32 NW >= read length, ROWB = 2 SB), so a slot is exactly NW DMA
 *     instructions of 64 lanes and every later access has a compile-time offset.  The span after
 *     the current one lands in the wave's second slot while this one is counted (no registers,
 *     no waits in between: `s_waitcnt vmcnt(0)` once per span).
 *   - phase S, four lanes per read (16 reads per wave): lane c of a quad adds the error rates of
 *     positions = c (mod 4), in position order, to chain c: the reference's four interleaved f64
 *     chains (:2062-2097), one per lane.  The same lane walks the automaton over quarter c of the
 *     read, restarted `longest adapter - 1` positions in front of its quarter (what the automaton
 *     knows about the text is at most that long); a match belongs to the quarter its last base
 *     lies in and is only a candidate: `ds_min` per (read, adapter) keeps the first (:2657-2668).
 *     On the way the lane turns its quarter's bases into class codes *in place* (phase H reads
 *     them) and counts G/C and non-ACGT bases.
 *   - phase H, lane = position: 2 rows x 32 positions per instruction, bytes read with
 *     `ds_read_u8` at compile-time offsets.  Base counts never touch LDS per base: a class code
 *     is the shift of its 6-bit field (A 0, C 6, G 12, T 18, N 24, padding 30: that field
 *     overflows out of the register), one `v_lshl_add_u32` per base into a register per
 *     (lane, window), unpacked into the LDS histogram every seventh span.  Phred bins are LDS
 *     atomics as in k_wide.
 */
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include <algorithm>
#include <type_traits>

#include "sq_pass.h"
#include "sq_span.h"
#include <hipcub/hipcub.hpp>

#include "sq_span_kernel.h"

namespace {

template <int NW, bool SEG, bool SPLIT>
int launch_nw(sq_ctx *ctx, const PassParams &P0, bool ad, uint32_t n_ad, int waves, size_t lds, int grid)
{
    PassParams P = P0;
    static bool attr = false;
    constexpr bool HAS_AD = NW <= (SPLIT ? SPAN_NW_AD_SPLIT : SPAN_NW_AD);
    if (!attr) {
        if constexpr (HAS_AD)
            SQ_HIP(hipFuncSetAttribute((const void *)k_span<NW, true, SEG, SPAN_W4, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        SQ_HIP(hipFuncSetAttribute((const void *)k_span<NW, false, SEG, SPAN_W4, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    sq_route(ctx, "k_span<%d,%s,%s,%s>", NW, ad ? "AD" : "QC", SEG ? "sorted" : "uniform", SPLIT ? "split" : "both");
    if (ad) {
        if constexpr (HAS_AD)
            hipLaunchKernelGGL((k_span<NW, true, SEG, SPAN_W4, SPLIT>), dim3(grid), dim3(waves * 64), lds, ctx->stream, P, n_ad);
        else
            return SQ_ERR_SYSTEM;
    } else {
        hipLaunchKernelGGL((k_span<NW, false, SEG, SPAN_W4, SPLIT>), dim3(grid), dim3(waves * 64), lds, ctx->stream, P, n_ad);
    }
    SQ_HIP(hipGetLastError());
    return SQ_OK;
}
template <bool SEG, bool SPLIT>
int launch_any(int nw, sq_ctx *ctx, const PassParams &P, bool ad, uint32_t n_ad, int waves, size_t lds, int grid)
{
#ifdef SQ_SPAN_ONLY_NW   /* experiment builds: one window count, a quarter of the compile time */
    if (nw != SQ_SPAN_ONLY_NW) { sq_set_error("this build holds k_span<%d> only", SQ_SPAN_ONLY_NW); return SQ_ERR_SYSTEM; }
    return launch_nw<SQ_SPAN_ONLY_NW, SEG, SPLIT>(ctx, P, ad, n_ad, waves, lds, grid);
#else
    switch (nw) {
        case 1: return launch_nw<1, SEG, SPLIT>(ctx, P, ad, n_ad, waves, lds, grid);
        case 2: return launch_nw<2, SEG, SPLIT>(ctx, P, ad, n_ad, waves, lds, grid);
        case 3: return launch_nw<3, SEG, SPLIT>(ctx, P, ad, n_ad, waves, lds, grid);
        case 4: return launch_nw<4, SEG, SPLIT>(ctx, P, ad, n_ad, waves, lds, grid);
        case 5: return launch_nw<5, SEG, SPLIT>(ctx, P, ad, n_ad, waves, lds, grid);
        case 6: return launch_nw<6, SEG, SPLIT>(ctx, P, ad, n_ad, waves, lds, grid);
        case 7: return launch_nw<7, SEG, SPLIT>(ctx, P, ad, n_ad, waves, lds, grid);
        default: return launch_nw<8, SEG, SPLIT>(ctx, P, ad, n_ad, waves, lds, grid);
    }
#endif
}

/* A register spilled inside the loop is reloaded behind an `s_waitcnt vmcnt(0)`, which also waits for
   the span in flight (the LDS-DMA counts in vmcnt): a build of the kernel with a wave per stream
   that spills is not used (the one with a wave for both streams has 168 registers) */
/* does the build <NW, ad, SEG, SPLIT> use scratch memory?  A register spilled inside the loop is reloaded behind an
   s_waitcnt vmcnt(0), which also waits for the span in flight (the LDS-DMA counts in vmcnt): such a build is not used */
template <int NW, bool SEG, bool SPLIT>
bool span_build_spills(bool ad)
{
    static int spills[2] = {-1, -1};
    if (spills[ad] < 0) {
        hipFuncAttributes fa{};
        const void *fn = (const void *)k_span<NW, false, SEG, SPAN_W4, SPLIT>;
        if constexpr (NW <= (SPLIT ? SPAN_NW_AD_SPLIT : SPAN_NW_AD)) if (ad) fn = (const void *)k_span<NW, true, SEG, SPAN_W4, SPLIT>;
        spills[ad] = hipFuncGetAttributes(&fa, fn) == hipSuccess && fa.localSizeBytes > 0 ? 1 : 0;
    }
    return spills[ad] != 0;
}
template <bool SEG, bool SPLIT>
bool span_build_spills_any(int nw, bool ad)
{
#ifdef SQ_SPAN_ONLY_NW
    return nw == SQ_SPAN_ONLY_NW ? span_build_spills<SQ_SPAN_ONLY_NW, SEG, SPLIT>(ad) : true;
#else
    switch (nw) {
        case 1: return span_build_spills<1, SEG, SPLIT>(ad);
        case 2: return span_build_spills<2, SEG, SPLIT>(ad);
        case 3: return span_build_spills<3, SEG, SPLIT>(ad);
        case 4: return span_build_spills<4, SEG, SPLIT>(ad);
        case 5: return span_build_spills<5, SEG, SPLIT>(ad);
        case 6: return span_build_spills<6, SEG, SPLIT>(ad);
        case 7: return span_build_spills<7, SEG, SPLIT>(ad);
        default: return span_build_spills<8, SEG, SPLIT>(ad);
    }
#endif
}

static bool span_needs_w6(const PassParams &P, bool ad) { return ad && (P.ad_maxlen + 2) / 4 > SPAN_W4; }

/* does k_span take this pass at all, and with how many waves per workgroup (split: an even number) */
int span_waves(const PassParams &P, int nw, uint32_t U, bool ad, uint32_t n_ad, bool seg, bool split)
{
    if (nw < 1 || nw > SPAN_NW_MAX || (ad && nw > (split ? SPAN_NW_AD_SPLIT : SPAN_NW_AD))) return 0; /* unsplit: the automaton's rounds spill registers from 161 positions on */
    if (ad && (SPAN_STATES(P) > SPAN_DFA_MAX_STATES || n_ad > 64)) return 0;
    /* the restart in front of a lane's quarter holds >= maxlen - 1 positions: three dwords (adapters of up to 13
       characters) in the builds of this file, six (up to 25) in those of sq_span_w6.hip */
    if (span_needs_w6(P, ad)) {
        const int w6 = sq_knobs().span_w6;
        if (w6 == 0 || (w6 < 0 && !seg && nw < 5) || (P.ad_maxlen + 2) / 4 > 6 || !sq_span_w6_exists(nw, seg, split)) return 0;
        if (sq_span_w6_spills(nw, seg, split)) return 0;
    } else if (
        (seg ? (split ? span_build_spills_any<true, true>(nw, ad) : span_build_spills_any<true, false>(nw, ad))
             : (split ? span_build_spills_any<false, true>(nw, ad) : span_build_spills_any<false, false>(nw, ad)))) return 0;
    const int step = split ? 2 : 1;
    int waves = span_max_waves(nw, split, seg);   /* as many as LDS takes */
    while (waves >= 4 && span_lds_layout(nw, U, ad ? SPAN_STATES(P) : 0, ad ? n_ad : 0, ad ? P.ad_lds : 0, waves, seg, split).total > 160 * 1024) waves -= step;
    if (waves < 4) return 0;
    return waves;
}

/* sort keys (longest first) and rows of the records of a batch, in stored order */
__global__ void k_span_keys(const sq_meta *metas, uint64_t n, uint32_t max_len, uint32_t *keys, SpanRow *rows)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const sq_meta m = metas[i];
        keys[i] = max_len - m.sequence_length;
        SpanRow r;
        r.seq = m.record_start + m.sequence_offset;
        r.qual_delta = m.qualities_offset - m.sequence_offset;
        r.record = (uint32_t)i;
        rows[i] = r;
    }
}
/* The rows of a batch in order of length (longest first) without a sort: how many reads have each length is
 * known since the batch was made (sq_batch::len_hist), so where the rows of a length start is too (start[L]).  A
 * workgroup takes 8 K records at a time: ranks them by length in LDS, reserves its share of every length's rows
 * with one atomic per length, and writes its rows there.  The order inside a length is whatever the atomics make
 * it: every table is a sum over reads. */
constexpr int SCATTER_THREADS = 512, SCATTER_PER = 16;
__global__ void __launch_bounds__(SCATTER_THREADS) k_span_scatter(const sq_meta *metas, uint64_t n, const uint32_t *start, uint32_t *cursor, SpanRow *rows)
{
    __shared__ uint32_t l_cnt[SQ_LEN_BINS], l_base[SQ_LEN_BINS];
    const uint64_t chunk = (uint64_t)SCATTER_THREADS * SCATTER_PER;
    for (uint64_t c0 = blockIdx.x * chunk; c0 < n; c0 += gridDim.x * chunk) {
        for (int i = threadIdx.x; i < SQ_LEN_BINS; i += SCATTER_THREADS) l_cnt[i] = 0;
        __syncthreads();
        uint32_t len[SCATTER_PER], rank[SCATTER_PER];
        SpanRow r[SCATTER_PER];
#pragma unroll
        for (int k = 0; k < SCATTER_PER; k++) {
            /* the load has no condition (behind the end the last record is read again): 16 of them are in flight
               together; behind `if (i < n)` hipcc waited for each with vmcnt(0) */
            const uint64_t i = c0 + (uint64_t)k * SCATTER_THREADS + threadIdx.x;
            const sq_meta m = metas[i < n ? i : n - 1];
            len[k] = i < n ? (m.sequence_length < SQ_LEN_BINS - 1 ? m.sequence_length : SQ_LEN_BINS - 1) : 0xFFFFFFFFu;
            r[k].seq = m.record_start + m.sequence_offset;
            r[k].qual_delta = m.qualities_offset - m.sequence_offset;
            r[k].record = (uint32_t)i;
        }
#pragma unroll
        for (int k = 0; k < SCATTER_PER; k++)
            if (len[k] != 0xFFFFFFFFu) rank[k] = atomicAdd(&l_cnt[len[k]], 1u);
        __syncthreads();
        for (int i = threadIdx.x; i < SQ_LEN_BINS; i += SCATTER_THREADS)
            if (l_cnt[i]) l_base[i] = start[i] + atomicAdd(&cursor[i], l_cnt[i]);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SCATTER_PER; k++)
            if (len[k] != 0xFFFFFFFFu) rows[l_base[len[k]] + rank[k]] = r[k];
        __syncthreads();
    }
}
/* keys sorted ascending: longer[w] = how many reads are longer than w (w = 0 .. max_len) */
__global__ void k_span_longer(const uint32_t *keys, uint64_t n, uint32_t max_len, unsigned long long *longer)
{
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w > max_len) return;
    uint64_t lo = 0, hi = n;   /* first index whose read is not longer than w: key >= max_len - w */
    while (lo < hi) {
        const uint64_t mid = (lo + hi) / 2;
        if (keys[mid] < max_len - w) lo = mid + 1; else hi = mid;
    }
    longer[w] = lo;
}


/* ---- k_ptspan: PerTileQuality alone on a batch of one read length, the whole table in LDS ----
 * PerTileQuality_add_meta (:3123-3222) adds the error rate of every quality to the row of the
 * read's tile.  With reads of random tiles the round-1 kernels sort the batch by tile first and
 * gather.  Here the batch is streamed as it lies (spans of 16 records by LDS-DMA, the qualities
 * only) and the table of the workgroup -- [tiles + 1][positions] doubles, 123 KB for 96 tiles of
 * 150 positions -- lives in LDS: lane (h, pl) gets the qualities of rows 8 h .. 8 h + 7 at its
 * position by one transposing read per window, looks their error rates up and adds each to the
 * row of that read's tile with ds_add_f64 (32 consecutive doubles per half wave: no bank
 * conflict; the sums are order-free within the 1e-6 the module is checked to).  Reads that do
 * not count (no tile, or behind the first header that did not parse, :3137-3148) go to a row
 * nobody reads.  Merged into the device tables once per workgroup. */
struct PtSpanLds { uint32_t table, cnt, dma, meta, rows, slots; size_t total; };
constexpr uint32_t PTSPAN_META = SPAN_META_LDS + 64;   /* the first 32 bytes of a span's metas and the tile slots of its records */
/* U: read length = doubles per table row (the row nobody reads is a full 32 nw long: lanes of the
   last window that lie behind the end of the reads add there) */
__host__ __device__ inline PtSpanLds ptspan_lds_layout(int nw, uint32_t U, uint32_t nslots, int waves)
{
    PtSpanLds L;
    const uint32_t qpr = 2 * (uint32_t)nw + 1;
    uint32_t o = 256 * 8;                      /* error rates by quality byte, at LDS address 0 */
    L.table = o; o += (nslots * U + 32 * (uint32_t)nw) * 8;
    L.cnt = o; o += ((nslots + 1) * 4 + 15u) & ~15u;
    L.dma = o; o += ((16 * qpr + 63) / 64) * 64 * 4;
    L.meta = o; o += (uint32_t)waves * PTSPAN_META;   /* one buffer: what is needed of it is taken when the rows are requested */
    L.rows = o; o += (uint32_t)waves * 16 * 4;
    L.slots = o;
    L.total = (size_t)o + (size_t)waves * 2 * 16 * 16 * qpr;
    return L;
}

template <int NW>
__global__ void __launch_bounds__(1024) k_ptspan(PassParams P, uint32_t nslots)
{
    constexpr uint32_t QPR = 2 * NW + 1, ROWB = 16 * QPR, SLOT = SPAN_R * ROWB, ND = (SPAN_R * QPR + 63) / 64;
    extern __shared__ __align__(16) uint8_t smem[];
    const uint32_t U = P.uniform_len;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, T = blockDim.x, W = T >> 6;
    const PtSpanLds L = ptspan_lds_layout(NW, U, nslots, W);
    double *l_err = (double *)smem;                        /* [256] by raw quality byte */
    double *l_pt = (double *)(smem + L.table);             /* [nslots][U], then 32 NW doubles nobody reads */
    uint32_t *l_cnt = (uint32_t *)(smem + L.cnt);          /* [nslots + 1] reads per tile */
    uint32_t *l_dma = (uint32_t *)(smem + L.dma);
    uint32_t *l_rows = (uint32_t *)(smem + L.rows) + wave * SPAN_R;
    const uint32_t meta_base = lds_addr(smem + L.meta) + wave * PTSPAN_META;
    const uint32_t slot_base = lds_addr(smem + L.slots) + wave * 2 * SLOT;
    if (lds_addr(l_err) != 0) __builtin_trap(); /* quality byte << 3 is the address of its error rate */
    for (int i = tid; i < 256; i += T) {
        double e = __longlong_as_double(0x7FF8000000000000LL);   /* not a phred character: NaN, as in the other kernels */
        if (i >= 33 && i <= 33 + SQ_PHRED_MAX) e = __longlong_as_double((long long)c_error_rate_bits[i - 33]);
        else if (i >= 128 && i < 136) e = 0.0;
        l_err[i] = e;
    }
    for (uint32_t i = tid; i < nslots * U + 32 * NW; i += T) l_pt[i] = 0.0;
    for (uint32_t i = tid; i <= nslots; i += T) l_cnt[i] = 0;
    if (wave == 0) {
#pragma unroll
        for (int k = 0; k < (int)ND; k++) {   /* piece i = 64 k + lane: 16 bytes of the qualities of row i / QPR */
            const uint32_t i = 64 * k + lane, row = i / QPR, pir = i % QPR;
            l_dma[64 * k + lane] = (i < SPAN_R * QPR && pir < 2 * NW) ? (row * 4) | ((pir * 16) << 8) | 0x80000000u : 0;
        }
    }
    __syncthreads();

    const uint32_t q = (uint32_t)lane >> 2, c = (uint32_t)lane & 3;
    const uint32_t h = (uint32_t)lane >> 5, pl = (uint32_t)lane & 31;
    const uint32_t three = 3, one = 1;
    const uint32_t dummy = lds_addr(l_pt) + nslots * U * 8 + 8 * pl;
    const uint64_t nspans = P.n / SPAN_R, stride = (uint64_t)gridDim.x * W;
    uint64_t s = (uint64_t)blockIdx.x * W + wave;
    auto issue_meta = [&](uint64_t sp, uint32_t maddr) {
        if (lane < 2 * (int)SPAN_R)   /* bytes 0 .. 31 of every meta: record_start, qualities_offset */
            dma16((const uint8_t *)(P.metas + sp * SPAN_R + ((uint32_t)lane >> 1)) + 16 * (lane & 1), __builtin_amdgcn_readfirstlane(maddr));
        if (lane < 4)
            dma16((const uint8_t *)(P.pt_slot + sp * SPAN_R) + 16 * lane, __builtin_amdgcn_readfirstlane(maddr + SPAN_META_LDS));
    };
    /* the rows of span sp are requested; what the counting of that span needs of its metas (the
       tile rows of this lane's eight reads, the tile of read `lane` for the read counts) is
       taken along: the buffer is handed to the span after it */
    uint32_t toff_n[8], cnt_n = 0;
    auto issue = [&](uint32_t slot_addr, uint32_t maddr, uint64_t sp) {
        const uint32_t ma = maddr + 32 * q;
        const unsigned long long m_rs = *(SQ_LDS const unsigned long long *)(uintptr_t)ma; /* record_start */
        const uint32_t m_qo = lds_u32(ma + 20);                                            /* qualities_offset */
        const unsigned long long base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(m_rs >> 32)) << 32) |
                                        (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)m_rs);
        if (c == 0) l_rows[q] = (uint32_t)(m_rs - base) + m_qo;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int32_t sl = (int32_t)lds_u32(maddr + SPAN_META_LDS + 4 * (8 * h + k));
            const bool on = sl >= 0 && P.first_read_index + sp * SPAN_R + 8 * h + k < P.pt_first_bad;
            toff_n[k] = on ? lds_addr(l_pt) + (uint32_t)sl * U * 8 + 8 * pl : dummy;
        }
        {
            const int32_t sl = (int32_t)lds_u32(maddr + SPAN_META_LDS + 4 * (lane & 15));
            const bool on = sl >= 0 && P.first_read_index + sp * SPAN_R + (lane & 15) < P.pt_first_bad;
            cnt_n = lds_addr(l_cnt) + 4 * (on ? (uint32_t)sl : nslots);
        }
        const uint8_t *g0 = P.buf + base;
        uint32_t pk[ND];
        int32_t rr[ND];
#pragma unroll
        for (int k = 0; k < (int)ND; k++) pk[k] = l_dma[64 * k + lane];
#pragma unroll
        for (int k = 0; k < (int)ND; k++) rr[k] = (int32_t)lds_u32(lds_addr(l_rows) + (pk[k] & 0xFFu));
#pragma unroll
        for (int k = 0; k < (int)ND; k++)
            if ((int32_t)pk[k] < 0)
                dma16(g0 + (long long)rr[k] + ((pk[k] >> 8) & 0xFFFu), __builtin_amdgcn_readfirstlane(slot_addr + 1024 * k));
    };

    uint32_t toff[8], cnt_a = 0;
    int cur = 0;
    if (s < nspans) {
        issue_meta(s, meta_base);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        issue(slot_base, meta_base, s);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (s + stride < nspans) issue_meta(s + stride, meta_base);
    }
    while (s < nspans) {
#pragma unroll
        for (int k = 0; k < 8; k++) toff[k] = toff_n[k];
        cnt_a = cnt_n;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (s + stride < nspans) {
            issue(slot_base + (cur ^ 1) * SLOT, meta_base, s + stride);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   /* the metas have been read: the next ones may land on them */
            if (s + 2 * stride < nspans) issue_meta(s + 2 * stride, meta_base);
        }
        if (lane < (int)SPAN_R) lds_add(cnt_a, one);
        const uint32_t sa = slot_base + cur * SLOT;
        /* transposing reads: lane 2 q' + p of a group of 16 hands in row 8 h + q', bytes 8 p .. 8 p + 7 */
        const uint32_t trb = sa + (8 * h + (((uint32_t)lane & 15) >> 1)) * ROWB + 16 * (((uint32_t)lane >> 4) & 1) + 8 * ((uint32_t)lane & 1);
        /* all windows' bytes first; then the error rates of window w + 1 are on their way while
           those of window w are added (LDS answers in order: behind the 8 adds of the window
           before, the 8 lookups in front of them have arrived) */
        sq_u32x2 t[NW];
        static_for<0, NW>([&](auto wc) { constexpr int w = decltype(wc)::value; t[w] = rd_tr8<32 * w>(trb); });
        wait_lgkm<0>();
        static_for<0, NW>([&](auto wc) { tie2(t[decltype(wc)::value]); });
        double e[2][8];
        static_for<0, 8>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            e[0][k] = rd_f64(shl3_byte_of<k % 4>(k < 4 ? t[0].x : t[0].y, three));
        });
        static_for<0, NW>([&](auto wc) {
            constexpr int w = decltype(wc)::value;
            if constexpr (w == 0) wait_lgkm<0>(); else wait_lgkm<8>();
            static_for<0, 8>([&](auto kc) { tie_f64(e[w & 1][decltype(kc)::value]); });
            if constexpr (w + 1 < NW)
                static_for<0, 8>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    e[(w + 1) & 1][k] = rd_f64(shl3_byte_of<k % 4>(k < 4 ? t[w + 1].x : t[w + 1].y, three));
                });
            const bool inside = 32 * w + pl < U;   /* the last window reaches behind the end of the reads */
            static_for<0, 8>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                if constexpr (w + 1 < NW) add_f64_lds(toff[k] + 256 * w, e[w & 1][k]);
                else add_f64_lds((inside ? toff[k] : dummy) + 256 * w, e[w & 1][k]);
            });
        });
        cur ^= 1;
        s += stride;
    }
    __syncthreads();
    for (uint32_t i = tid; i < nslots; i += T)
        if (l_cnt[i]) atomicAdd(&P.pt_len_counts[(uint64_t)i * P.pt_cap + (U - 1)], (unsigned long long)l_cnt[i]);
    for (uint32_t i = tid; i < nslots * U; i += T) {
        const double v = l_pt[i];
        if (v != 0.0) unsafeAtomicAdd(&P.pt_errors[(uint64_t)(i / U) * P.pt_cap + i % U], v);
    }
}

template <int NW>
int launch_ptspan(sq_ctx *ctx, const PassParams &P, uint32_t nslots, int waves, size_t lds, int grid)
{
    static bool attr = false;
    if (!attr) {
        SQ_HIP(hipFuncSetAttribute((const void *)k_ptspan<NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    sq_route(ctx, "k_ptspan<%d>", NW);
    hipLaunchKernelGGL((k_ptspan<NW>), dim3(grid), dim3(waves * 64), lds, ctx->stream, P, nslots);
    SQ_HIP(hipGetLastError());
    return SQ_OK;
}


/* ---- k_isz_span: the overlap scan of InsertSizeMetrics on pairs of one read length each ----
 * calculate_insert_size (_qcmodule.c:5667-5707) slides a 16-byte window over read 1 and compares
 * it with the reverse complements of the first and of the last 16 bases of read 2.  A lane per
 * pair streaming its own read (k_insert_size) is the least efficient way to ask memory: 5.6 ms
 * per 25 M pairs for 1.4 ms of arithmetic.  Here a wave takes 16 pairs per span: the sequences
 * of read 1 come by LDS-DMA as in k_span (16-byte pieces from any alignment), of read 2 only the
 * two ends; four lanes share a pair, lane c scans the windows [8 NW c, 8 NW (c + 1)) from LDS
 * (all its bytes are read up front: the scan itself is registers only), the first match of the
 * quad in window order is the pair's.  The adapter remainders of the pairs that have one
 * (InsertSizeMetrics_add_adapter :5570-5611; they need the device hash tables) are left to
 * k_isz_adapters: this kernel only notes the insert size of such a pair in `results`. */
struct IszSpanLds { uint32_t sizes, dma, meta, rows, ends, slots; size_t total; };
__host__ __device__ inline IszSpanLds isz_span_lds_layout(int nw, uint32_t lds_sizes, int waves)
{
    IszSpanLds L;
    const uint32_t qpr = 2 * (uint32_t)nw + 1;
    uint32_t o = 16;                                  /* the workgroup's largest insert size */
    L.sizes = o; o += (lds_sizes * 4 + 15u) & ~15u;
    L.dma = o; o += ((16 * qpr + 63) / 64) * 64 * 4;
    L.meta = o; o += (uint32_t)waves * 2 * 2 * SPAN_META_BYTES;
    L.rows = o; o += (uint32_t)waves * 16 * 4;
    L.ends = o; o += (uint32_t)waves * 2 * 512;       /* per row the first and the last 16 bases of read 2 */
    L.slots = o;
    L.total = (size_t)o + (size_t)waves * 2 * 16 * 16 * qpr;
    return L;
}

__device__ __forceinline__ uint32_t isz_nonzero_bytes_of(uint32_t v)
{
    return (uint32_t)__popc((((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v) & 0x80808080u);
}

template <int NW>
__global__ void __launch_bounds__(1024) k_isz_span(IszSpanParams P)
{
    constexpr uint32_t QPR = 2 * NW + 1, ROWB = 16 * QPR, SLOT = SPAN_R * ROWB, ND = (SPAN_R * QPR + 63) / 64, WQ = 8 * NW;
    extern __shared__ __align__(16) uint8_t smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, T = blockDim.x, W = T >> 6;
    const IszSpanLds L = isz_span_lds_layout(NW, P.lds_sizes, W);
    uint32_t *l_max = (uint32_t *)smem;
    uint32_t *l_sizes = (uint32_t *)(smem + L.sizes);
    uint32_t *l_dma = (uint32_t *)(smem + L.dma);
    uint32_t *l_rows = (uint32_t *)(smem + L.rows) + wave * SPAN_R;
    const uint32_t meta_base = lds_addr(smem + L.meta) + wave * 4 * SPAN_META_BYTES;   /* [2 buffers][read 1, read 2] */
    const uint32_t ends_base = lds_addr(smem + L.ends) + wave * 2 * 512;
    const uint32_t slot_base = lds_addr(smem + L.slots) + wave * 2 * SLOT;
    for (uint32_t i = tid; i < P.lds_sizes; i += T) l_sizes[i] = 0;
    if (tid == 0) *l_max = 0;
    if (wave == 0) {
#pragma unroll
        for (int k = 0; k < (int)ND; k++) {   /* piece i = 64 k + lane: 16 bytes of the sequence of row i / QPR */
            const uint32_t i = 64 * k + lane, row = i / QPR, pir = i % QPR;
            l_dma[64 * k + lane] = (i < SPAN_R * QPR && 16 * pir < P.L1) ? (row * 4) | ((pir * 16) << 8) | 0x80000000u : 0;
        }
    }
    __syncthreads();
    const uint32_t q = (uint32_t)lane >> 2, c = (uint32_t)lane & 3;
    const uint64_t nspans = P.n / SPAN_R, stride = (uint64_t)gridDim.x * W;
    uint64_t s = (uint64_t)blockIdx.x * W + wave;
    auto issue_meta = [&](uint64_t sp, uint32_t maddr) {
        if (lane < (int)(SPAN_META_BYTES / 16)) {
            dma16((const uint8_t *)(P.metas1 + sp * SPAN_R) + 16 * lane, __builtin_amdgcn_readfirstlane(maddr));
            dma16((const uint8_t *)(P.metas2 + sp * SPAN_R) + 16 * lane, __builtin_amdgcn_readfirstlane(maddr + SPAN_META_BYTES));
        }
    };
    auto issue = [&](uint32_t slot_addr, uint32_t maddr, uint32_t ends_addr) {
        const uint32_t ma = maddr + 40 * q;
        const unsigned long long rs1 = *(SQ_LDS const unsigned long long *)(uintptr_t)ma;
        const uint32_t so1 = lds_u32(ma + 12);
        const unsigned long long base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(rs1 >> 32)) << 32) |
                                        (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)rs1);
        if (c == 0) l_rows[q] = (uint32_t)(rs1 - base) + so1;
        /* the ends of read 2: lane (row, e) with row = lane / 2 */
        if (lane < 32) {
            const uint32_t m2 = maddr + SPAN_META_BYTES + 40 * ((uint32_t)lane >> 1);
            const unsigned long long rs2 = *(SQ_LDS const unsigned long long *)(uintptr_t)m2;
            const uint32_t so2 = lds_u32(m2 + 12);
            dma16(P.buf2 + rs2 + so2 + ((lane & 1) ? P.L2 - 16 : 0), __builtin_amdgcn_readfirstlane(ends_addr));
        }
        const uint8_t *g0 = P.buf1 + base;
        uint32_t pk[ND];
        int32_t rr[ND];
#pragma unroll
        for (int k = 0; k < (int)ND; k++) pk[k] = l_dma[64 * k + lane];
#pragma unroll
        for (int k = 0; k < (int)ND; k++) rr[k] = (int32_t)lds_u32(lds_addr(l_rows) + (pk[k] & 0xFFu));
#pragma unroll
        for (int k = 0; k < (int)ND; k++)
            if ((int32_t)pk[k] < 0)
                dma16(g0 + (long long)rr[k] + ((pk[k] >> 8) & 0xFFFu), __builtin_amdgcn_readfirstlane(slot_addr + 1024 * k));
    };

    const uint32_t last = P.L1 - 16, UP4 = 0xDFDFDFDFu;
    uint32_t local_max = 0;
    int cur = 0;
    if (s < nspans) {
        issue_meta(s, meta_base);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        issue(slot_base, meta_base, ends_base);
        if (s + stride < nspans) issue_meta(s + stride, meta_base + 2 * SPAN_META_BYTES);
    }
    while (s < nspans) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (s + stride < nspans) {
            issue(slot_base + (cur ^ 1) * SLOT, meta_base + (cur ^ 1) * 2 * SPAN_META_BYTES, ends_base + (cur ^ 1) * 512);
            if (s + 2 * stride < nspans) issue_meta(s + 2 * stride, meta_base + cur * 2 * SPAN_META_BYTES);
        }
        /* the needles: lane c makes one of the four halves (h_lo, h_hi, t_lo, t_hi), the quad shares them */
        const uint32_t ea = ends_base + cur * 512 + 32 * q + ((c & 1) ? 0 : 8) + ((c & 2) ? 16 : 0);
        const unsigned long long mine = pair_revcomp8(*(SQ_LDS const unsigned long long *)(uintptr_t)ea);
        const uint32_t m_lo = (uint32_t)mine, m_hi = (uint32_t)(mine >> 32);
        const uint32_t hl = quad_bcast<0x00>(m_lo), hl2 = quad_bcast<0x00>(m_hi), hh = quad_bcast<0x55>(m_lo), hh2 = quad_bcast<0x55>(m_hi);
        const uint32_t tl = quad_bcast<0xAA>(m_lo), tl2 = quad_bcast<0xAA>(m_hi), th = quad_bcast<0xFF>(m_lo), th2 = quad_bcast<0xFF>(m_hi);
        /* This lane's windows of read 1 start at bytes [WQ c, WQ c + WQ): the prefilter names the positions where a needle
           half can match (pair_scan_candidates, sq_span_kernel.h: the same scan as in read 1's pass of the paired route); the
           rare candidates are looked at below, in window order. */
        const uint32_t ra = slot_base + cur * SLOT + q * ROWB + WQ * c;
        uint32_t result = 0;
        unsigned long long cand = pair_scan_candidates<NW>(ra, hl, hl2, hh, hh2, tl, tl2, th, th2);   /* bit 63 - k: position WQ c + k */
        while (cand) {   /* :5695-5704: a half matches case-insensitively, then at most one raw byte of the 16 may differ */
            const uint32_t j = (uint32_t)__clzll((long long)cand), i = WQ * c + j;
            cand &= ~(0x8000000000000000ull >> j);
            if (i > last) break;
            uint32_t b[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const uint32_t a = ra + j + 4 * t;   /* any alignment */
                b[t] = lds_u8(a) | (lds_u8(a + 1) << 8) | (lds_u8(a + 2) << 16) | (lds_u8(a + 3) << 24);
            }
            const uint32_t u0 = b[0] & UP4, u1 = b[1] & UP4, u2 = b[2] & UP4, u3 = b[3] & UP4;
            if ((u0 == hl && u1 == hl2) || (u2 == hh && u3 == hh2)) {
                const uint32_t d = isz_nonzero_bytes_of(b[0] ^ hl) + isz_nonzero_bytes_of(b[1] ^ hl2) +
                                   isz_nonzero_bytes_of(b[2] ^ hh) + isz_nonzero_bytes_of(b[3] ^ hh2);
                if (d <= 1) { result = i + 16; break; }
            }
            if ((u0 == tl && u1 == tl2) || (u2 == th && u3 == th2)) {
                const uint32_t d = isz_nonzero_bytes_of(b[0] ^ tl) + isz_nonzero_bytes_of(b[1] ^ tl2) +
                                   isz_nonzero_bytes_of(b[2] ^ th) + isz_nonzero_bytes_of(b[3] ^ th2);
                if (d <= 1) { result = i + P.L2; break; }
            }
        }
        /* the first match in window order: the lowest quarter that has one */
        const uint32_t r0 = quad_bcast<0x00>(result), r1 = quad_bcast<0x55>(result), r2 = quad_bcast<0xAA>(result), r3 = quad_bcast<0xFF>(result);
        result = r0 ? r0 : r1 ? r1 : r2 ? r2 : r3;
        if (c == 0) {
            if (result < P.lds_sizes) lds_add(lds_addr(l_sizes + result), 1u);
            else atomicAdd(&P.insert_sizes[result], 1ULL);
            local_max = max(local_max, result);
            const uint32_t note = (result && (P.L1 > result || P.L2 > result)) ? result : 0;
            uint32_t *dst = P.results + s * SPAN_R + q;
            asm volatile("global_store_dword %0, %1, off" :: "v"(dst), "v"(note) : "memory");
        }
        cur ^= 1;
        s += stride;
    }
    if (local_max) atomicMax(l_max, local_max);
    __syncthreads();
    if (tid == 0 && *l_max) atomicMax(P.max_insert, (unsigned long long)*l_max);
    for (uint32_t i = tid; i < P.lds_sizes; i += T)
        if (l_sizes[i]) atomicAdd(&P.insert_sizes[i], (unsigned long long)l_sizes[i]);
}

template <int NW>
int launch_isz_span(sq_ctx *ctx, const IszSpanParams &P, int waves, size_t lds, int grid)
{
    static bool attr = false;
    if (!attr) {
        SQ_HIP(hipFuncSetAttribute((const void *)k_isz_span<NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    sq_route(ctx, "k_isz_span<%d>", NW);
    hipLaunchKernelGGL((k_isz_span<NW>), dim3(grid), dim3(waves * 64), lds, ctx->stream, P);
    SQ_HIP(hipGetLastError());
    return SQ_OK;
}

} // namespace

namespace {

/* ---- long reads (longer than k_span's rows): segments of 128 positions through k_span<LONG> ----
 * The reads of the batch come sorted by length, longest first (P.order).  What a read needs as a
 * whole -- the four f64 chains of its error sum in the reference's order, its GC and phred_scores
 * bins (:2051-2137) -- is k_read_sums' work (sq_qc.hip); everything per position runs here, over
 * ROWS that are segments of reads: segment j of the sorted reads 0 .. n_j - 1 (the ones longer than
 * 128 j) is a stretch of spans of one positional window, so that a workgroup's LDS histograms are
 * those of positions [128 j, 128 j + 128) and go to the device tables when j changes.  The
 * automaton is restarted in front of every segment (the bases in front come along), its matches
 * are candidates for the first occurrence per read and adapter (P.long_first, k_adapter_first).
 * The end-anchored tables (the last `end_anchor_length` positions of every read, :1971-1972,
 * :2115-2124) have a small kernel of their own.  Reference: QCMetrics_add_meta :1966-2139,
 * AdapterCounter_add_meta :2786-2823. */


__global__ void k_long_rows(const sq_meta *metas, const uint32_t *order, uint64_t n, SpanRow *rows)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t rec = order[i];
        const sq_meta m = metas[rec];
        SpanRow r;
        r.seq = (m.record_start + m.sequence_offset) | ((uint64_t)m.sequence_length << 40);
        r.qual_delta = m.qualities_offset - m.sequence_offset;
        r.record = rec;
        rows[i] = r;
    }
}
/* rows sorted by length, longest first: counts[j] = how many are longer than j * seg */
__global__ void k_long_counts(const SpanRow *rows, uint64_t n, uint32_t seg, uint32_t n_segs, unsigned long long *counts)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_segs) return;
    const uint64_t limit = (uint64_t)j * seg;
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) / 2;
        if ((rows[mid].seq >> 40) > limit) lo = mid + 1; else hi = mid;
    }
    counts[j] = lo;
}
/* the end-anchored tables of a batch: item i = (read i / ea_len, slot i % ea_len of the tables) */
constexpr uint32_t EA_READS = 4;   /* reads whose loads a wave of k_long_ea has in flight */
__global__ void __launch_bounds__(256) k_long_ea(const uint8_t *buf, const sq_meta *metas, uint64_t n, uint32_t ea_len,
                                                 unsigned long long *ea_base, unsigned long long *ea_phred)
{
    extern __shared__ uint32_t l_ea[];   /* [5 + 12][ea_len] */
    for (uint32_t i = threadIdx.x; i < 17 * ea_len; i += blockDim.x) l_ea[i] = 0;
    __syncthreads();
    /* 64 reads per wave and round: lane j fetches the meta of read j, then the wave walks the reads, a lane per slot
       (round 3: an item per thread paid a 64-bit division and a dependent meta load per slot) */
    const uint32_t lane = threadIdx.x & 63, waves = blockDim.x >> 6;
    for (uint64_t r0 = ((uint64_t)blockIdx.x * waves + (threadIdx.x >> 6)) * 64; r0 < n; r0 += (uint64_t)gridDim.x * waves * 64) {
        unsigned long long so = 0, qo = 0;   /* where slot 0 would lie (in front of the read when it is shorter than ea_len: never touched) */
        uint32_t first = ea_len;
        if (r0 + lane < n) {
            const sq_meta m = metas[r0 + lane];
            const uint32_t L = m.sequence_length;
            first = ea_len - min(ea_len, L);   /* :1971-1972: right aligned */
            so = m.record_start + m.sequence_offset + L - ea_len;
            qo = m.record_start + m.qualities_offset + L - ea_len;
        }
        const uint32_t cnt = (uint32_t)min((uint64_t)64, n - r0);
        /* four reads at a time, 128 slots each, every load without a condition (a slot that does not exist reads
           the read's last byte again and counts nothing): hipcc issues the 16 loads and waits for them by count;
           behind `if (e < ea_len)` it waited for every read's loads with vmcnt(0), one memory latency per read */
        for (uint32_t e0 = 0; e0 < ea_len; e0 += 128)
            for (uint32_t j0 = 0; j0 < cnt; j0 += EA_READS) {
                uint32_t sb[EA_READS][2], qb[EA_READS][2], at[EA_READS][2];
#pragma unroll
                for (uint32_t u = 0; u < EA_READS; u++) {
                    const uint32_t j = min(j0 + u, cnt - 1);
                    const unsigned long long sj = __shfl(so, (int)j), qj = __shfl(qo, (int)j);
                    const uint32_t fj = j0 + u < cnt ? __shfl(first, (int)j) : ea_len;   /* (a repeated read counts nothing) */
#pragma unroll
                    for (uint32_t h = 0; h < 2; h++) {
                        const uint32_t e = e0 + 64 * h + lane;
                        const bool on = e >= fj && e < ea_len;
                        const uint32_t ec = on ? e : ea_len - 1;
                        sb[u][h] = buf[sj + ec];
                        qb[u][h] = buf[qj + ec];
                        at[u][h] = on ? e : 0xFFFFFFFFu;
                    }
                }
#pragma unroll
                for (uint32_t u = 0; u < EA_READS; u++)
#pragma unroll
                    for (uint32_t h = 0; h < 2; h++)
                        if (at[u][h] != 0xFFFFFFFFu) {
                            atomicAdd(&l_ea[sq_base_class((uint8_t)sb[u][h]) * ea_len + at[u][h]], 1u);
                            atomicAdd(&l_ea[(5 + (min(qb[u][h] - 33u, 47u) >> 2)) * ea_len + at[u][h]], 1u);
                        }
            }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 17 * ea_len; i += blockDim.x) {
        const uint32_t v = l_ea[i], row = i / ea_len, e = i % ea_len;
        if (!v) continue;
        if (row < 5) atomicAdd(&ea_base[(uint64_t)e * 5 + row], (unsigned long long)v);
        else atomicAdd(&ea_phred[(uint64_t)e * 12 + (row - 5)], (unsigned long long)v);
    }
}

/* QCMetrics' GC histogram from the counts k_span<LONG> summed per read (:2051-2060: reads without any A/C/G/T have no bin) */
__global__ void __launch_bounds__(256) k_long_gc_bins(const unsigned int *gc, uint64_t n, unsigned long long *qc_gc)
{
    __shared__ uint32_t l_gc[104];
    for (int i = threadIdx.x; i < 104; i += 256) l_gc[i] = 0;
    __syncthreads();
    for (uint64_t r = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; r < n; r += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t g = gc[2 * r], a = gc[2 * r + 1];
        if (a > 0) atomicAdd(&l_gc[(uint32_t)round((double)g * 100.0 / (double)a)], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 101; i += 256)
        if (l_gc[i]) atomicAdd(&qc_gc[i], (unsigned long long)l_gc[i]);
}

} // namespace

/* QCMetrics' positional tables, its end-anchored tables, its GC histogram and AdapterCounter's candidates (P.long_first,
 * [records][n_ad], preset to ~0) for a batch of long reads sorted by length (P.order).  *done = records
 * covered: all or none (0: the kernel does not take this pass; nothing has been counted). */
template <int NW>
int launch_long(sq_ctx *ctx, const PassParams &C0, bool ad, uint32_t n_ad, int waves, size_t lds, int grid)
{
    PassParams C = C0;
    static bool attr = false;
    if (!attr) {
        SQ_HIP(hipFuncSetAttribute((const void *)k_span<NW, true, true, SPAN_W4, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        SQ_HIP(hipFuncSetAttribute((const void *)k_span<NW, false, true, SPAN_W4, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    sq_route(ctx, "k_span<%d,%s,long>", NW, ad ? "AD" : "QC");
    if (ad) hipLaunchKernelGGL((k_span<NW, true, true, SPAN_W4, true, true>), dim3(grid), dim3(waves * 64), lds, ctx->stream, C, n_ad);
    else hipLaunchKernelGGL((k_span<NW, false, true, SPAN_W4, true, true>), dim3(grid), dim3(waves * 64), lds, ctx->stream, C, n_ad);
    SQ_HIP(hipGetLastError());
    return SQ_OK;
}

static int long_waves(const PassParams &P, bool ad, uint32_t n_ad, uint32_t max_len, int *nw_out)
{
    const uint64_t n = P.n;
    const int nw = 8;   /* windows of 32 positions per segment (4 and 6 measured slower in round 3) */
    const uint32_t LSEG = 32 * (uint32_t)nw;
    *nw_out = nw;
    if (!P.order || n < SPAN_R || n >= (1ull << 31) || !max_len || max_len >= (1u << 24) || P.buf_len >= (1ull << 40) || P.ea_len > 512) return 0;
    if (ad && (SPAN_STATES(P) > SPAN_DFA_MAX_STATES || n_ad > 64 || (P.ad_maxlen + 2) / 4 > SPAN_W4)) return 0;
    int waves = span_max_waves(nw, true, true, true);
    while (waves >= 4 && span_lds_layout(nw, LSEG, ad ? SPAN_STATES(P) : 0, ad ? n_ad : 0, 0, waves, true, true, true).total > 160 * 1024) waves -= 2;
    if (waves < 4) return 0;
    return waves;
}

/* Where every workgroup's stretch of consecutive spans begins ([grid + 1] span numbers): shares of equal cost, a span
 * costing 1 and every segment a workgroup meets K (a share that ends inside a segment makes the next workgroup meet it
 * too).  nspans[j] = spans of segment j, in span order. */
std::vector<uint32_t> span_cost_shares(const std::vector<uint32_t> &nspans, int grid, double K)
{
    uint64_t spans = 0;
    double total = K * grid;
    for (uint32_t n : nspans) { spans += n; total += (double)n + K; }
    std::vector<uint32_t> bounds((size_t)grid + 1, (uint32_t)spans);
    const double share = total / grid;
    bounds[0] = 0;
    int wg = 1;
    double acc = 0;   /* cost handed out so far */
    uint32_t span0 = 0;
    for (uint32_t n : nspans) {
        double left = (double)n;   /* spans of this segment not handed out yet */
        uint32_t at = span0;
        acc += K;
        while (wg < grid && acc + left > share * wg) {
            const double take = std::max(0.0, share * wg - acc);
            const uint32_t t = (uint32_t)std::min<double>(left, std::floor(take));
            at += t; left -= t; acc += t;
            bounds[wg++] = at;
            acc += K;   /* the next workgroup meets this segment too */
            if (left <= 0) break;
        }
        acc += left;
        span0 += n;
    }
    for (size_t i = 1; i < bounds.size(); i++) bounds[i] = std::max(bounds[i], bounds[i - 1]);
    bounds[grid] = (uint32_t)spans;
    return bounds;
}

SQ_EXPORT void sq_span_cost_shares(const uint32_t *nspans, size_t n, int grid, int cost, uint32_t *bounds)
{
    const std::vector<uint32_t> b = span_cost_shares(std::vector<uint32_t>(nspans, nspans + n), grid, (double)cost);
    memcpy(bounds, b.data(), b.size() * sizeof(uint32_t));
}

/* does k_span<LONG> take this batch (sq_span_launch_long may still decline: memory) */
bool sq_span_long_takes(const PassParams &P, bool ad, uint32_t n_ad, uint32_t max_len)
{
    int nw;
    return long_waves(P, ad, n_ad, max_len, &nw) > 0;
}

static int sq_span_long_followups(sq_ctx *ctx, const PassParams &P);

int sq_span_launch_long(sq_ctx *ctx, const PassParams &P, bool ad, uint32_t n_ad, uint32_t max_len, uint64_t *done)
{
    *done = 0;
    const uint64_t n = P.n;
    int nw = 8;
    const int waves = long_waves(P, ad, n_ad, max_len, &nw);
    if (!waves) return SQ_OK;
    const uint32_t LSEG = 32 * (uint32_t)nw;
    const uint32_t n_segs = (max_len + LSEG - 1) / LSEG;
    SpanRow *rows = (SpanRow *)sq_scratch(ctx, 15, n * sizeof(SpanRow));
    unsigned long long *d_counts = (unsigned long long *)sq_scratch(ctx, 23, (size_t)n_segs * 8);   /* a slot of its own: 3 holds P.order, which further adapter groups still walk */
    if (!rows || !d_counts) { sq_set_error("out of device memory for the segments of long reads"); return SQ_ERR_MEMORY; }
    hipLaunchKernelGGL(k_long_rows, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 8192)), dim3(256), 0, ctx->stream, P.metas, P.order, n, rows);
    hipLaunchKernelGGL(k_long_counts, dim3((n_segs + 255) / 256), dim3(256), 0, ctx->stream, rows, n, LSEG, n_segs, d_counts);
    std::vector<uint64_t> counts(n_segs);
    SQ_HIP(hipMemcpyAsync(counts.data(), d_counts, (size_t)n_segs * 8, hipMemcpyDeviceToHost, ctx->stream));
    SQ_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<SpanSeg> segs;
    uint64_t spans = 0;
    /* The order of the stretches: segment j of ALL reads, then segment j + 1 of all reads (one stretch per j).  (Walking the sorted
       reads in blocks, all segments of a block before the next block -- so that the line two segments of a read share is still
       cached -- measured 2-38 % SLOWER at blocks of 16 K-1 K reads: round 5, scripts/exp_r5.sh; removed.) */
    const uint64_t block = n;
    for (uint64_t b0 = 0; b0 < n; b0 += block) {
        for (uint32_t j = 0; j < n_segs && counts[j] > b0; j++) {
            const uint64_t count = std::min<uint64_t>(block, counts[j] - b0);   /* reads of the block that are longer than 256 j */
            SpanSeg g{};
            g.span0 = (uint32_t)spans;
            g.nspans = (uint32_t)((count + SPAN_R - 1) / SPAN_R);
            g.U = LSEG;
            g.last_rows = (uint32_t)(count - (uint64_t)(g.nspans - 1) * SPAN_R);
            g.first = (uint32_t)b0;
            g.pos_base = j * LSEG;
            segs.push_back(g);
            spans += g.nspans;
            if (spans >= (1ull << 32)) return SQ_OK;
        }
    }
    if (segs.empty()) return SQ_OK;
    SpanSeg *d_segs = (SpanSeg *)sq_scratch(ctx, 14, segs.size() * sizeof(SpanSeg));
    if (!d_segs) { sq_set_error("out of device memory for the segments of long reads"); return SQ_ERR_MEMORY; }
    SQ_HIP(hipMemcpyAsync(d_segs, segs.data(), segs.size() * sizeof(SpanSeg), hipMemcpyHostToDevice, ctx->stream));
    PassParams C = P;
    C.uniform_len = 0;
    C.ad_lds = 0;
    C.span_segs = d_segs;
    C.span_nsegs = (uint32_t)segs.size();
    C.span_total = (uint32_t)spans;
    C.span_rows = rows;
    const size_t lds = span_lds_layout(nw, LSEG, ad ? SPAN_STATES(P) : 0, ad ? n_ad : 0, 0, waves, true, true, true).total;
    const int grid = (int)std::max<uint64_t>(1, std::min<uint64_t>((spans + waves / 2 - 1) / (waves / 2), (uint64_t)ctx->num_cus));
    /* Where every workgroup's stretch of the spans begins.  Equal numbers of spans are not equal work: a
       workgroup that meets a new segment flushes, merges its histograms and restarts its DMA pipeline (what a
       dozen spans cost it), and the segments far into the reads hold a few spans each -- with equal shares
       the last workgroup of config 4 went through 155 of them, the others through 1-18, and the launch waited
       for it: 9.65 ms, 7.85 with shares of equal COST (spans + 16 per segment met; 12 and
       more all measure the same). */
    std::vector<uint32_t> nspans_of;
    for (const SpanSeg &g : segs) nspans_of.push_back(g.nspans);
    const std::vector<uint32_t> bounds = span_cost_shares(nspans_of, grid, 16.0);
    uint32_t *d_bounds = (uint32_t *)sq_scratch(ctx, 22, bounds.size() * 4);
    if (!d_bounds) { sq_set_error("out of device memory for the segments of long reads"); return SQ_ERR_MEMORY; }
    SQ_HIP(hipMemcpyAsync(d_bounds, bounds.data(), bounds.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    C.span_bounds = d_bounds;
    static bool attr = false;
    if (!attr) {
        SQ_HIP(hipFuncSetAttribute((const void *)k_long_ea, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        attr = true;
    }
    C.long_gc = (unsigned int *)sq_scratch(ctx, 16, n * 8);
    if (!C.long_gc) { sq_set_error("out of device memory for the segments of long reads"); return SQ_ERR_MEMORY; }
    SQ_HIP(hipMemsetAsync(C.long_gc, 0, n * 8, ctx->stream));
    int rc = launch_long<8>(ctx, C, ad, n_ad, waves, lds, grid);
    if (rc) return rc;
    rc = sq_span_long_followups(ctx, P);
    if (rc) return rc;
    SQ_HIP(hipGetLastError());
    SQ_HIP(hipStreamSynchronize(ctx->stream));   /* the host vector of the segments goes out of scope */
    *done = n;
    return SQ_OK;
}

/* behind k_span<LONG>: the G/C histogram from the reads' counts (scratch 16, where the pass left them) and the end-anchored tables */
static int sq_span_long_followups(sq_ctx *ctx, const PassParams &P)
{
    const uint64_t n = P.n;
    unsigned int *long_gc = (unsigned int *)sq_scratch(ctx, 16, n * 8);
    if (!long_gc) { sq_set_error("out of device memory for the segments of long reads"); return SQ_ERR_MEMORY; }
    hipLaunchKernelGGL(k_long_gc_bins, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, (uint64_t)ctx->num_cus * 4)), dim3(256), 0, ctx->stream,
                       long_gc, n, P.qc_gc);
    if (P.ea_len)
        hipLaunchKernelGGL(k_long_ea, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, (uint64_t)ctx->num_cus * 8)), dim3(256),
                           (size_t)17 * P.ea_len * 4, ctx->stream, P.buf, P.metas, n, P.ea_len, P.qc_ea_base, P.qc_ea_phred);
    SQ_HIP(hipGetLastError());
    return SQ_OK;
}

/* Runs k_span over the first 16 * (n / 16) records of the pass described by P (QCMetrics, with
 * AdapterCounter when `ad`).  *done = records covered, 0 when the kernel does not take this
 * pass (read length, automaton size, LDS). */
int sq_span_launch(sq_ctx *ctx, const PassParams &P, bool ad, uint32_t n_ad, uint64_t *done)
{
    *done = 0;
    const uint32_t U = P.uniform_len;
    if (!U || U > 32 * SPAN_NW_MAX || P.n < SPAN_R) return SQ_OK;
    const int nw = (int)((U + 31) / 32);
    /* up to 64 positions with adapters: k_wide (64 rows x 64 bytes per visit: a short read is one or two visits) is 22 % ahead at 50
       bases (766 against 628 Gbases/s on 25 M reads, scripts/bench_len_dev.py, round 5), 2 % at 75, level from 100 on */
    if (ad && nw <= 2 && !sq_knobs().span_short && !span_needs_w6(P, ad)) return SQ_OK;   /* (SQ_SPAN_SHORT=1: k_span all the same) */
    /* (225-256 positions with adapters went to k_wide until round 5: the 8-window build ran 8 waves of ~ 190 registers and was 3-11 %
       behind; at 98 registers it runs 12 and is 10-14 % ahead -- 1038 / 1038 / 1107 against 928 / 943 / 969 Gbases/s at 240 / 250 / 256
       bases, scripts/exp_len2.sh -- on 1.2 x the algorithmic bytes instead of k_wide's 2.85 x) */
    /* QCMetrics alone: a wave per stream from 6 windows on (the chain reads 16 bytes at a time and the builds hold ~ 100 registers since
       round 5: 12 waves a CU against the 8 of one wave for both streams -- 1218 / 1290 against 1066 / 1145 Gbases/s at 200 / 250 bases;
       up to 5 windows both run 16 waves and one wave for both streams is 2-4 % ahead, profiles/r5/exp_split_qc.txt) */
    const int sqc = sq_knobs().span_split_qc;
    bool split = sq_knobs().span_split && (ad || sqc > 0 || (sqc < 0 && nw >= 6));
    int waves = span_waves(P, nw, U, ad, n_ad, false, split);
    if (!waves && split) { split = false; waves = span_waves(P, nw, U, ad, n_ad, false, false); }
    if (!waves) return SQ_OK;
    const size_t lds = span_lds_layout(nw, U, ad ? SPAN_STATES(P) : 0, ad ? n_ad : 0, ad ? P.ad_lds : 0, waves, false, split).total;
    PassParams C = P;
    C.n = (P.n / SPAN_R) * SPAN_R;
    const uint64_t nspans = C.n / SPAN_R;
    const int seqs = split ? waves / 2 : waves;   /* sequences of spans per workgroup */
    const int grid = (int)std::max<uint64_t>(1, std::min<uint64_t>((nspans + seqs - 1) / seqs, (uint64_t)ctx->num_cus));
    int rc = span_needs_w6(C, ad) ? sq_span_launch_w6(nw, false, split, ctx, C, n_ad, waves, lds, grid)
                                  : split ? launch_any<false, true>(nw, ctx, C, ad, n_ad, waves, lds, grid) : launch_any<false, false>(nw, ctx, C, ad, n_ad, waves, lds, grid);
    if (rc) return rc;
    *done = C.n;
    return SQ_OK;
}

/* A batch of many read lengths (none longer than k_span takes; what adapter trimming leaves of a
 * file of one read length): its reads are sorted by length, longest first (one pass of a radix sort
 * that moves a 16-byte row per read: where its sequence and qualities lie, which record it is),
 * and the reads of the lengths that share a window count go through one launch of
 * k_span<NW, ., SEG>, cut into spans of 16 reads of one length.  *done = records covered: all of
 * them or none. */
int sq_span_launch_sorted(sq_ctx *ctx, const PassParams &P, bool ad, uint32_t n_ad, uint32_t min_len, uint32_t max_len, const uint32_t *len_hist, uint64_t *done)
{
    *done = 0;
    const uint64_t n = P.n;
    /* one wave for both streams: with rows gathered from all over the buffer a wave per stream measures the same
       (676-707 against 699-715 Gbases/s), and its 4-window build sits at exactly 128 registers -- a spill there
       sent the whole route to the other build without a word; profiles/r5/exp_sorted_split.txt) */
    if (!max_len || max_len > 32u * SPAN_NW_MAX || n < SPAN_R || n >= (1ull << 31)) return SQ_OK;
    /* every window count of the batch must be one the kernel takes, in one build or the other (each launch has its
       own: reads of 161-224 bases with adapters exist as a wave per stream only, the 4-window build of a wave per
       stream does not fit its registers) */
    bool split_of[SPAN_NW_MAX + 1] = {};
    for (int nw = (int)((std::max(min_len, 1u) + 31) / 32); nw <= (int)((max_len + 31) / 32); nw++) {   /* (the window counts the batch's lengths span) */
        bool found = false;
        /* the build the default dispatch names for this window count (sequali_amd/build.py::default_route_builds fails
           the build when one of them spills): one wave for both streams, except where that build does not exist (the
           automaton beyond 5 windows) or does not fit its registers (6 windows for sorted rows) */
        const bool first = ad ? nw > SPAN_NW_AD : nw == 6;
        for (const bool sp : {first, !first}) {
            if (found || (sp && !sq_knobs().span_split)) continue;
            if (span_waves(P, nw, 32 * nw, ad, n_ad, true, sp)) { split_of[nw] = sp; found = true; }
        }
        if (!found) return SQ_OK;
    }
    SpanRow *rows_out = (SpanRow *)sq_scratch(ctx, 15, n * sizeof(SpanRow));
    std::vector<uint64_t> longer((size_t)max_len + 1);   /* longer[w] = how many reads are longer than w */
    if (len_hist) {
        /* the counts per length came with the batch: the rows go to their places in one pass (k_span_scatter) */
        uint64_t sum = 0;
        for (uint32_t w = max_len; w > 0; w--) { longer[w] = sum; sum += len_hist[w]; }
        longer[0] = sum;
        if (sum != n) return SQ_OK;   /* a read without bases (or the counts are not those of these records): the general pass */
        uint32_t start[SQ_LEN_BINS] = {};
        for (uint32_t w = 1; w <= max_len; w++) start[w] = (uint32_t)longer[w];
        uint32_t *d_start = (uint32_t *)sq_scratch(ctx, 3, 2 * SQ_LEN_BINS * 4), *d_cursor = d_start ? d_start + SQ_LEN_BINS : nullptr;
        if (!rows_out || !d_start) { sq_set_error("out of device memory for the sorted spans"); return SQ_ERR_MEMORY; }
        SQ_HIP(hipMemcpyAsync(d_start, sq_host_keep(ctx, start, sizeof start), sizeof start, hipMemcpyHostToDevice, ctx->stream));
        SQ_HIP(hipMemsetAsync(d_cursor, 0, SQ_LEN_BINS * 4, ctx->stream));
        const uint64_t chunk = (uint64_t)SCATTER_THREADS * SCATTER_PER;
        sq_route(ctx, "k_span_scatter");
        hipLaunchKernelGGL(k_span_scatter, dim3((unsigned)std::min<uint64_t>((n + chunk - 1) / chunk, (uint64_t)ctx->num_cus * 4)), dim3(SCATTER_THREADS), 0,
                           ctx->stream, P.metas, n, d_start, d_cursor, rows_out);
    } else {
    uint32_t *keys_in = (uint32_t *)sq_scratch(ctx, 0, n * 4), *keys_out = (uint32_t *)sq_scratch(ctx, 1, n * 4);
    SpanRow *rows_in = (SpanRow *)sq_scratch(ctx, 14, n * sizeof(SpanRow));
    unsigned long long *d_longer = (unsigned long long *)sq_scratch(ctx, 3, ((size_t)max_len + 1) * 8);
    if (!keys_in || !keys_out || !rows_in || !rows_out || !d_longer) { sq_set_error("out of device memory for the sorted spans"); return SQ_ERR_MEMORY; }
    sq_route(ctx, "k_span_keys+radix_sort");
    hipLaunchKernelGGL(k_span_keys, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 8192)), dim3(256), 0, ctx->stream,
                       P.metas, n, max_len, keys_in, rows_in);
    int bits = 1;
    while ((1ull << bits) <= max_len) bits++;
    size_t temp_bytes = 0;
    if (hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, keys_in, keys_out, rows_in, rows_out, (int)n, 0, bits, ctx->stream) != hipSuccess)
        return SQ_OK;
    void *temp = sq_scratch(ctx, 4, temp_bytes ? temp_bytes : 8);
    if (!temp) { sq_set_error("out of device memory for the sorted spans"); return SQ_ERR_MEMORY; }
    if (hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys_in, keys_out, rows_in, rows_out, (int)n, 0, bits, ctx->stream) != hipSuccess) {
        sq_set_error("radix sort failed");
        return SQ_ERR_HIP;
    }
    hipLaunchKernelGGL(k_span_longer, dim3((max_len + 256) / 256), dim3(256), 0, ctx->stream, keys_out, n, max_len, d_longer);
    SQ_HIP(hipMemcpyAsync(longer.data(), d_longer, longer.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    SQ_HIP(hipStreamSynchronize(ctx->stream));
    }
    if (longer[0] != n) return SQ_OK;   /* a read without bases: the general pass knows what to do with it */
    struct Launch { int nw, waves; std::vector<SpanSeg> segs; uint32_t spans; };
    std::vector<Launch> launches;
    for (uint32_t U = max_len; U >= 1; U--) {      /* the order of the sorted rows */
        const uint64_t first = U == max_len ? 0 : longer[U], count = longer[U - 1] - first;
        if (!count) continue;
        const int nw = (int)((U + 31) / 32);
        if (launches.empty() || launches.back().nw != nw)
            launches.push_back(Launch{nw, span_waves(P, nw, 32 * nw, ad, n_ad, true, split_of[nw]), {}, 0});
        Launch &l = launches.back();
        SpanSeg g{};
        g.span0 = l.spans;
        g.nspans = (uint32_t)((count + SPAN_R - 1) / SPAN_R);
        g.U = U;
        g.last_rows = (uint32_t)(count - (uint64_t)(g.nspans - 1) * SPAN_R);
        g.first = (uint32_t)first;
        l.segs.push_back(g);
        l.spans += g.nspans;
    }
    size_t total_segs = 0;
    for (const Launch &l : launches) total_segs += l.segs.size();
    SpanSeg *d_segs = (SpanSeg *)sq_scratch(ctx, 5, total_segs * sizeof(SpanSeg));
    if (!d_segs) { sq_set_error("out of device memory for the sorted spans"); return SQ_ERR_MEMORY; }
    size_t seg_off = 0;
    std::vector<SpanSeg> all_segs;
    for (const Launch &l : launches) all_segs.insert(all_segs.end(), l.segs.begin(), l.segs.end());
    SQ_HIP(hipMemcpyAsync(d_segs, sq_host_keep(ctx, all_segs.data(), all_segs.size() * sizeof(SpanSeg)), all_segs.size() * sizeof(SpanSeg), hipMemcpyHostToDevice, ctx->stream));
    for (const Launch &l : launches) {
        PassParams C = P;
        C.uniform_len = 0;
        C.span_segs = d_segs + seg_off;
        C.span_nsegs = (uint32_t)l.segs.size();
        C.span_total = l.spans;
        C.span_rows = rows_out;
        const bool split = split_of[l.nw];
        const size_t lds = span_lds_layout(l.nw, 32 * l.nw, ad ? SPAN_STATES(P) : 0, ad ? n_ad : 0, ad ? P.ad_lds : 0, l.waves, true, split).total;
        const int seqs = split ? l.waves / 2 : l.waves;
        const int grid = (int)std::max<uint64_t>(1, std::min<uint64_t>(((uint64_t)l.spans + seqs - 1) / seqs, (uint64_t)ctx->num_cus));
        int rc = span_needs_w6(C, ad) ? sq_span_launch_w6(l.nw, true, split, ctx, C, n_ad, l.waves, lds, grid)
                                      : split ? launch_any<true, true>(l.nw, ctx, C, ad, n_ad, l.waves, lds, grid) : launch_any<true, false>(l.nw, ctx, C, ad, n_ad, l.waves, lds, grid);
        if (rc) return rc;
        seg_off += l.segs.size();
    }
    *done = n;
    return SQ_OK;
}

/* PerTileQuality alone over the first 16 * (n / 16) records of a batch of one read length with the
 * table in LDS (k_ptspan).  *done = records covered, 0 when the table does not fit or the reads
 * are too long. */
int sq_ptspan_launch(sq_ctx *ctx, const PassParams &P, uint32_t nslots, uint64_t *done)
{
    *done = 0;
    const uint32_t U = P.uniform_len;
    if (!U || U > 32 * SPAN_NW_MAX || P.n < SPAN_R || !nslots) return SQ_OK;
    const int nw = (int)((U + 31) / 32);
    int waves = 16;
    while (waves >= 4 && ptspan_lds_layout(nw, U, nslots, waves).total > 160 * 1024) waves--;
    if (waves < 4) return SQ_OK;
    const size_t lds = ptspan_lds_layout(nw, U, nslots, waves).total;
    PassParams C = P;
    C.n = (P.n / SPAN_R) * SPAN_R;
    const uint64_t nspans = C.n / SPAN_R;
    const int grid = (int)std::max<uint64_t>(1, std::min<uint64_t>((nspans + waves - 1) / waves, (uint64_t)ctx->num_cus));
    int rc;
    switch (nw) {
        case 1: rc = launch_ptspan<1>(ctx, C, nslots, waves, lds, grid); break;
        case 2: rc = launch_ptspan<2>(ctx, C, nslots, waves, lds, grid); break;
        case 3: rc = launch_ptspan<3>(ctx, C, nslots, waves, lds, grid); break;
        case 4: rc = launch_ptspan<4>(ctx, C, nslots, waves, lds, grid); break;
        case 5: rc = launch_ptspan<5>(ctx, C, nslots, waves, lds, grid); break;
        case 6: rc = launch_ptspan<6>(ctx, C, nslots, waves, lds, grid); break;
        case 7: rc = launch_ptspan<7>(ctx, C, nslots, waves, lds, grid); break;
        default: rc = launch_ptspan<8>(ctx, C, nslots, waves, lds, grid); break;
    }
    if (rc) return rc;
    *done = C.n;
    return SQ_OK;
}

/* The overlap scan of InsertSizeMetrics over the first 16 * (n / 16) pairs of two batches of one
 * read length each (k_isz_span).  *done = pairs covered (0: the kernel does not take these
 * lengths). */
int sq_isz_span_launch(sq_ctx *ctx, const IszSpanParams &P, uint64_t *done)
{
    *done = 0;
    if (P.L1 < 16 || P.L2 < 16 || P.L1 > 32u * SPAN_NW_MAX || P.n < SPAN_R) return SQ_OK;
    const int nw = (int)((P.L1 + 31) / 32);
    int waves = 16;
    while (waves >= 4 && isz_span_lds_layout(nw, P.lds_sizes, waves).total > 160 * 1024) waves--;
    if (waves < 4) return SQ_OK;
    const size_t lds = isz_span_lds_layout(nw, P.lds_sizes, waves).total;
    IszSpanParams C = P;
    C.n = (P.n / SPAN_R) * SPAN_R;
    const uint64_t nspans = C.n / SPAN_R;
    const int grid = (int)std::max<uint64_t>(1, std::min<uint64_t>((nspans + waves - 1) / waves, (uint64_t)ctx->num_cus));
    int rc;
    switch (nw) {
        case 1: rc = launch_isz_span<1>(ctx, C, waves, lds, grid); break;
        case 2: rc = launch_isz_span<2>(ctx, C, waves, lds, grid); break;
        case 3: rc = launch_isz_span<3>(ctx, C, waves, lds, grid); break;
        case 4: rc = launch_isz_span<4>(ctx, C, waves, lds, grid); break;
        case 5: rc = launch_isz_span<5>(ctx, C, waves, lds, grid); break;
        case 6: rc = launch_isz_span<6>(ctx, C, waves, lds, grid); break;
        case 7: rc = launch_isz_span<7>(ctx, C, waves, lds, grid); break;
        default: rc = launch_isz_span<8>(ctx, C, waves, lds, grid); break;
    }
    if (rc) return rc;
    *done = C.n;
    return SQ_OK;
}
