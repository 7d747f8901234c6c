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

#include "sq_pass.h"
#include "sq_span.h"

namespace {

constexpr uint32_t SPAN_R = 16;              /* records per span */
constexpr uint32_t CLS6_PAD4 = 0x1E1E1E1Eu;  /* code 30 */

/* Four sequence bytes -> four class codes that are the shifts of the class's 6-bit counter
 * field: A 0, C 6, G 12, T 18, anything else 24 (NUCLEOTIDE_TO_INDEX, _qcmodule.c:1748-1763);
 * see cls2_of_dword in sq_pass.h for the method.  Bytes must be 7-bit ASCII. */
__device__ __forceinline__ uint32_t cls6_of_dword(uint32_t w)
{
    const uint32_t idx = w & 0x07070707u;
    const uint32_t lut = __builtin_amdgcn_perm(0x0C181812u, 0x06180018u, idx);
    const uint32_t want = __builtin_amdgcn_perm(0x47000054u, 0x43004100u, idx);
    const uint32_t d = (w & 0xDFDFDFDFu) ^ want;
    const uint32_t ne4 = ((d + 0x7F7F7F7Fu) & 0x80808080u) >> 5;
    return __builtin_amdgcn_perm(lut, 0x18181818u, 0x07060504u - ne4);
}

__device__ __forceinline__ uint32_t lds_u8(uint32_t a) { return *(SQ_LDS const uint8_t *)(uintptr_t)a; }
__device__ __forceinline__ void lds_store_u32(uint32_t a, uint32_t v) { *(SQ_LDS uint32_t *)(uintptr_t)a = v; }
__device__ __forceinline__ void lds_add(uint32_t a, uint32_t v)
{
    __hip_atomic_fetch_add((SQ_LDS uint32_t *)(uintptr_t)a, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_min(uint32_t a, uint32_t v)
{
    __hip_atomic_fetch_min((SQ_LDS uint32_t *)(uintptr_t)a, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
/* (1 << sh) + c */
__device__ __forceinline__ uint32_t one_shl_add(uint32_t sh, uint32_t c)
{
    uint32_t r;
    asm("v_lshl_add_u32 %0, 1, %1, %2" : "=v"(r) : "v"(sh), "v"(c));
    return r;
}
/* sum of the four bytes of x, + c */
__device__ __forceinline__ uint32_t sum_bytes(uint32_t x, uint32_t c)
{
    uint32_t r;
    asm("v_sad_u8 %0, %1, 0, %2" : "=v"(r) : "v"(x), "v"(c));
    return r;
}
/* 16 bytes per lane from `g` (any alignment) to LDS at lds_dst + 16 * lane.  hipcc neither counts
 * this load nor knows that it writes LDS: the kernel waits for it by hand (vmcnt) */
__device__ __forceinline__ void dma16(const uint8_t *g, uint32_t lds_dst)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_dst) : "memory");
}
template <int CTRL> __device__ __forceinline__ uint32_t quad_bcast(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true);
}
template <int CTRL> __device__ __forceinline__ double quad_bcast_f64(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = quad_bcast<CTRL>((uint32_t)b), hi = quad_bcast<CTRL>((uint32_t)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

/* waves per workgroup: LDS leaves at most 12 of them from 128 positions per read on, which is
 * 168 registers per lane instead of 128 */
constexpr int span_max_waves(int nw) { return nw <= 3 ? 16 : 12; }

template <int NW, bool AD>
__global__ void __launch_bounds__(64 * span_max_waves(NW)) k_span(PassParams P, uint32_t n_ad)
{
    constexpr uint32_t SB = 32 * NW, ROWB = 64 * NW, SLOT = SPAN_R * ROWB, Q4 = 2 * NW;
    extern __shared__ __align__(16) uint8_t smem[];
    const uint32_t U = P.uniform_len, hs = hist_stride(U);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, T = blockDim.x, W = T >> 6;
    const SpanLds L = span_lds_layout(NW, U, AD ? P.dfa_states : 0, AD ? n_ad : 0, AD ? P.ad_lds : 0, W);
    double *l_err = (double *)smem;                        /* [136] by raw quality byte */
    double *l_thr = (double *)(smem + L.thr);              /* [96] */
    uint32_t *l_gc = (uint32_t *)(smem + L.gc);            /* [104] */
    uint32_t *l_ps = (uint32_t *)(smem + L.ps);            /* [96] */
    uint16_t *l_dfa = (uint16_t *)(smem + L.dfa);          /* [states][16]: rows of 32 bytes, entry at byte `code` */
    unsigned long long *l_out = (unsigned long long *)(smem + L.out); /* [states] adapters ending there */
    uint8_t *l_adlen = smem + L.adlen;                     /* [64] */
    uint32_t *l_hist_base = (uint32_t *)(smem + L.hist);   /* [5][hs] */
    uint32_t *l_hist_phred = l_hist_base + hs * BASE_COLS; /* [12][hs] */
    uint32_t *l_adf = l_hist_phred + hs * PHRED_COLS;      /* [ad_lds][hs] */
    uint32_t *l_first = (uint32_t *)(smem + L.first) + wave * SPAN_R * (AD ? n_ad : 0); /* [16][n_ad] */
    uint32_t *l_rows = (uint32_t *)(smem + L.rows) + wave * 2 * SPAN_R;                 /* [16][2] */
    const uint32_t slot_base = lds_addr(smem + L.slots) + wave * 2 * SLOT;

    if (lds_addr(l_err) != 0) __builtin_trap(); /* quality byte << 3 is the address of its error rate */
    const uint32_t dfa_root = AD ? lds_addr(l_dfa) : 0, dfa_hit = dfa_root + P.dfa_accept * 32;
    for (int i = tid; i < 136; i += T) {
        double e;
        if (i >= 33 && i <= 33 + SQ_PHRED_MAX) e = __longlong_as_double((long long)c_error_rate_bits[i - 33]);
        else if (i >= 128) e = 0.0;
        else e = __longlong_as_double(0x7FF8000000000000LL);
        l_err[i] = e;
    }
    for (int i = tid; i < 96; i += T) l_thr[i] = i < 94 ? P.thresholds[i] : 0.0;
    for (int i = tid; i < 104; i += T) l_gc[i] = 0;
    for (int i = tid; i < 96; i += T) l_ps[i] = 0;
    for (uint32_t i = tid; i < hs * (BASE_COLS + PHRED_COLS); i += T) l_hist_base[i] = 0;
    if (AD) {
        for (uint32_t i = tid; i < P.dfa_states * 16; i += T) {
            const uint32_t s = i >> 4, k = i & 15;
            /* byte offset 6 c holds the row behind class c; everything else (30: padding) the root */
            uint32_t next = dfa_root;
            if (k % 3 == 0 && k < 15) next = dfa_root + ((uint32_t)(P.dfa[s * 8 + k / 3] >> 4) << 5);
            l_dfa[i] = (uint16_t)next;
        }
        for (uint32_t i = tid; i < P.dfa_states; i += T) l_out[i] = P.dfa_out[i];
        for (uint32_t i = tid; i < 64; i += T) l_adlen[i] = P.ad_len[i];
        for (uint32_t i = tid; i < P.ad_lds * hs; i += T) l_adf[i] = 0;
        for (uint32_t i = lane; i < SPAN_R * n_ad; i += 64) l_first[i] = 0xFFFFFFFFu;
    }
    __syncthreads();

    const uint32_t q = (uint32_t)lane >> 2, c = (uint32_t)lane & 3;
    /* DMA: piece i = 64 k + lane of a slot is 16 bytes of row i / (4 NW), stream and offset by
       the rest; where that row's stream starts (relative to the span's first record) is read
       from l_rows */
    uint32_t dma_tbl[NW], dma_off[NW];
#pragma unroll
    for (int k = 0; k < NW; k++) {
        const uint32_t i = 64 * k + lane, row = i / (4 * NW), pir = i % (4 * NW), stream = pir >= 2 * NW;
        dma_tbl[k] = lds_addr(l_rows) + row * 8 + stream * 4;
        dma_off[k] = (pir - stream * 2 * NW) * 16;
    }
    const uint64_t nspans = P.n / SPAN_R;
    const uint64_t stride = (uint64_t)gridDim.x * W;
    uint64_t s = (uint64_t)blockIdx.x * W + wave;

    unsigned long long m_rs = 0; /* metas of the span to be fetched next: record start, offsets */
    uint32_t m_so = 0, m_qo = 0;
    auto load_meta = [&](uint64_t sp) {
        const sq_meta *m = P.metas + sp * SPAN_R + q;
        m_rs = m->record_start;
        m_so = m->sequence_offset;
        m_qo = m->qualities_offset;
    };
    auto issue = [&](uint32_t slot_addr) {
        const unsigned long long base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(m_rs >> 32)) << 32) |
                                        (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)m_rs);
        const uint32_t rel = (uint32_t)(m_rs - base);
        if (c == 0) { l_rows[2 * q] = rel + m_so; l_rows[2 * q + 1] = rel + m_qo; }
        const uint8_t *g0 = P.buf + base;
#pragma unroll
        for (int k = 0; k < NW; k++) {
            const int32_t r = (int32_t)lds_u32(dma_tbl[k]);
            dma16(g0 + (long long)r + dma_off[k], __builtin_amdgcn_readfirstlane(slot_addr + 1024 * k));
        }
    };

    const uint32_t Lmain = 4 * ((U - 1) / 4), nsteps = Lmain / 4; /* _qcmodule.c:2062,2068 */
    const uint32_t W4 = AD ? (P.ad_maxlen + 2) / 4 : 0;            /* dwords holding >= maxlen - 1 positions */
    const uint32_t npad = SB - U;
    const uint32_t h = (uint32_t)lane >> 5, pl = (uint32_t)lane & 31;
    uint32_t cnt[NW];
#pragma unroll
    for (int w = 0; w < NW; w++) cnt[w] = 0;
    uint32_t since_flush = 0;
    auto flush_counts = [&]() {
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const uint32_t p = 32 * w + pl;
            if (p < U) {
#pragma unroll
                for (uint32_t f = 0; f < 5; f++) {
                    const uint32_t v = __builtin_amdgcn_ubfe(cnt[w], 6 * f, 6);
                    if (v) lds_add(lds_addr(l_hist_base + f * hs + p), v);
                }
            }
            cnt[w] = 0;
        }
    };

    int cur = 0;
    if (s < nspans) {
        load_meta(s);
        issue(slot_base);
        if (s + stride < nspans) load_meta(s + stride);
    }
    while (s < nspans) {
        /* the span in slot `cur` has landed (and the metas of the one after it have arrived) */
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (s + stride < nspans) {
            issue(slot_base + (cur ^ 1) * SLOT);
            if (s + 2 * stride < nspans) load_meta(s + 2 * stride);
        }
        const uint32_t sa = slot_base + cur * SLOT;
        const uint64_t r = s * SPAN_R + q;
        const uint32_t seq_row = sa + q * ROWB, qual_row = seq_row + SB;

        /* ---------------- phase S: four lanes per read ---------------- */
        uint32_t st = dfa_root, gacc = 0, nacc = 0;
        bool any_hit = false;
        const uint32_t seg = seq_row + 4 * Q4 * c; /* this lane's quarter: dwords [Q4 c, Q4 (c + 1)) */
        if (AD) {
#pragma unroll 1
            for (uint32_t t = 0; t < W4; t++) { /* the stretch in front of the quarter: state only */
                uint32_t cl = cls6_of_dword(lds_u32(seg - 4 * W4 + 4 * t));
                cl = Q4 * c + t < W4 ? CLS6_PAD4 : cl; /* nothing in front of the read */
                st = lds_u16(or_byte<0>(st, cl));
                st = lds_u16(or_byte<1>(st, cl));
                st = lds_u16(or_byte<2>(st, cl));
                st = lds_u16(or_byte<3>(st, cl));
            }
        }
#pragma unroll
        for (uint32_t t = 0; t < Q4; t++) {
            uint32_t cl = cls6_of_dword(lds_u32(seg + 4 * t));
            if (4 * (3 * Q4 + t) + 4 > U) { /* a dword that reaches behind the reads in the last quarter */
                const uint32_t p0 = 4 * (Q4 * c + t);
                cl = pad_tail(cl, p0 < U ? (int)min(4u, U - p0) : 0, CLS6_PAD4);
            }
            lds_store_u32(seg + 4 * t, cl);
            gacc += cl & 0x04040404u;                 /* C, G and padding */
            nacc += cl & (cl >> 1) & 0x08080808u;     /* N and padding */
            if (AD) {
                uint32_t e[4];
                e[0] = lds_u16(or_byte<0>(st, cl));
                e[1] = lds_u16(or_byte<1>(e[0], cl));
                e[2] = lds_u16(or_byte<2>(e[1], cl));
                e[3] = lds_u16(or_byte<3>(e[2], cl));
                st = e[3];
                if (max(max(e[0], e[1]), max(e[2], e[3])) >= dfa_hit) {
                    any_hit = true;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        if (e[j] < dfa_hit) continue;
                        unsigned long long hits = l_out[(e[j] - dfa_root) >> 5];
                        const uint32_t pos = 4 * (Q4 * c + t) + j; /* where the match ends */
                        while (hits) {
                            const int a = __ffsll((long long)hits) - 1;
                            hits &= hits - 1;
                            lds_min(lds_addr(l_first + q * n_ad + a), pos);
                        }
                    }
                }
            }
        }
        /* the four chains, :2062-2097 */
        double acc = 0.0;
        {
            const uint32_t qa = qual_row + c;
#pragma unroll 4
            for (uint32_t k = 0; k < nsteps; k++) acc += lds_f64(lds_u8(qa + 4 * k) << 3);
        }
        double total = ((acc + quad_bcast_f64<0x55>(acc)) + quad_bcast_f64<0xAA>(acc)) + quad_bcast_f64<0xFF>(acc); /* :2098-2099 (lane c = 0) */
        for (uint32_t p = Lmain; p < U; p++) total += lds_f64(lds_u8(qual_row + p) << 3); /* :2100-2112 */
        uint32_t gsum = sum_bytes(gacc, 0), nsum = sum_bytes(nacc, 0);
        gsum += quad_bcast<0xB1>(gsum); nsum += quad_bcast<0xB1>(nsum); /* quad_perm [1,0,3,2] */
        gsum += quad_bcast<0x4E>(gsum); nsum += quad_bcast<0x4E>(nsum); /* quad_perm [2,3,0,1] */
        if (c == 0) {
            const uint32_t gc_cnt = (gsum >> 2) - npad, acgt_cnt = SB - (nsum >> 3);
            P.metas[r].accumulated_error_rate = total; /* :2126 */
            if (total != total) atomicMin(P.qc_first_bad, (unsigned long long)(P.first_read_index + r));
            if (acgt_cnt > 0) atomicAdd(&l_gc[(uint32_t)round((double)gc_cnt * 100.0 / (double)acgt_cnt)], 1u);
            const double avg = total / (double)U;
            uint32_t lo = 0, hi = 93;
            while (lo < hi) {
                const uint32_t mid = (lo + hi + 1) >> 1;
                if (avg <= l_thr[mid]) lo = mid; else hi = mid - 1;
            }
            atomicAdd(&l_ps[lo], 1u);
        }
        if (AD && __builtin_amdgcn_ballot_w64(any_hit)) { /* update_adapter_count_array, :2643-2672 */
            for (uint32_t i = lane; i < SPAN_R * n_ad; i += 64) {
                const uint32_t v = l_first[i];
                if (v == 0xFFFFFFFFu) continue;
                l_first[i] = 0xFFFFFFFFu;
                const uint32_t a = i % n_ad, start = v - l_adlen[a] + 1;
                if (P.ad_lds) {
                    atomicAdd(&l_adf[a * hs + start], 1u);
                } else {
                    atomicAdd(&P.ad_fwd[a * P.ad_cap + start], 1ULL);
                    atomicAdd(&P.ad_rev[a * P.ad_cap + (U - 1 - start)], 1ULL);
                }
            }
        }

        /* ---------------- phase H: lane = position, two rows per instruction ---------------- */
        {
            const uint32_t bs = sa + h * ROWB + pl;
            const uint32_t hpp = lds_addr(l_hist_phred + pl);
#pragma unroll
            for (int w = 0; w < NW; w++) {
                if (32u * w + pl < U) {
#pragma unroll
                    for (int j = 0; j < (int)SPAN_R / 2; j++) {
                        const uint32_t cb = lds_u8(bs + 2 * j * ROWB + 32 * w);
                        const uint32_t qb = lds_u8(bs + 2 * j * ROWB + 32 * w + SB);
                        cnt[w] = one_shl_add(cb, cnt[w]);
                        const uint32_t bin = min(qb - 33u, 47u) >> 2; /* :1767-1784 */
                        lds_inc(hpp + 128 * w + __umul24(bin, hs * 4));
                    }
                }
            }
        }
        if (++since_flush == 7) { flush_counts(); since_flush = 0; }
        cur ^= 1;
        s += stride;
    }
    flush_counts();

    /* ---- merge the workgroup's histograms (end-anchored = a window of the positional) ---- */
    __syncthreads();
    if (AD)
        for (uint32_t i = tid; i < P.ad_lds * hs; i += T) {
            const uint32_t v = l_adf[i];
            if (!v) continue;
            const uint32_t a = i / hs, start = i % hs;
            atomicAdd(&P.ad_fwd[a * P.ad_cap + start], (unsigned long long)v);
            atomicAdd(&P.ad_rev[a * P.ad_cap + (U - 1 - start)], (unsigned long long)v);
        }
    const uint32_t ean = min(P.ea_len, U);
    for (uint32_t i = tid; i < hs * BASE_COLS; i += T) {
        const uint32_t v = l_hist_base[i], cc = i / hs, pos = i % hs;
        if (!v || pos >= U) continue;
        atomicAdd(&P.qc_base[(uint64_t)pos * 5 + cc], (unsigned long long)v);
        if (pos >= U - ean) atomicAdd(&P.qc_ea_base[(uint64_t)(P.ea_len - U + pos) * 5 + cc], (unsigned long long)v);
    }
    for (uint32_t i = tid; i < hs * PHRED_COLS; i += T) {
        const uint32_t v = l_hist_phred[i], cc = i / hs, pos = i % hs;
        if (!v || pos >= U) continue;
        atomicAdd(&P.qc_phred[(uint64_t)pos * 12 + cc], (unsigned long long)v);
        if (pos >= U - ean) atomicAdd(&P.qc_ea_phred[(uint64_t)(P.ea_len - U + pos) * 12 + cc], (unsigned long long)v);
    }
    for (uint32_t i = tid; i < 101; i += T)
        if (l_gc[i]) atomicAdd(&P.qc_gc[i], (unsigned long long)l_gc[i]);
    for (uint32_t i = tid; i < 94; i += T)
        if (l_ps[i]) atomicAdd(&P.qc_ps[i], (unsigned long long)l_ps[i]);
}

template <int NW>
int launch_nw(sq_ctx *ctx, const PassParams &P, bool ad, uint32_t n_ad, int waves, size_t lds, int grid)
{
    static bool attr = false;
    if (!attr) {
        SQ_HIP(hipFuncSetAttribute((const void *)k_span<NW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        SQ_HIP(hipFuncSetAttribute((const void *)k_span<NW, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    if (ad) hipLaunchKernelGGL((k_span<NW, true>), dim3(grid), dim3(waves * 64), lds, ctx->stream, P, n_ad);
    else hipLaunchKernelGGL((k_span<NW, false>), dim3(grid), dim3(waves * 64), lds, ctx->stream, P, n_ad);
    SQ_HIP(hipGetLastError());
    return SQ_OK;
}

} // namespace

/* Runs k_span over the first 16 * (n / 16) records of the pass described by P (QCMetrics, with
 * AdapterCounter when `ad`).  *done = records covered, 0 when the kernel does not take this
 * pass (read length, automaton size, LDS). */
int sq_span_launch(sq_ctx *ctx, const PassParams &P, bool ad, uint32_t n_ad, uint64_t *done)
{
    *done = 0;
    const uint32_t U = P.uniform_len;
    if (!U || U > 32 * SPAN_NW_MAX || P.n < SPAN_R) return SQ_OK;
    if (ad && (P.dfa_states > SPAN_DFA_MAX_STATES || n_ad > 64 || P.ad_maxlen > 64)) return SQ_OK;
    int nw = (int)((U + 31) / 32);
    while (nw <= SPAN_NW_MAX && nw != 2 && nw != 5 && nw != 8) nw++;
    if (nw > SPAN_NW_MAX) return SQ_OK;
    /* as many waves as LDS takes, at most 16 */
    int waves = span_max_waves(nw);
    while (waves >= 4 && span_lds_layout(nw, U, ad ? P.dfa_states : 0, ad ? n_ad : 0, ad ? P.ad_lds : 0, waves).total > 160 * 1024) waves--;
    if (waves < 4) return SQ_OK;
    if (const char *e = getenv("SQ_SPAN_WAVES")) waves = std::max(1, std::min(waves, atoi(e)));
    const size_t lds = span_lds_layout(nw, U, ad ? P.dfa_states : 0, ad ? n_ad : 0, ad ? P.ad_lds : 0, waves).total;
    PassParams C = P;
    C.n = (P.n / SPAN_R) * SPAN_R;
    const uint64_t nspans = C.n / SPAN_R;
    const int grid = (int)std::max<uint64_t>(1, std::min<uint64_t>((nspans + waves - 1) / waves, (uint64_t)ctx->num_cus));
    int rc;
    switch (nw) {
        case 2: rc = launch_nw<2>(ctx, C, ad, n_ad, waves, lds, grid); break;
        case 5: rc = launch_nw<5>(ctx, C, ad, n_ad, waves, lds, grid); break;
        default: rc = launch_nw<8>(ctx, C, ad, n_ad, waves, lds, grid); break;
    }
    if (rc) return rc;
    *done = C.n;
    return SQ_OK;
}
