/*
 * sq_qc.hip -- the per-base pass: QCMetrics, AdapterCounter, PerTileQuality.
 *
 * One kernel template, k_pass<QC, AD, PT>, reads a batch once and feeds up to
 * three modules (reference: three separate C loops over the same record
 * array, _qcmodule.c:1966-2139, 2786-2823, 3123-3222).
 *
 * Work decomposition (wave64; a workgroup is 4 independent waves that share
 * the LDS histograms):
 *   - a wave owns 64 consecutive reads at a time ("group") and walks them in
 *     chunks of 64 positions;
 *   - STAGE: the 64x64-byte sequence and quality tiles of the chunk are
 *     fetched with 16-byte loads (4 lanes per read, so every 64-byte segment
 *     of a read is one request) and written to LDS, XOR-swizzled so that both
 *     orientations below are bank-conflict free; bytes past the end of a read
 *     become 0x80, a value FASTQ text cannot contain;
 *   - phase S (lane = read): everything that is sequential inside a read --
 *     the four interleaved f64 error-rate chains in the reference's exact
 *     order, and the adapter automaton;
 *   - phase H (lane = position): the per-position histograms.  All 64 lanes
 *     hit different positions, so the LDS atomics never collide inside an
 *     instruction (a lane=read layout would send 64 lanes to <= 5 counters);
 *     per-read GC/AT totals fall out of two ballots per row;
 *   - per-read epilogue (lane = read): the 1-4 trailing qualities in order,
 *     GC% and mean-phred bins, accumulated_error_rate write-back.
 *   LDS histograms are merged into the u64 tables in HBM once per workgroup.
 */
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>

#include "sq_pass.h"
#include "sq_span.h"

/* state of a read between two stripes */
struct sq_carry {
    double acc[4];
    unsigned long long found;
    uint32_t st, gc_cnt, acgt_cnt, pad_;
};

namespace {

template <bool QC, bool AD, bool PT, bool DFA_LDS>
__global__ void __launch_bounds__(WG_THREADS, 4) k_pass(PassParams P)
{
    extern __shared__ __align__(16) uint8_t smem[];
    /* ---- LDS carve-up (every region 16-byte aligned) ---- */
    double *l_err = (double *)smem;                        /* [136] by raw quality byte, 128 = padding */
    double *l_thr = l_err + 136;                           /* [96] */
    uint32_t *l_gc = (uint32_t *)(l_thr + 96);             /* [104] */
    uint32_t *l_ps = l_gc + 104;                           /* [96] */
    /* the automaton sits in front of everything of variable size: its rows are addressed by
       16-bit LDS addresses (table entries are the address of the next row) */
    uint16_t *l_dfa = (uint16_t *)(l_ps + 96);
    uint32_t *l_wave = (uint32_t *)(l_dfa + ((AD && DFA_LDS) ? P.dfa_states * 8 : 0)); /* per wave: tiles, offsets, lengths */
    const uint32_t hs = QC ? hist_stride(P.lds_len) : 0;   /* words per class row */
    uint32_t *l_hist_base = l_wave + WAVES * WAVE_WORDS;   /* [5][hs] */
    uint32_t *l_hist_phred = l_hist_base + hs * BASE_COLS;               /* [12][hs] */
    uint32_t *l_ea_base = l_hist_phred + hs * PHRED_COLS;
    const uint32_t ea_rows = (QC && P.ea_in_lds && !P.uniform_len) ? P.ea_len : 0;
    const uint32_t es = hist_stride(ea_rows);
    uint32_t *l_ea_phred = l_ea_base + es * BASE_COLS;
    uint32_t *l_adf = l_ea_phred + es * PHRED_COLS;       /* [ad_lds][hs] */
    /* PerTileQuality: per wave, the error-rate sums of the tile its groups are in, [position] */
    const uint32_t pts = PT ? hist_stride(P.lds_len) : 0;
    double *l_pt = (double *)(l_adf + (((AD && DFA_LDS && QC) ? P.ad_lds * hs : 0) + 1) / 2 * 2);

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    /* the fused loop turns a quality byte into the address of its error rate with one shift:
       the kernel has no static LDS, so the dynamic block (and l_err) starts at address 0 */
    if (lds_addr(l_err) != 0) __builtin_trap();

    /* ---- fill tables ---- */
    for (int i = tid; i < 136; i += WG_THREADS) {
        double e;
        if (i >= 33 && i <= 33 + SQ_PHRED_MAX) {
            unsigned long long bits = c_error_rate_bits[i - 33];
            e = __longlong_as_double((long long)bits);
        } else if (i >= 128) {
            e = 0.0; /* padding past the end of a read adds nothing */
        } else {
            e = __longlong_as_double(0x7FF8000000000000LL); /* invalid phred byte poisons the sum */
        }
        l_err[i] = e;
    }
    if (QC) {
        for (int i = tid; i < 96; i += WG_THREADS) l_thr[i] = i < 94 ? P.thresholds[i] : 0.0;
        for (int i = tid; i < 104; i += WG_THREADS) l_gc[i] = 0;
        for (int i = tid; i < 96; i += WG_THREADS) l_ps[i] = 0;
        uint32_t nh = (hs + es) * (BASE_COLS + PHRED_COLS);
        for (uint32_t i = tid; i < nh; i += WG_THREADS) l_hist_base[i] = 0;
    }
    /* in LDS a table entry is the LDS address of the next row; a row some adapter ends in lies
       at or behind dfa_hit (build_dfa numbers those states last) */
    const uint32_t dfa_root = (AD && DFA_LDS) ? lds_addr(l_dfa) : 0;
    const uint32_t dfa_hit = dfa_root + P.dfa_accept * 16;
    if (PT)
        for (uint32_t i = tid; i < WAVES * pts; i += WG_THREADS) l_pt[i] = 0.0;
    if (AD && DFA_LDS) {
        for (uint32_t i = tid; i < P.dfa_states * 8; i += WG_THREADS)
            l_dfa[i] = (uint16_t)((P.dfa[i] & 0xFFF0u) + dfa_root);
        if (QC)
            for (uint32_t i = tid; i < P.ad_lds * hs; i += WG_THREADS) l_adf[i] = 0;
    }
    __syncthreads();

    uint32_t *w_seq = l_wave + wave * WAVE_WORDS;
    uint32_t *w_qual = w_seq + QUAL_WORDS;
    const uint32_t s_row = lds_addr(w_seq) + ((uint32_t)lane >> 3) * 512 + ((uint32_t)lane & 7) * 32;
    unsigned long long *w_soff = (unsigned long long *)(w_seq + 2 * TILE_WORDS);
    unsigned long long *w_qoff = w_soff + 64;
    uint32_t *w_len = (uint32_t *)(w_qoff + 64);

    const uint64_t ngroups = (P.n + 63) / 64;
    const bool ea_atomics = QC && !P.uniform_len;
    /* phase H: lanes 0-31 take an even row, lanes 32-63 the odd row after it */
    const uint32_t half = (uint32_t)lane >> 5, pl = (uint32_t)lane & 31;
    const uint32_t h_sh = 8 * (pl & 3), h_dw = pl >> 2;
    /* byte-address parts of the fused loop (ptile_idx / ptile_idx_h split into what is fixed per
       lane and what changes per step) */
    const uint32_t s_sw4 = 4 * (((uint32_t)lane >> 2) & 7), h_u4 = 32 * half + 4 * h_dw;

    const uint64_t n_waves = (uint64_t)gridDim.x * WAVES, wave_id = (uint64_t)blockIdx.x * WAVES + wave;
    const uint64_t per_wave = (ngroups + n_waves - 1) / n_waves;
    const uint64_t g_begin = P.blocked ? wave_id * per_wave : wave_id;
    const uint64_t g_end = P.blocked ? min(ngroups, g_begin + per_wave) : ngroups;
    const uint64_t g_step = P.blocked ? 1 : n_waves;
    /* device atomics on a tile's row come from every wave that is in that tile, from all XCDs:
       a wave keeps the sums of its current tile in LDS and hands them over when its groups move
       on to another tile */
    double *w_pt = l_pt + wave * pts;
    int32_t pt_acc_slot = -1;
    uint32_t pt_acc_reads = 0; /* reads of that tile the wave has seen (all of length uniform_len) */
    auto pt_flush = [&]() {
        if (pt_acc_slot < 0) return;
        if (lane == 0 && pt_acc_reads)
            atomicAdd(&P.pt_len_counts[(uint64_t)pt_acc_slot * P.pt_cap + (P.uniform_len - 1)],
                      (unsigned long long)pt_acc_reads);
        pt_acc_reads = 0;
        for (uint32_t i = lane; i < pts; i += 64) {
            const double v = w_pt[i];
            if (v != 0.0) {
                unsafeAtomicAdd(&P.pt_errors[(uint64_t)pt_acc_slot * P.pt_cap + i], v);
                w_pt[i] = 0.0;
            }
        }
    };
    for (uint64_t g = g_begin; g < g_end; g += g_step) {
        const uint64_t slot_index = g * 64 + lane;
        bool valid = slot_index < P.n;
        const uint64_t r = (valid && P.order) ? P.order[slot_index] : slot_index;
        sq_meta m;
        if (valid) m = P.metas[r];
        /* in a later stripe only the reads that reach it take part */
        if (valid && P.pos_base && m.sequence_length <= P.pos_base) valid = false;
        const uint32_t L = valid ? m.sequence_length : 0;
        const uint64_t soff = valid ? m.record_start + m.sequence_offset : 0;
        const uint64_t qoff = valid ? m.record_start + m.qualities_offset : 0;
        w_soff[lane] = soff;
        w_qoff[lane] = qoff;
        w_len[lane] = L;
        const uint32_t maxL = wave_max_u32(L);
        const uint32_t Lmain = L > 0 ? 4 * ((L - 1) / 4) : 0; /* _qcmodule.c:2062,2068 */

        double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
        uint32_t st = dfa_root;          /* automaton state: where its row starts (LDS address or byte offset) */
        unsigned long long found = 0;    /* adapters already seen in this read */
        uint32_t gc_cnt = 0, acgt_cnt = 0;
        if (P.pos_base && valid) {       /* pick up where the stripe in front left this read */
            const sq_carry cy = P.carry[slot_index];
            acc0 = cy.acc[0]; acc1 = cy.acc[1]; acc2 = cy.acc[2]; acc3 = cy.acc[3];
            found = cy.found; st = cy.st + dfa_root; gc_cnt = cy.gc_cnt; acgt_cnt = cy.acgt_cnt;
        }
        const uint32_t stripe_end = min(maxL, P.pos_end);
        bool pt_on = false;
        int32_t pt_slot = -1;
        if (PT) {
            pt_slot = valid ? P.pt_slot[r] : -1;
            pt_on = valid && pt_slot >= 0 && (P.first_read_index + r) < P.pt_first_bad;
            if (!pt_on) pt_slot = -1;
        }
        /* every read of the group counts for the same tile (wave-uniform slot) */
        const int32_t pt_group_slot = PT ? __builtin_amdgcn_readfirstlane(pt_slot) : -1;
        const bool pt_one_tile = PT && pt_group_slot >= 0 && __all(pt_on && pt_slot == pt_group_slot);
        if (PT && pt_one_tile && pt_group_slot != pt_acc_slot) {
            pt_flush();
            pt_acc_slot = pt_group_slot;
        }
        /* the staging lanes of other rows read w_soff / w_qoff / w_len */
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        /* global loads of a chunk are issued one chunk ahead and sit in registers
           while the previous chunk is being counted */
        uint4 pf_s[2], pf_q[2];
        auto prefetch = [&](uint32_t c0) {
#pragma unroll
            for (int it = 0; it < 2; it++) {
                const uint32_t row = it * 32 + ((uint32_t)lane >> 1), piece = (uint32_t)lane & 1;
                const uint32_t p0 = c0 + piece * 16;
                pf_s[it] = make_uint4(0, 0, 0, 0);
                pf_q[it] = make_uint4(PAD4, PAD4, PAD4, PAD4);
                if (p0 < w_len[row]) {
                    if (QC || AD) pf_s[it] = load16(P.buf, w_soff[row] + p0, P.buf_len);
                    if (QC || PT) pf_q[it] = load16(P.buf, w_qoff[row] + p0, P.buf_len);
                }
            }
        };
        if (stripe_end > P.pos_base) prefetch(P.pos_base);

        for (uint32_t c0 = P.pos_base; c0 < stripe_end; c0 += CW) {
            /* ---------------- STAGE: 2 lanes x 16 bytes per row ---------------- */
#pragma unroll
            for (int it = 0; it < 2; it++) {
                const uint32_t row = it * 32 + ((uint32_t)lane >> 1), piece = (uint32_t)lane & 1;
                const uint32_t Lr = w_len[row];
                const uint32_t p0 = c0 + piece * 16;
                uint4 sv = make_uint4(CLS2_PAD4, CLS2_PAD4, CLS2_PAD4, CLS2_PAD4);
                uint4 qv = pf_q[it];
                if (p0 < Lr) {
                    const int nv = (int)min(16u, Lr - p0);
                    if (QC || AD) {
                        sv.x = cls2_of_dword(pf_s[it].x); sv.y = cls2_of_dword(pf_s[it].y);
                        sv.z = cls2_of_dword(pf_s[it].z); sv.w = cls2_of_dword(pf_s[it].w);
                    }
                    if (nv < 16) {
                        sv.x = pad_tail(sv.x, nv, CLS2_PAD4); sv.y = pad_tail(sv.y, nv - 4, CLS2_PAD4);
                        sv.z = pad_tail(sv.z, nv - 8, CLS2_PAD4); sv.w = pad_tail(sv.w, nv - 12, CLS2_PAD4);
                        qv.x = pad_tail(qv.x, nv, PAD4); qv.y = pad_tail(qv.y, nv - 4, PAD4);
                        qv.z = pad_tail(qv.z, nv - 8, PAD4); qv.w = pad_tail(qv.w, nv - 12, PAD4);
                    }
                }
                const uint32_t d0 = piece * 4;
                if (QC || AD) {
                    w_seq[ptile_idx(row, d0 + 0)] = sv.x;
                    w_seq[ptile_idx(row, d0 + 1)] = sv.y;
                    w_seq[ptile_idx(row, d0 + 2)] = sv.z;
                    w_seq[ptile_idx(row, d0 + 3)] = sv.w;
                }
                if (QC || PT) {
                    w_qual[ptile_idx(row, d0 + 0)] = qv.x;
                    w_qual[ptile_idx(row, d0 + 1)] = qv.y;
                    w_qual[ptile_idx(row, d0 + 2)] = qv.z;
                    w_qual[ptile_idx(row, d0 + 3)] = qv.w;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (c0 + CW < stripe_end) prefetch(c0 + CW);

            /* ---------------- phase S: lane = read ---------------- */
            /* one dword (4 positions) of the per-read work; the automaton's four table entries
               come back in e[] so that the caller can put other work in front of the (rare)
               hit handling */
            auto s_main = [&](uint32_t d, uint32_t e[4]) -> bool {
                const uint32_t pos0 = c0 + d * 4;
                const uint32_t ti = ptile_idx((uint32_t)lane, d);
                const uint32_t sd = w_seq[ti];
                if (QC) {
                    uint32_t qd = w_qual[ti];
                    /* the reference's four chains stop four short of the end (:2068) */
                    qd = pos0 < Lmain ? qd : PAD4;
                    const double e0 = l_err[qd & 0xFF], e1 = l_err[(qd >> 8) & 0xFF];
                    const double e2 = l_err[(qd >> 16) & 0xFF], e3 = l_err[qd >> 24];
                    acc0 += e0;
                    acc1 += e1;
                    acc2 += e2;
                    acc3 += e3;
                    /* class codes are 0 2 4 6 (ACGT) 8 (other) 14 (padding): G/C have
                       bit1 != bit2, non-ACGT have bit 3 (:1997-2049 counts) */
                    gc_cnt += __popc(((sd >> 1) ^ sd) & 0x02020202u);
                    acgt_cnt += __popc(~sd & 0x08080808u);
                }
                if (!AD) return false;
                if (DFA_LDS) { /* one VALU + one LDS read per base */
                    e[0] = lds_u16(or_byte<0>(st, sd));
                    e[1] = lds_u16(or_byte<1>(e[0], sd));
                    e[2] = lds_u16(or_byte<2>(e[1], sd));
                    e[3] = lds_u16(or_byte<3>(e[2], sd));
                    st = e[3];
                    return max(max(e[0], e[1]), max(e[2], e[3])) >= dfa_hit;
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t cls2 = (sd >> (8 * j)) & 0xFF;
                    e[j] = *(const uint16_t *)((const uint8_t *)P.dfa + (st | cls2));
                    st = e[j] & 0xFFF0u;
                }
                return ((e[0] | e[1] | e[2] | e[3]) & 1u) != 0;
            };
            /* update_adapter_count_array, _qcmodule.c:2643-2672 */
            auto s_hits = [&](uint32_t d, const uint32_t e[4]) {
                const uint32_t pos0 = c0 + d * 4;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (DFA_LDS ? e[j] < dfa_hit : !(e[j] & 1u)) continue;
                    unsigned long long hits = P.dfa_out[(e[j] - dfa_root) >> 4] & ~found;
                    found |= hits;
                    const uint32_t pos = pos0 + j;
                    while (hits) {
                        const int a = __ffsll((long long)hits) - 1;
                        hits &= hits - 1;
                        const uint32_t start = pos - P.ad_len[a] + 1;
                        if (QC && AD && DFA_LDS && P.ad_lds) {
                            atomicAdd(&l_adf[a * hs + start], 1u);
                        } else {
                            atomicAdd(&P.ad_fwd[a * P.ad_cap + start], 1ULL);
                            atomicAdd(&P.ad_rev[a * P.ad_cap + (L - 1 - start)], 1ULL);
                        }
                    }
                }
            };
            auto s_step = [&](uint32_t d) {
                uint32_t e[4];
                if (s_main(d, e)) s_hits(d, e);
            };
            /* phase H of a full group of equally long reads, four row pairs: nothing depends on
               the row but its two tile words (end-anchored tables are derived at the merge) */
            const uint32_t p = c0 + pl;
            auto h_fast = [&](uint32_t rp0) {
                uint32_t *hb = l_hist_base + p, *hp = l_hist_phred + p;
                uint32_t sw[4], qw[4];
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) {
                    const uint32_t rp = rp0 + k;
                    const uint32_t ti = ptile_idx_h(rp, half, h_dw);
                    sw[k] = w_seq[ti];
                    qw[k] = w_qual[ti];
                }
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) {
                    const uint32_t cls = __builtin_amdgcn_ubfe(sw[k], h_sh + 1, 3);
                    const uint32_t bin = min(__builtin_amdgcn_ubfe(qw[k], h_sh, 8) - 33u, 47u) >> 2;
                    atomicAdd(hist_row(hb, cls, hs * 4), 1u);
                    atomicAdd(hist_row(hp, bin, hs * 4), 1u);
                }
            };
            const bool fast_group = QC && !PT && P.uniform_len && (g + 1) * 64 <= P.n;
            const bool pt_group = PT && P.uniform_len && P.uniform_len <= P.lds_len && !P.pos_base && (g + 1) * 64 <= P.n && pt_one_tile;
            /* a chunk inside every read, and inside what the four chains cover (:2068) */
            if ((fast_group || (QC && pt_group)) && c0 + CW <= P.uniform_len &&
                c0 + CW - 4 < 4 * ((P.uniform_len - 1) / 4)) {
                /* The per-read chains of phase S (a table walk with one LDS round trip per base)
                   and the histogram updates of phase H are independent, so they share one
                   instruction stream and phase H fills the waits of phase S.  Step d takes
                   dword d of every row (S) and the rows of block d (H); LDS addresses are
                   computed by hand, one v_xad_u32 each: (swizzle ^ step) + base. */
                uint32_t hv = lds_addr(w_seq);
                const uint32_t hbp = lds_addr(l_hist_base + p), hpp = lds_addr(l_hist_phred + p);
                double pt_run = 0.0; /* PerTileQuality, a group of one tile: this position's error rates */
#pragma unroll 1
                for (uint32_t d = 0; d < ROW_WORDS; d++) {
                    const uint32_t sa = xor_add(s_sw4, 4 * d, s_row);
                    const uint32_t sd = lds_u32(sa), qd = lds_u32(sa + 4 * QUAL_WORDS);
                    /* l_err is at LDS address 0 (checked at the top): byte << 3 is the address */
                    const double e0 = lds_f64(shl3_byte<0>(qd)), e1 = lds_f64(shl3_byte<1>(qd));
                    const double e2 = lds_f64(shl3_byte<2>(qd)), e3 = lds_f64(shl3_byte<3>(qd));
                    acc0 += e0;
                    acc1 += e1;
                    acc2 += e2;
                    acc3 += e3;
                    gc_cnt += __popc(((sd >> 1) ^ sd) & 0x02020202u);
                    acgt_cnt += __popc(~sd & 0x08080808u);
                    uint32_t e[4];
                    bool hit = false;
                    if (AD && DFA_LDS) {
                        e[0] = lds_u16(or_byte<0>(st, sd));
                        e[1] = lds_u16(or_byte<1>(e[0], sd));
                        e[2] = lds_u16(or_byte<2>(e[1], sd));
                        e[3] = lds_u16(or_byte<3>(e[2], sd));
                        st = e[3];
                        hit = max(max(e[0], e[1]), max(e[2], e[3])) >= dfa_hit;
                    } else if (AD) {
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            e[j] = *(const uint16_t *)((const uint8_t *)P.dfa + (st | ((sd >> (8 * j)) & 0xFF)));
                            st = e[j] & 0xFFF0u;
                        }
                        hit = ((e[0] | e[1] | e[2] | e[3]) & 1u) != 0;
                    }
                    /* phase H: rows 8 d .. 8 d + 7 = block d, two rows per step */
                    const uint32_t sx = (8 * d) & 28;
                    const uint32_t ha = xor_add(h_u4, sx, hv), hb = xor_add(h_u4, sx | 4, hv);
                    hv += 8 * ROW_WORDS * 4 * 2;
                    uint32_t sw[4], qw[4];
                    sw[0] = lds_u32(ha);       qw[0] = lds_u32(ha + 256);
                    sw[1] = lds_u32(ha + 64);  qw[1] = lds_u32(ha + 64 + 256);
                    sw[2] = lds_u32(hb + 128); qw[2] = lds_u32(hb + 128 + 256);
                    sw[3] = lds_u32(hb + 192); qw[3] = lds_u32(hb + 192 + 256);
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++) {
                        const uint32_t cls = __builtin_amdgcn_ubfe(sw[k], h_sh + 1, 3);
                        const uint32_t bin = min(__builtin_amdgcn_ubfe(qw[k], h_sh, 8) - 33u, 47u) >> 2;
                        lds_inc(hbp + __umul24(cls, hs * 4));
                        lds_inc(hpp + __umul24(bin, hs * 4));
                        if (PT) pt_run += lds_f64(__builtin_amdgcn_ubfe(qw[k], h_sh, 8) << 3);
                    }
                    if (hit) s_hits(d, e);
                }
                if (PT) unsafeAtomicAdd(&w_pt[p], pt_run); /* two lanes per position: even rows, odd rows */
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                continue;
            }
            if (QC || AD) {
                const uint32_t nd = min(ROW_WORDS, (stripe_end - c0 + 3) / 4); /* dwords any read still has */
                for (uint32_t d = 0; d < nd; d++) s_step(d);
            }

            /* ---------------- phase H: lane = position, two rows at a time ---------------- */
            if (QC || PT) {
                uint32_t *hb = l_hist_base + (p - P.pos_base);   /* + class * hs */
                uint32_t *hp = l_hist_phred + (p - P.pos_base);
                if (fast_group) {
                    if (p < P.uniform_len) {
#pragma unroll
                        for (uint32_t rp0 = 0; rp0 < 32; rp0 += 4) h_fast(rp0);
                    }
                } else if (PT && pt_group) {
                    /* PerTileQuality on a full group of equally long reads of ONE tile (what the
                       tile-sorted order makes of nearly every group): no row asks anything, a
                       lane sums the error rates of its position over the 64 reads in a register
                       and adds them to the tile's row once */
                    if (p < P.uniform_len) {
                        double run = 0.0;
#pragma unroll 2
                        for (uint32_t rp0 = 0; rp0 < 32; rp0 += 4) {
                            uint32_t sw[4], qw[4];
#pragma unroll
                            for (uint32_t k = 0; k < 4; k++) {
                                const uint32_t ti = ptile_idx_h(rp0 + k, half, h_dw);
                                sw[k] = QC ? w_seq[ti] : 0;
                                qw[k] = w_qual[ti];
                            }
#pragma unroll
                            for (uint32_t k = 0; k < 4; k++) {
                                const uint32_t qb = __builtin_amdgcn_ubfe(qw[k], h_sh, 8);
                                if (QC) {
                                    const uint32_t cls = __builtin_amdgcn_ubfe(sw[k], h_sh + 1, 3);
                                    const uint32_t bin = min(qb - 33u, 47u) >> 2;
                                    atomicAdd(hist_row(hb, cls, hs * 4), 1u);
                                    atomicAdd(hist_row(hp, bin, hs * 4), 1u);
                                }
                                run += l_err[qb < 128 ? qb : 0];
                            }
                        }
                        unsafeAtomicAdd(&w_pt[p], run);
                    }
                } else {
                    const bool in_lds = p - P.pos_base < P.lds_len;
                    /* PerTileQuality: the records come sorted by tile (P.order), so a lane
                       keeps the running error sum of its position in a register and only
                       touches memory when the tile changes */
                    int32_t run_slot = -1;
                    double run_sum = 0.0;
                    for (uint32_t rp0 = 0; rp0 < 32; rp0 += 4) {
                        /* rows of this group that are still inside their read (uniform) */
                        {
                            const uint32_t La = (uint32_t)__builtin_amdgcn_readlane((int)L, (int)(2 * rp0));
                            const uint32_t Lb = (uint32_t)__builtin_amdgcn_readlane((int)L, (int)(2 * rp0 + 1));
                            const uint32_t Lc = (uint32_t)__builtin_amdgcn_readlane((int)L, (int)(2 * rp0 + 2));
                            const uint32_t Ld = (uint32_t)__builtin_amdgcn_readlane((int)L, (int)(2 * rp0 + 3));
                            const uint32_t Le = (uint32_t)__builtin_amdgcn_readlane((int)L, (int)(2 * rp0 + 4));
                            const uint32_t Lf = (uint32_t)__builtin_amdgcn_readlane((int)L, (int)(2 * rp0 + 5));
                            const uint32_t Lg = (uint32_t)__builtin_amdgcn_readlane((int)L, (int)(2 * rp0 + 6));
                            const uint32_t Lh = (uint32_t)__builtin_amdgcn_readlane((int)L, (int)(2 * rp0 + 7));
                            if (c0 >= max(max(max(La, Lb), max(Lc, Ld)), max(max(Le, Lf), max(Lg, Lh)))) continue;
                        }
                        uint32_t sw[4], qw[4];
#pragma unroll
                        for (uint32_t k = 0; k < 4; k++) {
                            const uint32_t rp = rp0 + k;
                            const uint32_t ti = ptile_idx_h(rp, half, h_dw);
                            sw[k] = QC ? w_seq[ti] : 0;
                            qw[k] = w_qual[ti];
                        }
#pragma unroll
                        for (uint32_t k = 0; k < 4; k++) {
                            const uint32_t rp = rp0 + k;
                            const uint32_t L_even = (uint32_t)__builtin_amdgcn_readlane((int)L, (int)(2 * rp));
                            const uint32_t L_odd = (uint32_t)__builtin_amdgcn_readlane((int)L, (int)(2 * rp + 1));
                            const uint32_t Lr = half ? L_odd : L_even;
                            const bool act = p < Lr;
                            const uint32_t qb = (qw[k] >> h_sh) & 0xFF;
                            if (QC) {
                                const uint32_t cls = ((sw[k] >> h_sh) & 0xFF) >> 1;
                                /* an invalid byte makes the pass fail anyway, so it may land in any bin */
                                const uint32_t bin = min(qb - 33u, 47u) >> 2;
                                if (act) {
                                    if (in_lds) {
                                        atomicAdd(hist_row(hb, cls, hs * 4), 1u);
                                        atomicAdd(hist_row(hp, bin, hs * 4), 1u);
                                    } else {
                                        atomicAdd(&P.qc_base[(uint64_t)p * 5 + min(cls, 4u)], 1ULL);
                                        atomicAdd(&P.qc_phred[(uint64_t)p * 12 + bin], 1ULL);
                                    }
                                    if (ea_atomics) {
                                        /* last min(ea, L) positions, right aligned (:1971-1972) */
                                        const uint32_t ean = min(P.ea_len, Lr);
                                        if (p >= Lr - ean) {
                                            const uint32_t e = P.ea_len - Lr + p;
                                            if (P.ea_in_lds) {
                                                atomicAdd(hist_row(l_ea_base + e, cls, es * 4), 1u);
                                                atomicAdd(hist_row(l_ea_phred + e, bin, es * 4), 1u);
                                            } else {
                                                atomicAdd(&P.qc_ea_base[(uint64_t)e * 5 + min(cls, 4u)], 1ULL);
                                                atomicAdd(&P.qc_ea_phred[(uint64_t)e * 12 + bin], 1ULL);
                                            }
                                        }
                                    }
                                }
                            }
                            if (PT) {
                                const int32_t s_even = __builtin_amdgcn_readlane(pt_slot, (int)(2 * rp));
                                const int32_t s_odd = __builtin_amdgcn_readlane(pt_slot, (int)(2 * rp + 1));
                                const int32_t slot = (half ? s_odd : s_even);
                                if (act && slot >= 0) {
                                    if (slot != run_slot) {
                                        if (run_slot >= 0)
                                            unsafeAtomicAdd(&P.pt_errors[(uint64_t)run_slot * P.pt_cap + p], run_sum);
                                        run_slot = slot;
                                        run_sum = 0.0;
                                    }
                                    run_sum += l_err[qb < 128 ? qb : 0];
                                }
                            }
                        }
                    }
                    if (PT && run_slot >= 0)
                        unsafeAtomicAdd(&P.pt_errors[(uint64_t)run_slot * P.pt_cap + p], run_sum);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }

        /* ---------------- per-read epilogue: lane = read ---------------- */
        if (valid && L > P.pos_end) { /* the read goes on in the next stripe */
            sq_carry cy;
            cy.acc[0] = acc0; cy.acc[1] = acc1; cy.acc[2] = acc2; cy.acc[3] = acc3;
            cy.found = found; cy.st = st - dfa_root; cy.gc_cnt = gc_cnt; cy.acgt_cnt = acgt_cnt; cy.pad_ = 0;
            P.carry[slot_index] = cy;
        }
        if (QC && valid && L <= P.pos_end) {
            double total = acc0 + acc1 + acc2 + acc3; /* :2098-2099, left to right */
            for (uint32_t pos = Lmain; pos < L; pos++) { /* :2100-2112 */
                const uint32_t qb = P.buf[qoff + pos];
                total += l_err[qb < 128 ? qb : 0];
            }
            P.metas[r].accumulated_error_rate = total;  /* :2126 */
            if (total != total) atomicMin(P.qc_first_bad, (unsigned long long)(P.first_read_index + r));
            if (acgt_cnt > 0) { /* :2051-2058 */
                const double pct = (double)gc_cnt * 100.0 / (double)acgt_cnt;
                atomicAdd(&l_gc[(uint32_t)round(pct)], 1u);
            }
            if (L > 0) { /* :2127-2137 via the host-libm threshold table */
                const double avg = total / (double)L;
                uint32_t lo = 0, hi = 93; /* largest k with avg <= thr[k]; thr[0] = +inf */
                while (lo < hi) {
                    const uint32_t mid = (lo + hi + 1) >> 1;
                    if (avg <= l_thr[mid]) lo = mid; else hi = mid - 1;
                }
                atomicAdd(&l_ps[lo], 1u);
            }
        }
        if (PT && pt_one_tile && P.uniform_len && !P.pos_base && P.pos_end >= P.uniform_len && pt_acc_slot == pt_group_slot)
            pt_acc_reads += 64; /* one device atomic per tile and wave instead of one per read on one address */
        else if (PT && pt_on && L > 0 && L <= P.pos_end)
            atomicAdd(&P.pt_len_counts[(uint64_t)pt_slot * P.pt_cap + (L - 1)], 1ULL);
    }

    if (PT) pt_flush();
    /* ---------------- merge the workgroup's histograms ---------------- */
    if (QC && AD && DFA_LDS && P.ad_lds) {
        __syncthreads();
        for (uint32_t i = tid; i < P.ad_lds * hs; i += WG_THREADS) {
            const uint32_t v = l_adf[i];
            if (!v) continue;
            const uint32_t a = i / hs, start = i % hs;
            atomicAdd(&P.ad_fwd[a * P.ad_cap + start], (unsigned long long)v);
            atomicAdd(&P.ad_rev[a * P.ad_cap + (P.uniform_len - 1 - start)], (unsigned long long)v);
        }
    }
    if (QC) {
        __syncthreads();
        for (uint32_t i = tid; i < hs * BASE_COLS; i += WG_THREADS) {
            const uint32_t v = l_hist_base[i], c = i / hs, pos = i % hs;
            if (v) {
                atomicAdd(&P.qc_base[(uint64_t)(P.pos_base + pos) * 5 + c], (unsigned long long)v);
                if (P.uniform_len) {
                    /* every read has the same length: the end-anchored table is a
                       window of the positional one (:1971-1972, 2034-2043) */
                    const uint32_t ean = min(P.ea_len, P.uniform_len);
                    if (pos >= P.uniform_len - ean)
                        atomicAdd(&P.qc_ea_base[(uint64_t)(P.ea_len - P.uniform_len + pos) * 5 + c],
                                  (unsigned long long)v);
                }
            }
        }
        for (uint32_t i = tid; i < hs * PHRED_COLS; i += WG_THREADS) {
            const uint32_t v = l_hist_phred[i], c = i / hs, pos = i % hs;
            if (v) {
                atomicAdd(&P.qc_phred[(uint64_t)(P.pos_base + pos) * 12 + c], (unsigned long long)v);
                if (P.uniform_len) {
                    const uint32_t ean = min(P.ea_len, P.uniform_len);
                    if (pos >= P.uniform_len - ean)
                        atomicAdd(&P.qc_ea_phred[(uint64_t)(P.ea_len - P.uniform_len + pos) * 12 + c],
                                  (unsigned long long)v);
                }
            }
        }
        for (uint32_t i = tid; i < es * BASE_COLS; i += WG_THREADS) {
            const uint32_t v = l_ea_base[i];
            if (v) atomicAdd(&P.qc_ea_base[(uint64_t)(i % es) * 5 + i / es], (unsigned long long)v);
        }
        for (uint32_t i = tid; i < es * PHRED_COLS; i += WG_THREADS) {
            const uint32_t v = l_ea_phred[i];
            if (v) atomicAdd(&P.qc_ea_phred[(uint64_t)(i % es) * 12 + i / es], (unsigned long long)v);
        }
        for (uint32_t i = tid; i < 101; i += WG_THREADS)
            if (l_gc[i]) atomicAdd(&P.qc_gc[i], (unsigned long long)l_gc[i]);
        for (uint32_t i = tid; i < 94; i += WG_THREADS)
            if (l_ps[i]) atomicAdd(&P.qc_ps[i], (unsigned long long)l_ps[i]);
    }
}

/* ---- long reads: k_read_sums + k_seg + k_adapter_first ---------------------------------
 * A wave of k_pass walks 64 reads front to back, so a batch is as slow as its longest read
 * (a 100 kb read is 3125 chunks of ~4 us).  Only two things are sequential per read: the f64
 * error sum (exact order, :2062-2112) and the first-occurrence rule of AdapterCounter
 * (:2657-2668).  So for long reads
 *   k_read_sums     one lane per read streams the read once: the four f64 chains, the GC and
 *                   ACGT counts, the per-read bins and accumulated_error_rate;
 *   k_seg           everything per position, over SEGMENTS of SEG positions: a workgroup owns
 *                   segment w of a block of reads (LDS histograms of SEG positions), all
 *                   segments of all reads are in flight at once.  The automaton is restarted
 *                   64 positions in front of the segment (a pattern is at most 64 long), a
 *                   match belongs to the segment its last base lies in, and what a segment
 *                   finds is only a candidate: atomicMin of the start per (read, adapter);
 *   k_adapter_first turns the surviving first occurrences into the count tables.
 * Records come longest first (P.order), so the reads that reach segment w are a prefix. */
constexpr uint32_t SEG = 256;          /* positions per segment */
constexpr uint32_t SEG_WARMUP = 64;    /* restart distance of the automaton: the longest pattern */

struct SegParams {
    const uint32_t *wg_table; /* per workgroup: segment, first group, number of groups */
    const unsigned long long *seg_reads; /* [segments] reads longer than SEG * w */
    unsigned int *first;      /* [records][n_adapters]: start of the first occurrence, ~0: none */
    uint32_t n_adapters;
};

/* Four lanes per read (16 reads per wave): per step the quad loads 64 consecutive bytes of
 * the read's qualities (and, GC, of its sequence: whole sectors, where a lane-private stream would
 * fetch 16-byte pieces); lane c of the quad then adds the error rates of positions = c (mod 4), in
 * position order, to chain c -- the four interleaved f64 chains of the reference (:2062-2097), one per
 * lane -- taking byte c of the quad's sixteen quality dwords as the other lanes hand them round by DPP
 * (round 3; through an LDS buffer of error rates before: three LDS operations per position instead of one). */
constexpr uint32_t SUMS_QUAD_STRIDE = 66; /* doubles per quad in LDS: 64 + padding against bank conflicts */

/* the value lane `from` (0..3, a constant after unrolling) of each quad holds: DPP quad_perm broadcast */
__device__ __forceinline__ uint32_t quad_lane_u32(uint32_t v, int from)
{
    switch (from) {
    case 0: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x00, 0xF, 0xF, true);
    case 1: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x55, 0xF, 0xF, true);
    case 2: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xAA, 0xF, 0xF, true);
    default: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xFF, 0xF, 0xF, true);
    }
}

/* GC = false: the sequences are not read at all (k_span<LONG> counts G/C per read on its way) */
template <bool GC>
__global__ void __launch_bounds__(256) k_read_sums(PassParams P)
{
    __shared__ double l_err[256], l_thr[96];   /* by raw quality byte: NaN for what is no phred character (also >= 128: BAM qualities + 33) */
    __shared__ uint32_t l_gc[104], l_ps[96];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 256; i += 256) {
        double e = __longlong_as_double(0x7FF8000000000000LL);
        if (i >= 33 && i <= 33 + SQ_PHRED_MAX) e = __longlong_as_double((long long)c_error_rate_bits[i - 33]);
        l_err[i] = e;
    }
    for (int i = tid; i < 96; i += 256) l_thr[i] = i < 94 ? P.thresholds[i] : 0.0;
    for (int i = tid; i < 104; i += 256) l_gc[i] = 0;
    for (int i = tid; i < 96; i += 256) l_ps[i] = 0;
    __syncthreads();
    const uint32_t quad = (uint32_t)lane >> 2, c = (uint32_t)lane & 3;
    const uint64_t n_waves = (uint64_t)gridDim.x * 4, wave_id = (uint64_t)blockIdx.x * 4 + (tid >> 6);
    for (uint64_t first = wave_id * 16; first < P.n; first += n_waves * 16) {
        const uint64_t slot = first + quad;
        const bool valid = slot < P.n;
        const uint64_t r = valid ? (P.order ? P.order[slot] : slot) : 0;
        sq_meta m;
        if (valid) m = P.metas[r];
        const uint32_t L = valid ? m.sequence_length : 0;
        const uint64_t soff = valid ? m.record_start + m.sequence_offset : 0;
        const uint64_t qoff = valid ? m.record_start + m.qualities_offset : 0;
        const uint32_t Lmain = L > 0 ? 4 * ((L - 1) / 4) : 0; /* :2062,2068 */
        const uint32_t maxL = wave_max_u32(L);
        const uint32_t minLmain = ~wave_max_u32(valid ? ~Lmain : 0u);
        double acc = 0.0;                   /* chain c of this read */
        uint32_t gc_cnt = 0, acgt_cnt = 0;  /* this lane's share, :1997-2049 */
        /* a step = 128 bytes of every read of the wave: whole lines, each asked for once (64 bytes per step left
           the second half of a line to the caches, which 24 waves x 16 reads per CU do not hold) */
        uint4 cs[2], cq[2], ns[2], nq[2];
        auto load_piece = [&](uint32_t pos, uint4 &sv, uint4 &qv) {
            if constexpr (GC) {
                sv = make_uint4(0, 0, 0, 0);
                qv = make_uint4(PAD4, PAD4, PAD4, PAD4);
                if (pos < L) {
                    sv = load16(P.buf, soff + pos, P.buf_len);
                    qv = load16(P.buf, qoff + pos, P.buf_len);
                }
            } else {
                /* no branch around the load: hipcc waits for a load it cannot count with vmcnt(0), and that drained
                   the loads of the NEXT step in front of every step.  Behind a read's end the 16 bytes are whatever
                   follows (the batch owns 64 spare bytes behind its text: sq_fused_add_batch sends only such
                   batches here); nothing looks at them (`inside`, Lmain). */
                sv = make_uint4(0, 0, 0, 0);
                __builtin_memcpy(&qv, P.buf + qoff + min(pos, L), 16);
            }
        };
        load_piece(16 * c, cs[0], cq[0]);
        load_piece(16 * c + 64, cs[1], cq[1]);
        /* two steps per turn of the loop, the buffers swapping roles: a copy `current = next` at the end of a step
           would have to wait for the loads of the next step there */
        auto step = [&](uint4 (&cs)[2], uint4 (&cq)[2], uint4 (&ns)[2], uint4 (&nq)[2], uint32_t base0) {
            if (!GC || base0 + 128 < maxL) {   /* without a condition where the load has none (behind the end it reads the clamped address again): a load on one path only cannot be waited for by count */
                load_piece(base0 + 128 + 16 * c, ns[0], nq[0]);
                load_piece(base0 + 192 + 16 * c, ns[1], nq[1]);
            }
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const uint32_t base = base0 + 64 * half, p0 = base + 16 * c;
                const uint32_t qw[4] = {cq[half].x, cq[half].y, cq[half].z, cq[half].w};
                if constexpr (GC) {
                    const uint32_t sw[4] = {cs[half].x, cs[half].y, cs[half].z, cs[half].w};
#pragma unroll
                    for (int d = 0; d < 4; d++) {
                        const uint32_t at = p0 + 4 * d;
                        if (at < L) {
                            const uint32_t sd = pad_tail(cls2_of_dword(sw[d]), (int)(L - at), CLS2_PAD4);
                            gc_cnt += __popc(((sd >> 1) ^ sd) & 0x02020202u);
                            acgt_cnt += __popc(~sd & 0x08080808u);
                        }
                    }
                }
                /* chain c: positions c, c + 4, ..., c + 60 of the 64, in that order: byte c of the quad's 16
                   quality dwords (lane cc holds dwords 4 cc .. 4 cc + 3), handed round by DPP; positions behind
                   Lmain (a multiple of four, like a dword's first position) add +0.0: the chains stop there */
                double e[16];
                if (base + 64 <= minLmain) {   /* no read of the wave ends here (sorted by length: all but the last step or two) */
#pragma unroll
                    for (int cc = 0; cc < 4; cc++)
#pragma unroll
                        for (int mm = 0; mm < 4; mm++)
                            e[4 * cc + mm] = l_err[__builtin_amdgcn_ubfe(quad_lane_u32(qw[mm], cc), 8 * c, 8)];
                } else {
#pragma unroll
                    for (int cc = 0; cc < 4; cc++)
#pragma unroll
                        for (int mm = 0; mm < 4; mm++) {
                            const uint32_t w = quad_lane_u32(qw[mm], cc);
                            const bool inside = base + 16 * cc + 4 * mm < Lmain;
                            e[4 * cc + mm] = inside ? l_err[(w >> (8 * c)) & 0xFFu] : 0.0;
                        }
                }
#pragma unroll
                for (int i = 0; i < 16; i++) acc += e[i];
            }
        };
        for (uint32_t base0 = 0; base0 < maxL; base0 += 256) {
            step(cs, cq, ns, nq, base0);
            if (base0 + 128 < maxL) step(ns, nq, cs, cq, base0 + 128);
        }
        /* the quad's four chains and counts */
        const double a0 = __shfl(acc, (lane & ~3) + 0), a1 = __shfl(acc, (lane & ~3) + 1);
        const double a2 = __shfl(acc, (lane & ~3) + 2), a3 = __shfl(acc, (lane & ~3) + 3);
        gc_cnt += __shfl_xor(gc_cnt, 1); gc_cnt += __shfl_xor(gc_cnt, 2);
        acgt_cnt += __shfl_xor(acgt_cnt, 1); acgt_cnt += __shfl_xor(acgt_cnt, 2);
        if (valid && c == 0) {
            double total = a0 + a1 + a2 + a3; /* :2098-2099, left to right */
            for (uint32_t pos = Lmain; pos < L; pos++) { /* :2100-2112 */
                const uint32_t qb = P.buf[qoff + pos];
                total += l_err[qb];
            }
            P.metas[r].accumulated_error_rate = total; /* :2126 */
            if (total != total) atomicMin(P.qc_first_bad, (unsigned long long)(P.first_read_index + r));
            if (GC && acgt_cnt > 0) atomicAdd(&l_gc[(uint32_t)round((double)gc_cnt * 100.0 / (double)acgt_cnt)], 1u);
            if (L > 0) {
                const double avg = total / (double)L;
                uint32_t lo = 0, hi = 93;
                while (lo < hi) {
                    const uint32_t mid = (lo + hi + 1) >> 1;
                    if (avg <= l_thr[mid]) lo = mid; else hi = mid - 1;
                }
                atomicAdd(&l_ps[lo], 1u);
            }
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < 101; i += 256)
        if (l_gc[i]) atomicAdd(&P.qc_gc[i], (unsigned long long)l_gc[i]);
    for (uint32_t i = tid; i < 94; i += 256)
        if (l_ps[i]) atomicAdd(&P.qc_ps[i], (unsigned long long)l_ps[i]);
}


/* QCMetrics' GC histogram alone (:1997-2060), a wave per read: the error path behind k_read_sums<false>
 * (a batch with an invalid phred byte keeps k_seg, which counts no G/C) */
__global__ void __launch_bounds__(256) k_read_gc(PassParams P)
{
    __shared__ uint32_t l_gc[104];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 104; i += 256) l_gc[i] = 0;
    __syncthreads();
    const uint64_t n_waves = (uint64_t)gridDim.x * 4;
    for (uint64_t r = (uint64_t)blockIdx.x * 4 + (tid >> 6); r < P.n; r += n_waves) {
        const sq_meta m = P.metas[r];
        const uint32_t L = m.sequence_length;
        const uint64_t soff = m.record_start + m.sequence_offset;
        uint32_t gc_cnt = 0, acgt_cnt = 0;
        for (uint32_t pos = 4 * lane; pos < L; pos += 256) {
            uint32_t w = 0;
            for (uint32_t k = 0; k < 4 && pos + k < L; k++) w |= (uint32_t)P.buf[soff + pos + k] << (8 * k);
            const uint32_t sd = pad_tail(cls2_of_dword(w), (int)(L - pos), CLS2_PAD4);
            gc_cnt += __popc(((sd >> 1) ^ sd) & 0x02020202u);
            acgt_cnt += __popc(~sd & 0x08080808u);
        }
        for (int off = 32; off > 0; off >>= 1) { gc_cnt += __shfl_xor(gc_cnt, off); acgt_cnt += __shfl_xor(acgt_cnt, off); }
        if (lane == 0 && acgt_cnt > 0) atomicAdd(&l_gc[(uint32_t)round((double)gc_cnt * 100.0 / (double)acgt_cnt)], 1u);
    }
    __syncthreads();
    for (uint32_t i = tid; i < 101; i += 256)
        if (l_gc[i]) atomicAdd(&P.qc_gc[i], (unsigned long long)l_gc[i]);
}

constexpr int SEG_DEPTH = 2; /* chunks of loads in flight per wave (3: -2 %, 4 costs a wave per SIMD: -15 %) */
template <bool AD>
__global__ void __launch_bounds__(WG_THREADS, 3) k_seg(PassParams P, SegParams S)
{
    extern __shared__ __align__(16) uint8_t smem[];
    uint16_t *l_dfa = (uint16_t *)smem;                    /* in front: rows are addressed with 16 bits (see k_pass) */
    uint32_t *l_wave = (uint32_t *)(l_dfa + (AD ? P.dfa_states * 8 : 0)); /* per wave: tiles, offsets, lengths */
    constexpr uint32_t hs = SEG;
    uint32_t *l_hist_base = l_wave + WAVES * WAVE_WORDS;   /* [5][SEG] */
    uint32_t *l_hist_phred = l_hist_base + hs * BASE_COLS; /* [12][SEG] */
    uint32_t *l_ea_base = l_hist_phred + hs * PHRED_COLS;
    const uint32_t ea_rows = P.ea_in_lds ? P.ea_len : 0, es = hist_stride(ea_rows);
    uint32_t *l_ea_phred = l_ea_base + es * BASE_COLS;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (uint32_t i = tid; i < (hs + es) * (BASE_COLS + PHRED_COLS); i += WG_THREADS) l_hist_base[i] = 0;
    const uint32_t dfa_root = AD ? lds_addr(l_dfa) : 0, dfa_hit = dfa_root + P.dfa_accept * 16;
    if (AD)
        for (uint32_t i = tid; i < P.dfa_states * 8; i += WG_THREADS)
            l_dfa[i] = (uint16_t)((P.dfa[i] & 0xFFF0u) + dfa_root);
    __syncthreads();

    uint32_t *w_seq = l_wave + wave * WAVE_WORDS, *w_qual = w_seq + TILE_WORDS;
    unsigned long long *w_soff = (unsigned long long *)(w_qual + TILE_WORDS), *w_qoff = w_soff + 64;
    uint32_t *w_len = (uint32_t *)(w_qoff + 64);
    const uint32_t half = (uint32_t)lane >> 5, pl = (uint32_t)lane & 31;
    const uint32_t h_sh = 8 * (pl & 3), h_dw = pl >> 2, row_base = half * ROW_WORDS;

    const uint32_t seg = S.wg_table[3 * blockIdx.x], g_first = S.wg_table[3 * blockIdx.x + 1];
    const uint32_t g_count = S.wg_table[3 * blockIdx.x + 2];
    const uint32_t pos_base = seg * SEG, pos_stop = pos_base + SEG;
    const uint64_t n_here = S.seg_reads[seg]; /* reads longer than pos_base: a prefix of the order */

    for (uint32_t gi = wave; gi < g_count; gi += WAVES) {
        const uint64_t slot_index = (uint64_t)(g_first + gi) * 64 + lane;
        const bool valid = slot_index < n_here;
        const uint64_t r = valid ? P.order[slot_index] : 0;
        sq_meta m;
        if (valid) m = P.metas[r];
        const uint32_t L = valid ? m.sequence_length : 0;
        w_soff[lane] = valid ? m.record_start + m.sequence_offset : 0;
        w_qoff[lane] = valid ? m.record_start + m.qualities_offset : 0;
        w_len[lane] = L;
        const uint32_t stop = min(wave_max_u32(L), pos_stop);
        uint32_t st = dfa_root;
        unsigned long long found = 0; /* adapters this segment has reported */
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        /* LDS leaves this kernel three waves per SIMD (168 VGPRs): the loads run SEG_DEPTH chunks
           ahead of the counting */
        uint4 pf_s[SEG_DEPTH][2], pf_q[SEG_DEPTH][2];
        auto fetch = [&](uint32_t c0, uint4 (&fs)[2], uint4 (&fq)[2]) {
#pragma unroll
            for (int it = 0; it < 2; it++) {
                const uint32_t row = it * 32 + ((uint32_t)lane >> 1), p0 = c0 + ((uint32_t)lane & 1) * 16;
                fs[it] = make_uint4(0, 0, 0, 0);
                fq[it] = make_uint4(PAD4, PAD4, PAD4, PAD4);
                if (p0 < w_len[row]) {
                    fs[it] = load16(P.buf, w_soff[row] + p0, P.buf_len);
                    if (c0 >= pos_base) fq[it] = load16(P.buf, w_qoff[row] + p0, P.buf_len);
                }
            }
        };
        /* the automaton starts SEG_WARMUP positions early (not in segment 0) */
        const uint32_t c_begin = (AD && seg) ? pos_base - SEG_WARMUP : pos_base;
#pragma unroll
        for (int k = 0; k < SEG_DEPTH; k++)
            if (c_begin + k * CW < stop) fetch(c_begin + k * CW, pf_s[k], pf_q[k]);
        for (uint32_t c0 = c_begin; c0 < stop; c0 += CW) {
            const bool warm = c0 < pos_base;
#pragma unroll
            for (int it = 0; it < 2; it++) {
                const uint32_t row = it * 32 + ((uint32_t)lane >> 1), piece = (uint32_t)lane & 1;
                const uint32_t Lr = w_len[row], p0 = c0 + piece * 16;
                uint4 sv = make_uint4(CLS2_PAD4, CLS2_PAD4, CLS2_PAD4, CLS2_PAD4), qv = pf_q[0][it];
                if (p0 < Lr) {
                    const int nv = (int)min(16u, Lr - p0);
                    sv.x = cls2_of_dword(pf_s[0][it].x); sv.y = cls2_of_dword(pf_s[0][it].y);
                    sv.z = cls2_of_dword(pf_s[0][it].z); sv.w = cls2_of_dword(pf_s[0][it].w);
                    if (nv < 16) {
                        sv.x = pad_tail(sv.x, nv, CLS2_PAD4); sv.y = pad_tail(sv.y, nv - 4, CLS2_PAD4);
                        sv.z = pad_tail(sv.z, nv - 8, CLS2_PAD4); sv.w = pad_tail(sv.w, nv - 12, CLS2_PAD4);
                        qv.x = pad_tail(qv.x, nv, PAD4); qv.y = pad_tail(qv.y, nv - 4, PAD4);
                        qv.z = pad_tail(qv.z, nv - 8, PAD4); qv.w = pad_tail(qv.w, nv - 12, PAD4);
                    }
                }
                const uint32_t d0 = piece * 4;
                w_seq[tile_idx(row, d0 + 0)] = sv.x; w_seq[tile_idx(row, d0 + 1)] = sv.y;
                w_seq[tile_idx(row, d0 + 2)] = sv.z; w_seq[tile_idx(row, d0 + 3)] = sv.w;
                w_qual[tile_idx(row, d0 + 0)] = qv.x; w_qual[tile_idx(row, d0 + 1)] = qv.y;
                w_qual[tile_idx(row, d0 + 2)] = qv.z; w_qual[tile_idx(row, d0 + 3)] = qv.w;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k + 1 < SEG_DEPTH; k++)
#pragma unroll
                for (int it = 0; it < 2; it++) { pf_s[k][it] = pf_s[k + 1][it]; pf_q[k][it] = pf_q[k + 1][it]; }
            if (c0 + SEG_DEPTH * CW < stop) fetch(c0 + SEG_DEPTH * CW, pf_s[SEG_DEPTH - 1], pf_q[SEG_DEPTH - 1]);

            if (AD) { /* phase S: the automaton only */
                const uint32_t nd = min(ROW_WORDS, (stop - c0 + 3) / 4);
                for (uint32_t d = 0; d < nd; d++) {
                    const uint32_t sd = w_seq[tile_idx((uint32_t)lane, d)];
                    uint32_t e[4];
                    e[0] = lds_u16(or_byte<0>(st, sd));
                    e[1] = lds_u16(or_byte<1>(e[0], sd));
                    e[2] = lds_u16(or_byte<2>(e[1], sd));
                    e[3] = lds_u16(or_byte<3>(e[2], sd));
                    st = e[3];
                    if (!warm && max(max(e[0], e[1]), max(e[2], e[3])) >= dfa_hit) {
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            if (e[j] < dfa_hit) continue;
                            unsigned long long hits = P.dfa_out[(e[j] - dfa_root) >> 4] & ~found;
                            found |= hits;
                            const uint32_t pos = c0 + d * 4 + j;
                            while (hits) {
                                const int a = __ffsll((long long)hits) - 1;
                                hits &= hits - 1;
                                atomicMin(&S.first[r * S.n_adapters + a], pos - P.ad_len[a] + 1);
                            }
                        }
                    }
                }
            }
            if (!warm) { /* phase H: lane = position, two rows at a time */
                const uint32_t p = c0 + pl;
                uint32_t *hb = l_hist_base + (p - pos_base), *hp = l_hist_phred + (p - pos_base);
                for (uint32_t rp0 = 0; rp0 < 32; rp0 += 4) {
                    uint32_t sw[4], qw[4];
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++) {
                        const uint32_t rp = rp0 + k;
                        const uint32_t ti = 2 * rp * ROW_WORDS + row_base + (h_dw ^ ((rp >> 1) & 7));
                        sw[k] = w_seq[ti];
                        qw[k] = w_qual[ti];
                    }
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++) {
                        const uint32_t rp = rp0 + k;
                        const uint32_t L_even = (uint32_t)__builtin_amdgcn_readlane((int)L, (int)(2 * rp));
                        const uint32_t L_odd = (uint32_t)__builtin_amdgcn_readlane((int)L, (int)(2 * rp + 1));
                        const uint32_t Lr = half ? L_odd : L_even;
                        if (p >= Lr) continue;
                        const uint32_t cls = __builtin_amdgcn_ubfe(sw[k], h_sh + 1, 3);
                        const uint32_t bin = min(__builtin_amdgcn_ubfe(qw[k], h_sh, 8) - 33u, 47u) >> 2;
                        atomicAdd(hist_row(hb, cls, hs * 4), 1u);
                        atomicAdd(hist_row(hp, bin, hs * 4), 1u);
                        const uint32_t ean = min(P.ea_len, Lr); /* :1971-1972 */
                        if (p >= Lr - ean) {
                            const uint32_t e = P.ea_len - Lr + p;
                            if (P.ea_in_lds) {
                                atomicAdd(hist_row(l_ea_base + e, cls, es * 4), 1u);
                                atomicAdd(hist_row(l_ea_phred + e, bin, es * 4), 1u);
                            } else {
                                atomicAdd(&P.qc_ea_base[(uint64_t)e * 5 + min(cls, 4u)], 1ULL);
                                atomicAdd(&P.qc_ea_phred[(uint64_t)e * 12 + bin], 1ULL);
                            }
                        }
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < hs * BASE_COLS; i += WG_THREADS) {
        const uint32_t v = l_hist_base[i];
        if (v) atomicAdd(&P.qc_base[(uint64_t)(pos_base + i % hs) * 5 + i / hs], (unsigned long long)v);
    }
    for (uint32_t i = tid; i < hs * PHRED_COLS; i += WG_THREADS) {
        const uint32_t v = l_hist_phred[i];
        if (v) atomicAdd(&P.qc_phred[(uint64_t)(pos_base + i % hs) * 12 + i / hs], (unsigned long long)v);
    }
    for (uint32_t i = tid; i < es * BASE_COLS; i += WG_THREADS) {
        const uint32_t v = l_ea_base[i];
        if (v) atomicAdd(&P.qc_ea_base[(uint64_t)(i % es) * 5 + i / es], (unsigned long long)v);
    }
    for (uint32_t i = tid; i < es * PHRED_COLS; i += WG_THREADS) {
        const uint32_t v = l_ea_phred[i];
        if (v) atomicAdd(&P.qc_ea_phred[(uint64_t)(i % es) * 12 + i / es], (unsigned long long)v);
    }
}

/* update_adapter_count_array (:2643-2672) for the first occurrences k_seg left */
__global__ void k_adapter_first(const unsigned int *first, const sq_meta *metas, uint64_t n, uint32_t n_adapters,
                                const uint8_t *ad_len, unsigned long long *fwd, unsigned long long *rev,
                                uint64_t cap)
{
    const uint64_t cells = n * n_adapters;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < cells;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned int start = first[i];
        if (start == ~0u) continue;
        const uint64_t r = i / n_adapters;
        const uint32_t a = (uint32_t)(i % n_adapters), L = metas[r].sequence_length;
        atomicAdd(&fwd[a * cap + start], 1ULL);
        atomicAdd(&rev[a * cap + (L - 1 - start)], 1ULL);
    }
}

size_t seg_lds_bytes(uint32_t ea_rows, bool ad, uint32_t dfa_states)
{
    size_t b = (size_t)WAVES * WAVE_WORDS * 4 + (size_t)(SEG + hist_stride(ea_rows)) * (BASE_COLS + PHRED_COLS) * 4;
    if (ad) b += (size_t)dfa_states * 16;
    return b + 16;
}

/* ---- k_ring: the fused pass for uniform short reads, every 64-byte sector fetched once ----
 * k_pass fetches a read's 32-position chunk from wherever it lies: a 64-byte sector is
 * touched by two or three chunk loads some 10 us apart, has left the L2 in between and is
 * fetched again (17.0 GB moved for 8.7 GB of records; 32-byte aligned
 * records run 25 % faster through the same kernel).  Here the loads follow memory, not
 * the read:
 *   - a row (read) and stream (sequence / quality) is cut into 32-byte ALIGNED windows;
 *     a lane owns 16 of those bytes (two lanes per row, two row sets, two streams) and
 *     loads the two windows of a 64-byte sector back to back, one of them a step early,
 *     so that the sector is requested once; the later window waits in registers;
 *   - on its way to LDS a window is rotated into position space: whole dwords by the
 *     address it is written to, the last 0-3 bytes by v_alignbyte with the neighbouring
 *     lane's dword (DPP), so that the tile looks exactly as in k_pass;
 *   - the tile is a ring of two windows (64 bytes) per row and stream, because a
 *     32-position chunk straddles two aligned windows.
 * 512 threads share one set of LDS histograms: 8 x 8 KB of rings + 11 KB of histograms is
 * 80 KB, two workgroups (16 waves) per CU.  Conditions (the host checks them):
 * QCMetrics (+ AdapterCounter, automaton in LDS), no PerTileQuality, one length U <= 512 for
 * the whole batch, stored order, full groups; the rest goes through k_pass. */
constexpr int RING_THREADS = 512, RING_WAVES = RING_THREADS / 64;
constexpr uint32_t RING_ROW_WORDS = 16, RING_TILE_WORDS = 64 * RING_ROW_WORDS;

/* ring address of position dword m of a row: the row's 16 dwords are XOR-ed with bits of
 * the row so that "lane = row, same dword" (phase S) hits 32 different banks.  The two streams
 * are interleaved in pairs of rows (128 B of classes, then the 128 B of qualities of the same
 * two rows), so that the quality word is RING_QUAL_WORDS behind the class word (one
 * ds_read2_b32) and row pair rp of phase H starts at 256 rp */
constexpr uint32_t RING_QUAL_WORDS = 2 * RING_ROW_WORDS;
__device__ __forceinline__ uint32_t ring_row(uint32_t row)
{
    return (row >> 1) * 4 * RING_ROW_WORDS + (row & 1u) * RING_ROW_WORDS;
}
__device__ __forceinline__ uint32_t ring_idx(uint32_t row, uint32_t m)
{
    return ring_row(row) + ((m & 15u) ^ ((row >> 1) & 15u));
}

/* value of the neighbouring lane (lane ^ 1) */
__device__ __forceinline__ uint32_t swap_adjacent(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, false);
}

template <bool AD>
__global__ void __launch_bounds__(RING_THREADS, 4) k_ring(PassParams P)
{
    extern __shared__ __align__(16) uint8_t smem[];
    double *l_err = (double *)smem;                        /* [136] */
    double *l_thr = l_err + 136;                           /* [96] */
    uint32_t *l_gc = (uint32_t *)(l_thr + 96);             /* [104] */
    uint32_t *l_ps = l_gc + 104;                           /* [96] */
    uint16_t *l_dfa = (uint16_t *)(l_ps + 96);             /* in front: rows are addressed with 16 bits */
    uint32_t *l_ring = (uint32_t *)(l_dfa + (AD ? P.dfa_states * 8 : 0)); /* [waves][2 * RING_TILE_WORDS] */
    const uint32_t U = P.uniform_len, hs = hist_stride(U);
    uint32_t *l_hist_base = l_ring + RING_WAVES * 2 * RING_TILE_WORDS; /* [5][hs] */
    uint32_t *l_hist_phred = l_hist_base + hs * BASE_COLS;             /* [12][hs] */
    uint32_t *l_adf = l_hist_phred + hs * PHRED_COLS;     /* [ad_lds][hs] adapter hits of this workgroup */

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (lds_addr(l_err) != 0) __builtin_trap(); /* see k_pass: byte << 3 is the address of its error rate */
    const uint32_t dfa_root = AD ? lds_addr(l_dfa) : 0, dfa_hit = dfa_root + P.dfa_accept * 16;
    for (int i = tid; i < 136; i += RING_THREADS) {
        double e;
        if (i >= 33 && i <= 33 + SQ_PHRED_MAX) e = __longlong_as_double((long long)c_error_rate_bits[i - 33]);
        else if (i >= 128) e = 0.0;
        else e = __longlong_as_double(0x7FF8000000000000LL);
        l_err[i] = e;
    }
    for (int i = tid; i < 96; i += RING_THREADS) l_thr[i] = i < 94 ? P.thresholds[i] : 0.0;
    for (int i = tid; i < 104; i += RING_THREADS) l_gc[i] = 0;
    for (int i = tid; i < 96; i += RING_THREADS) l_ps[i] = 0;
    for (uint32_t i = tid; i < hs * (BASE_COLS + PHRED_COLS); i += RING_THREADS) l_hist_base[i] = 0;
    if (AD) { /* entries become the LDS address of the next row, hits are the rows from dfa_hit on */
        for (uint32_t i = tid; i < P.dfa_states * 8; i += RING_THREADS)
            l_dfa[i] = (uint16_t)((P.dfa[i] & 0xFFF0u) + dfa_root);
        for (uint32_t i = tid; i < P.ad_lds * hs; i += RING_THREADS) l_adf[i] = 0;
    }
    __syncthreads();

    uint32_t *w_seq = l_ring + wave * 2 * RING_TILE_WORDS, *w_qual = w_seq + RING_QUAL_WORDS;
    const uint32_t nch = (U + CW - 1) / CW;
    const uint32_t Lmain = 4 * ((U - 1) / 4);
    const uint64_t ngroups = P.n / 64;
    const uint64_t bufaddr = (uint64_t)(uintptr_t)P.buf;
    /* phase H: lanes 0-31 an even row, lanes 32-63 the odd row after it */
    const uint32_t half = (uint32_t)lane >> 5, pl = (uint32_t)lane & 31;
    const uint32_t h_sh = 8 * (pl & 3), h_dw = pl >> 2;
    /* byte addresses of the fused loop: what is fixed per lane */
    const uint32_t s_row = lds_addr(w_seq) + ((uint32_t)lane >> 1) * 256 + ((uint32_t)lane & 1) * 64;
    const uint32_t s_sw4 = 4 * (((uint32_t)lane >> 1) & 15);
    const uint32_t h_row = lds_addr(w_seq) + 64 * half;
    /* loader: rows st_row and st_row + 32, bytes [16 k, 16 k + 16) of every window */
    const uint32_t st_row = (uint32_t)lane >> 1, k16 = ((uint32_t)lane & 1) * 16;

    const uint64_t n_waves = (uint64_t)gridDim.x * RING_WAVES, wave_id = (uint64_t)blockIdx.x * RING_WAVES + wave;
    for (uint64_t g = wave_id; g < ngroups; g += n_waves) {
        const uint64_t r = g * 64 + lane;
        const sq_meta m = P.metas[r];
        const uint64_t soff = m.record_start + m.sequence_offset, qoff = m.record_start + m.qualities_offset;

        /* per (row set, stream): the lane's 16 bytes of the next window to load, how many
           windows hold bytes of the read, by how much the read is off its first window
           (s = 4 t + sb: t whole dwords, handled by the ring address; sb bytes, by
           v_alignbyte) and whether window 0 is the second half of its 64-byte sector.
           A window with at least one byte of the read lies in a mapped page as a whole
           (32-byte aligned), so these loads need no bounds checks; bytes that are not
           the read's never reach a result. */
        const uint8_t *nextp[2][2];
        uint32_t s6[2][2]; /* address & 63: s = s6 & 31, sector half of window 0 = s6 >> 5 (kept packed:
                              registers are what limits this kernel) */
        uint4 R0[2][2], R1[2][2];
        uint32_t last3[2][2];
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const int row = it * 32 + (int)st_row;
            const uint64_t off[2] = {(uint64_t)__shfl((unsigned long long)soff, row),
                                     (uint64_t)__shfl((unsigned long long)qoff, row)};
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const uint64_t addr = bufaddr + off[q];
                const uint32_t s = (uint32_t)addr & 31u;
                nextp[it][q] = (const uint8_t *)(uintptr_t)(addr - s + k16);
                s6[it][q] = (uint32_t)addr & 63u;
                last3[it][q] = 0;
            }
        }
        /* rotate window j into position space and write it to the ring */
        /* rotate window j into position space and write it to the ring.  Every read has U
           bases, so what lies behind the read's end needs no masking here: phase S pads the
           one dword that straddles the end and nothing reads further */
        auto stage = [&](int it, int q, uint32_t j, uint4 v, bool head) {
            const uint32_t sb = s6[it][q] & 3u;
            if (q == 0) {
                v.x = cls2_of_dword(v.x); v.y = cls2_of_dword(v.y);
                v.z = cls2_of_dword(v.z); v.w = cls2_of_dword(v.w);
            }
            /* the dword in front of mine: the other lane's last one, of this window (k = 1)
               or of the window before (k = 0) */
            const uint32_t other_now = swap_adjacent(v.w), other_before = swap_adjacent(last3[it][q]);
            const uint32_t prev = k16 ? other_now : other_before;
            last3[it][q] = v.w;
            uint32_t y[4];
            y[0] = __builtin_amdgcn_alignbyte(v.x, prev, sb);
            y[1] = __builtin_amdgcn_alignbyte(v.y, v.x, sb);
            y[2] = __builtin_amdgcn_alignbyte(v.z, v.y, sb);
            y[3] = __builtin_amdgcn_alignbyte(v.w, v.z, sb);
            /* y[i] = positions 4 m .. 4 m + 3 with m = m0 + i */
            /* position dword of y[0]: window 0 starts s bytes in front of the read */
            const int m0 = (int)(8 * j) + (int)(k16 >> 2) - 1 - (int)((s6[it][q] & 31u) >> 2);
            uint32_t *trow = (q ? w_qual : w_seq) + ring_row((uint32_t)it * 32 + st_row);
            const uint32_t swz = (((uint32_t)it * 32 + st_row) >> 1) & 15u;
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (!head || m0 + i >= 0) trow[((uint32_t)(m0 + i) ^ swz) & 15u] = y[i];
        };

        /* prologue: window 0 goes to the ring; the registers are primed so that window 1 is
           there at step 0: in R1 when it is the second half of window 0's sector, else in
           R0 with its own second half in R1 */
        {
            uint4 T[2][2];
#pragma unroll
            for (int it = 0; it < 2; it++)
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const uint8_t *p0 = nextp[it][q];
                    const uint32_t nw = ((s6[it][q] & 31u) + U + 31u) >> 5; /* windows with bytes of the read */
                    const uint32_t par = s6[it][q] >> 5;
                    T[it][q] = *(const uint4 *)p0;
                    R0[it][q] = make_uint4(0, 0, 0, 0);
                    R1[it][q] = make_uint4(0, 0, 0, 0);
                    if (par) {
                        if (1 < nw) R0[it][q] = *(const uint4 *)(p0 + 32);
                        if (2 < nw) R1[it][q] = *(const uint4 *)(p0 + 64);
                        nextp[it][q] = p0 + 96;
                    } else {
                        if (1 < nw) R1[it][q] = *(const uint4 *)(p0 + 32);
                        nextp[it][q] = p0 + 64;
                    }
                }
#pragma unroll
            for (int it = 0; it < 2; it++)
#pragma unroll
                for (int q = 0; q < 2; q++) stage(it, q, 0, T[it][q], true);
        }

        double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
        uint32_t st = dfa_root, gc_cnt = 0, acgt_cnt = 0;
        unsigned long long found = 0;

        for (uint32_t c = 0; c < nch; c++) {
            /* ---------------- STAGE window c + 1 ---------------- */
            {
#pragma unroll
                for (int it = 0; it < 2; it++)
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        /* second half of its sector: it waited in R1 and the next sector is due */
                        const bool second = (((s6[it][q] >> 5) + c + 1) & 1u) != 0;
                        const uint32_t nw = ((s6[it][q] & 31u) + U + 31u) >> 5;
                        uint4 v;
                        v.x = second ? R1[it][q].x : R0[it][q].x;
                        v.y = second ? R1[it][q].y : R0[it][q].y;
                        v.z = second ? R1[it][q].z : R0[it][q].z;
                        v.w = second ? R1[it][q].w : R0[it][q].w;
                        stage(it, q, c + 1, v, false);
                        if (second) {
                            const uint8_t *pn = nextp[it][q];
                            if (c + 2 < nw) R0[it][q] = *(const uint4 *)pn;
                            if (c + 3 < nw) R1[it][q] = *(const uint4 *)(pn + 32);
                            nextp[it][q] = pn + 64;
                        }
                    }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();

            const uint32_t c0 = c * CW;
            const uint32_t p = c0 + pl;
            const uint32_t mm = (8 * c + h_dw) & 15u;
            /* update_adapter_count_array, _qcmodule.c:2643-2672 */
            auto hits_of = [&](uint32_t pos0, const uint32_t e[4]) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (e[j] < dfa_hit) continue;
                    unsigned long long hits = P.dfa_out[(e[j] - dfa_root) >> 4] & ~found;
                    found |= hits;
                    const uint32_t pos = pos0 + j;
                    while (hits) {
                        const int a = __ffsll((long long)hits) - 1;
                        hits &= hits - 1;
                        const uint32_t start = pos - P.ad_len[a] + 1;
                        if (P.ad_lds) {
                            atomicAdd(&l_adf[a * hs + start], 1u);
                        } else {
                            atomicAdd(&P.ad_fwd[a * P.ad_cap + start], 1ULL);
                            atomicAdd(&P.ad_rev[a * P.ad_cap + (U - 1 - start)], 1ULL);
                        }
                    }
                }
            };
            /* ---------------- phase S: lane = read ----------------
               LDS addresses by hand as in k_pass, one v_xad_u32 each.  Dwords from d_plain on
               need care: the four chains stop four short of the end (:2068) and the dword that
               straddles the end of the read is padded */
            const uint32_t nd = min(ROW_WORDS, (U - c0 + 3) / 4);
            const uint32_t d_plain = Lmain > c0 ? min(ROW_WORDS, (Lmain - c0) / 4) : 0;
#pragma unroll 1
            for (uint32_t d = 0; d < nd; d++) {
                const uint32_t sa = xor_add(s_sw4, 4 * ((8 * c + d) & 15u), s_row);
                uint32_t sd = lds_u32(sa), qd = lds_u32(sa + 4 * RING_QUAL_WORDS);
                if (d >= d_plain) {
                    const uint32_t pos0 = c0 + d * 4;
                    if (pos0 + 4 > U) sd = pad_tail(sd, (int)(U - pos0), CLS2_PAD4);
                    if (pos0 >= Lmain) qd = PAD4;
                }
                const double e0 = lds_f64(shl3_byte<0>(qd)), e1 = lds_f64(shl3_byte<1>(qd));
                const double e2 = lds_f64(shl3_byte<2>(qd)), e3 = lds_f64(shl3_byte<3>(qd));
                acc0 += e0;
                acc1 += e1;
                acc2 += e2;
                acc3 += e3;
                gc_cnt += __popc(((sd >> 1) ^ sd) & 0x02020202u);
                acgt_cnt += __popc(~sd & 0x08080808u);
                if (AD) {
                    uint32_t e[4];
                    e[0] = lds_u16(or_byte<0>(st, sd));
                    e[1] = lds_u16(or_byte<1>(e[0], sd));
                    e[2] = lds_u16(or_byte<2>(e[1], sd));
                    e[3] = lds_u16(or_byte<3>(e[2], sd));
                    st = e[3];
                    if (max(max(e[0], e[1]), max(e[2], e[3])) >= dfa_hit) hits_of(c0 + 4 * d, e);
                }
            }

            /* ---------------- phase H: lane = position, two rows at a time ----------------
               step d: row pairs rp = 4 d + k at 256 rp + 64 half + 4 (mm ^ (rp & 15)) */
            if (p < U) {
                const uint32_t hbp = lds_addr(l_hist_base + p), hpp = lds_addr(l_hist_phred + p);
                const uint32_t mm4 = 4 * mm;
                uint32_t hv = h_row;
#pragma unroll 1
                for (uint32_t d = 0; d < 8; d++) {
                    const uint32_t sx = (16 * d) & 60;
                    uint32_t sw[4], qw[4];
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++) {
                        const uint32_t ha = xor_add(mm4, sx | (4 * k), hv);
                        sw[k] = lds_u32(ha + 256 * k);
                        qw[k] = lds_u32(ha + 256 * k + 4 * RING_QUAL_WORDS);
                    }
                    hv += 4 * 256;
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++) {
                        const uint32_t cls = __builtin_amdgcn_ubfe(sw[k], h_sh + 1, 3);
                        const uint32_t bin = min(__builtin_amdgcn_ubfe(qw[k], h_sh, 8) - 33u, 47u) >> 2;
                        lds_inc(hbp + __umul24(cls, hs * 4));
                        lds_inc(hpp + __umul24(bin, hs * 4));
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }

        /* ---------------- per-read epilogue: lane = read ---------------- */
        double total = acc0 + acc1 + acc2 + acc3; /* :2098-2099 */
        for (uint32_t pos = Lmain; pos < U; pos++) { /* :2100-2112 */
            const uint32_t qb = P.buf[qoff + pos];
            total += l_err[qb < 128 ? qb : 0];
        }
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

    /* ---- merge the workgroup's histograms (end-anchored = a window of the positional) ---- */
    __syncthreads();
    if (AD)
        for (uint32_t i = tid; i < P.ad_lds * hs; i += RING_THREADS) {
            const uint32_t v = l_adf[i];
            if (!v) continue;
            const uint32_t a = i / hs, start = i % hs;
            atomicAdd(&P.ad_fwd[a * P.ad_cap + start], (unsigned long long)v);
            atomicAdd(&P.ad_rev[a * P.ad_cap + (U - 1 - start)], (unsigned long long)v);
        }
    const uint32_t ean = min(P.ea_len, U);
    for (uint32_t i = tid; i < hs * BASE_COLS; i += RING_THREADS) {
        const uint32_t v = l_hist_base[i], c = i / hs, pos = i % hs;
        if (!v) continue;
        atomicAdd(&P.qc_base[(uint64_t)pos * 5 + c], (unsigned long long)v);
        if (pos >= U - ean) atomicAdd(&P.qc_ea_base[(uint64_t)(P.ea_len - U + pos) * 5 + c], (unsigned long long)v);
    }
    for (uint32_t i = tid; i < hs * PHRED_COLS; i += RING_THREADS) {
        const uint32_t v = l_hist_phred[i], c = i / hs, pos = i % hs;
        if (!v) continue;
        atomicAdd(&P.qc_phred[(uint64_t)pos * 12 + c], (unsigned long long)v);
        if (pos >= U - ean) atomicAdd(&P.qc_ea_phred[(uint64_t)(P.ea_len - U + pos) * 12 + c], (unsigned long long)v);
    }
    for (uint32_t i = tid; i < 101; i += RING_THREADS)
        if (l_gc[i]) atomicAdd(&P.qc_gc[i], (unsigned long long)l_gc[i]);
    for (uint32_t i = tid; i < 94; i += RING_THREADS)
        if (l_ps[i]) atomicAdd(&P.qc_ps[i], (unsigned long long)l_ps[i]);
}


/* ---- k_wide: batches of one read length, 64 bytes per row and visit ----
 * What bounds the fused pass on short reads is how the memory system takes the gather: a wave
 * that visits its 64 rows 32 bytes at a time (k_pass, k_ring) moves 8.7 GB of records in
 * 3.7 ms whatever it computes; 64 bytes per row and visit, four lanes side by side on one row,
 * take 2.3 ms (a linear stream: 1.4 ms).  So this kernel walks the reads in
 * chunks of 64 positions: tile rows are 64 bytes (the geometry of k_ring's rings, without the
 * rotation: loads start at positions, not at sectors), a chunk is staged by four lanes per
 * row, phase S takes 16 dwords per row and chunk and phase H two halves of 32 positions.
 * 8 KB of tiles per wave: one workgroup of 16 waves per CU shares one set of LDS histograms
 * (150 KB in all).  Row offsets live in registers (no LDS left for them). */
constexpr int WIDE_THREADS = 1024, WIDE_WAVES = WIDE_THREADS / 64;
constexpr uint32_t WIDE_CW = 64;

size_t wide_lds_bytes(uint32_t uniform_len, bool ad, uint32_t dfa_states, uint32_t ad_lds)
{
    size_t b = FIXED_BYTES + (size_t)WIDE_WAVES * 2 * RING_TILE_WORDS * 4;
    b += (size_t)hist_stride(uniform_len) * (BASE_COLS + PHRED_COLS) * 4;
    if (ad) b += (size_t)dfa_states * 16 + (size_t)ad_lds * hist_stride(uniform_len) * 4;
    return b + 16;
}

template <bool AD>
__global__ void __launch_bounds__(WIDE_THREADS) k_wide(PassParams P)
{
    extern __shared__ __align__(16) uint8_t smem[];
    double *l_err = (double *)smem;                        /* [136] by raw quality byte, 128 = padding */
    double *l_thr = l_err + 136;                           /* [96] */
    uint32_t *l_gc = (uint32_t *)(l_thr + 96);             /* [104] */
    uint32_t *l_ps = l_gc + 104;                           /* [96] */
    uint16_t *l_dfa = (uint16_t *)(l_ps + 96);             /* rows addressed with 16 bits */
    uint32_t *l_tiles = (uint32_t *)(l_dfa + (AD ? P.dfa_states * 8 : 0));
    const uint32_t U = P.uniform_len, hs = hist_stride(U);
    uint32_t *l_hist_base = l_tiles + WIDE_WAVES * 2 * RING_TILE_WORDS; /* [5][hs] */
    uint32_t *l_hist_phred = l_hist_base + hs * BASE_COLS;               /* [12][hs] */
    uint32_t *l_adf = l_hist_phred + hs * PHRED_COLS;                    /* [ad_lds][hs] */

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (lds_addr(l_err) != 0) __builtin_trap(); /* see k_pass: byte << 3 is the address of its error rate */
    const uint32_t dfa_root = AD ? lds_addr(l_dfa) : 0, dfa_hit = dfa_root + P.dfa_accept * 16;
    for (int i = tid; i < 136; i += WIDE_THREADS) {
        double e;
        if (i >= 33 && i <= 33 + SQ_PHRED_MAX) e = __longlong_as_double((long long)c_error_rate_bits[i - 33]);
        else if (i >= 128) e = 0.0;
        else e = __longlong_as_double(0x7FF8000000000000LL);
        l_err[i] = e;
    }
    for (int i = tid; i < 96; i += WIDE_THREADS) l_thr[i] = i < 94 ? P.thresholds[i] : 0.0;
    for (int i = tid; i < 104; i += WIDE_THREADS) l_gc[i] = 0;
    for (int i = tid; i < 96; i += WIDE_THREADS) l_ps[i] = 0;
    for (uint32_t i = tid; i < hs * (BASE_COLS + PHRED_COLS); i += WIDE_THREADS) l_hist_base[i] = 0;
    if (AD) {
        for (uint32_t i = tid; i < P.dfa_states * 8; i += WIDE_THREADS)
            l_dfa[i] = (uint16_t)((P.dfa[i] & 0xFFF0u) + dfa_root);
        for (uint32_t i = tid; i < P.ad_lds * hs; i += WIDE_THREADS) l_adf[i] = 0;
    }
    __syncthreads();

    uint32_t *w_seq = l_tiles + wave * 2 * RING_TILE_WORDS, *w_qual = w_seq + RING_QUAL_WORDS;
    const uint32_t half = (uint32_t)lane >> 5, pl = (uint32_t)lane & 31;
    const uint32_t h_sh = 8 * (pl & 3), h_dw = pl >> 2;
    /* v_perm selector of bins_of_two: this lane's byte of the first word into the low half, of
       the second word into the high half, zeros (0x0c) above them */
    const uint32_t q_sel = 0x0c000c00u | (pl & 3) | ((4 + (pl & 3)) << 16);
    /* byte addresses of the fused loop: what is fixed per lane (ring_idx split up) */
    const uint32_t s_row = lds_addr(w_seq) + ((uint32_t)lane >> 1) * 256 + ((uint32_t)lane & 1) * 64;
    const uint32_t s_sw4 = 4 * (((uint32_t)lane >> 1) & 15);
    const uint32_t h_row = lds_addr(w_seq) + 64 * half;
    const uint32_t Lmain = 4 * ((U - 1) / 4); /* _qcmodule.c:2062,2068 */
    /* loader: rows it * 16 + (lane >> 2), bytes [16 piece, 16 piece + 16) of the chunk */
    const uint32_t l_row = (uint32_t)lane >> 2, piece16 = ((uint32_t)lane & 3) * 16;

    const uint64_t ngroups = P.n / 64;
    /* the wave's number as a scalar (hipcc does not know that tid >> 6 is uniform): the group counter g lives in scalar registers instead of
       in a vector pair that was spilled across the chunk loop */
    const uint64_t n_waves = (uint64_t)gridDim.x * WIDE_WAVES, wave_id = (uint64_t)blockIdx.x * WIDE_WAVES + (uint32_t)__builtin_amdgcn_readfirstlane(wave);
    const uint32_t c_last = WIDE_CW * ((U - 1) / WIDE_CW);   /* the chunk that holds the end of the reads */

    /* One pipeline over all (group, chunk) steps of the wave: the loads of the step after the
       current one are always in flight, across group boundaries too (the first chunk of the next
       group is fetched while the last chunk of this one is counted; its record offsets were
       fetched a group earlier). */
    uint64_t g = wave_id;
    const uint8_t *sp[4], *qp[4]; /* where this lane's 16 bytes of the four rows it loads start */
    uint4 pf_s[4], pf_q[4];
#pragma unroll
    for (int it = 0; it < 4; it++) pf_s[it] = pf_q[it] = make_uint4(0, 0, 0, 0);
    auto rows_of = [&](unsigned long long soff, unsigned long long qoff) {
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int row = it * 16 + (int)l_row;
            sp[it] = P.buf + (unsigned long long)__shfl(soff, row) + piece16;
            qp[it] = P.buf + (unsigned long long)__shfl(qoff, row) + piece16;
        }
    };
    auto fetch = [&](uint32_t c0) {
        /* a piece behind the end of the reads is not loaded: STAGE turns whatever the registers
           still hold into padding.  The buffer is the library's own: 64 readable bytes behind
           its end */
        if (c0 + piece16 < U) {
#pragma unroll
            for (int it = 0; it < 4; it++) {
                pf_s[it] = *(const uint4 *)(sp[it] + c0);
                pf_q[it] = *(const uint4 *)(qp[it] + c0);
            }
        }
    };
    unsigned long long soff_n = 0, qoff_n = 0; /* sequence / qualities of this lane's read in the next group */
    if (g < ngroups) {
        const sq_meta m = P.metas[g * 64 + lane];
        rows_of(m.record_start + m.sequence_offset, m.record_start + m.qualities_offset);
        fetch(0);
    }
    while (g < ngroups) {
        const bool has_next = g + n_waves < ngroups;
        if (has_next) {
            uint32_t lane_n = (uint32_t)lane;   /* opaque: the lane's part of the address is made here (kept across the loop it was spilled) */
            asm volatile("" : "+v"(lane_n));
            const sq_meta m = P.metas[(g + n_waves) * 64 + lane_n];
            soff_n = m.record_start + m.sequence_offset;
            qoff_n = m.record_start + m.qualities_offset;
        }

        double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
        uint32_t st = dfa_root, gc_cnt = 0, acgt_cnt = 0;
        unsigned long long found = 0;

        for (uint32_t c0 = 0; c0 < U; c0 += WIDE_CW) {
            /* ---------------- STAGE: 4 lanes x 16 bytes per row ---------------- */
            {
                const uint32_t p0 = c0 + piece16;
                const int nv = p0 < U ? (int)min(16u, U - p0) : 0;
                const uint32_t m0 = piece16 >> 2;
#pragma unroll
                for (int it = 0; it < 4; it++) {
                    const uint32_t row = (uint32_t)it * 16 + l_row;
                    uint4 sv, qv = pf_q[it];
                    sv.x = cls2_of_dword(pf_s[it].x); sv.y = cls2_of_dword(pf_s[it].y);
                    sv.z = cls2_of_dword(pf_s[it].z); sv.w = cls2_of_dword(pf_s[it].w);
                    if (nv < 16) { /* the chunk at the end of the reads */
                        sv.x = pad_tail(sv.x, nv, CLS2_PAD4); sv.y = pad_tail(sv.y, nv - 4, CLS2_PAD4);
                        sv.z = pad_tail(sv.z, nv - 8, CLS2_PAD4); sv.w = pad_tail(sv.w, nv - 12, CLS2_PAD4);
                        qv.x = pad_tail(qv.x, nv, PAD4); qv.y = pad_tail(qv.y, nv - 4, PAD4);
                        qv.z = pad_tail(qv.z, nv - 8, PAD4); qv.w = pad_tail(qv.w, nv - 12, PAD4);
                    }
                    uint32_t *ts = w_seq + ring_row(row), *tq = w_qual + ring_row(row);
                    const uint32_t swz = (row >> 1) & 15u;
                    ts[(m0 + 0) ^ swz] = sv.x; ts[(m0 + 1) ^ swz] = sv.y;
                    ts[(m0 + 2) ^ swz] = sv.z; ts[(m0 + 3) ^ swz] = sv.w;
                    tq[(m0 + 0) ^ swz] = qv.x; tq[(m0 + 1) ^ swz] = qv.y;
                    tq[(m0 + 2) ^ swz] = qv.z; tq[(m0 + 3) ^ swz] = qv.w;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (c0 + WIDE_CW < U) {
                fetch(c0 + WIDE_CW);
            } else if (has_next) {
                rows_of(soff_n, qoff_n);
                fetch(0);
            }
            /* update_adapter_count_array, _qcmodule.c:2643-2672 */
            auto hits_of = [&](uint32_t pos0, const uint32_t e[4]) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (e[j] < dfa_hit) continue;
                    unsigned long long hits = P.dfa_out[(e[j] - dfa_root) >> 4] & ~found;
                    found |= hits;
                    const uint32_t pos = pos0 + j;
                    while (hits) {
                        const int a = __ffsll((long long)hits) - 1;
                        hits &= hits - 1;
                        const uint32_t start = pos - P.ad_len[a] + 1;
                        if (P.ad_lds) {
                            atomicAdd(&l_adf[a * hs + start], 1u);
                        } else {
                            atomicAdd(&P.ad_fwd[a * P.ad_cap + start], 1ULL);
                            atomicAdd(&P.ad_rev[a * P.ad_cap + (U - 1 - start)], 1ULL);
                        }
                    }
                }
            };
            /* Phases S (lane = read: the chains) and H (lane = position: the histograms) share
               one instruction stream, H fills the waits of S.  A chunk is two halves of 32
               positions; step d of half hc: dword 8 hc + d of every row (S), row pairs 4 d ..
               4 d + 3 at the positions of that half (H).  Nothing is asked per step: what lies
               behind the end of the reads is padding that counts for nothing (class 7 walks the
               automaton back to its root, its quality adds +0.0, and in the histograms it falls
               into columns >= U, which the merge skips); only the four chains need a word: they
               stop four short of the end (:2068), dwords from d_plain on are taken as padding */
            const uint32_t d_plain = Lmain > c0 ? min(2 * ROW_WORDS, (Lmain - c0) / 4) : 0;
#pragma unroll 1
            for (uint32_t hc = 0; hc < 2; hc++) {
                if (c0 + 32 * hc >= U) break;
                const uint32_t p = c0 + 32 * hc + pl; /* < hist_stride(U) */
                uint32_t hv = h_row;
                const uint32_t hbp = lds_addr(l_hist_base + p), hpp = lds_addr(l_hist_phred + p);
                const uint32_t mm4 = 4 * (h_dw + 8 * hc);
#pragma unroll 1
                for (uint32_t d = 0; d < ROW_WORDS; d++) {
                    const uint32_t ds = 8 * hc + d;
                    const uint32_t sa = xor_add(s_sw4, 4 * ds, s_row);
                    const uint32_t sd = lds_u32(sa);
                    uint32_t qd = lds_u32(sa + 4 * RING_QUAL_WORDS);
                    qd = ds < d_plain ? qd : PAD4;
                    const double e0 = lds_f64(shl3_byte<0>(qd)), e1 = lds_f64(shl3_byte<1>(qd));
                    const double e2 = lds_f64(shl3_byte<2>(qd)), e3 = lds_f64(shl3_byte<3>(qd));
                    acc0 += e0;
                    acc1 += e1;
                    acc2 += e2;
                    acc3 += e3;
                    gc_cnt += __popc(((sd >> 1) ^ sd) & 0x02020202u);
                    acgt_cnt += __popc(~sd & 0x08080808u);
                    uint32_t e[4];
                    bool hit = false;
                    if (AD) {
                        e[0] = lds_u16(or_byte<0>(st, sd));
                        e[1] = lds_u16(or_byte<1>(e[0], sd));
                        e[2] = lds_u16(or_byte<2>(e[1], sd));
                        e[3] = lds_u16(or_byte<3>(e[2], sd));
                        st = e[3];
                        hit = max(max(e[0], e[1]), max(e[2], e[3])) >= dfa_hit;
                    }
                    /* row pair rp = 4 d + k: 256 rp + 64 half + 4 (mm ^ (rp & 15)) */
                    const uint32_t sx = (16 * d) & 60;
                    uint32_t sw[4], qw[4];
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++) {
                        const uint32_t ha = xor_add(mm4, sx | (4 * k), hv);
                        sw[k] = lds_u32(ha + 256 * k);
                        qw[k] = lds_u32(ha + 256 * k + 4 * RING_QUAL_WORDS);
                    }
#pragma unroll
                    for (uint32_t k = 0; k < 4; k += 2) {
                        const uint32_t bins = bins_of_two(qw[k], qw[k + 1], q_sel);
                        lds_inc(hbp + __umul24(__builtin_amdgcn_ubfe(sw[k], h_sh + 1, 3), hs * 4));
                        lds_inc(mad_half<0>(bins, hs * 4, hpp));
                        lds_inc(hbp + __umul24(__builtin_amdgcn_ubfe(sw[k + 1], h_sh + 1, 3), hs * 4));
                        lds_inc(mad_half<1>(bins, hs * 4, hpp));
                    }
                    hv += 4 * 256;
                    if (AD && hit) hits_of(c0 + 4 * ds, e);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }

        /* ---------------- per-read epilogue: lane = read ---------------- */
        double total = acc0 + acc1 + acc2 + acc3; /* :2098-2099 */
        {   /* :2100-2112: the 1-4 qualities behind the chains lie in one dword of the tile the last
               chunk left; what is past the end of the read is padding and adds +0.0 */
            const uint32_t qt = w_qual[ring_idx((uint32_t)lane, (Lmain - c_last) >> 2)];
            total += l_err[qt & 0xFF];
            total += l_err[(qt >> 8) & 0xFF];
            total += l_err[(qt >> 16) & 0xFF];
            total += l_err[qt >> 24];
        }
        uint32_t lane_e = (uint32_t)lane;   /* the record's index made again (kept from the top of the group it was spilled across the chunk loop) */
        asm volatile("" : "+v"(lane_e));
        const uint64_t r_e = g * 64 + lane_e;
        P.metas[r_e].accumulated_error_rate = total; /* :2126 */
        if (total != total) atomicMin(P.qc_first_bad, (unsigned long long)(P.first_read_index + r_e));
        if (acgt_cnt > 0) atomicAdd(&l_gc[(uint32_t)round((double)gc_cnt * 100.0 / (double)acgt_cnt)], 1u);
        uint32_t Ud = U;   /* opaque: (double)U kept across the loop was spilled to scratch and reloaded here behind an s_waitcnt vmcnt(0) */
        asm volatile("" : "+s"(Ud));
        const double avg = total / (double)Ud;
        uint32_t lo = 0, hi = 93;
        while (lo < hi) {
            const uint32_t mid = (lo + hi + 1) >> 1;
            if (avg <= l_thr[mid]) lo = mid; else hi = mid - 1;
        }
        atomicAdd(&l_ps[lo], 1u);
        /* the next STAGE overwrites the tile this epilogue read */
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        g += n_waves;
    }

    /* ---- merge the workgroup's histograms (end-anchored = a window of the positional) ---- */
    __syncthreads();
    uint32_t tid_e = threadIdx.x;   /* made again here: what the merge derives from the thread's index was kept from the top of the kernel and spilled across the whole loop */
    asm volatile("" : "+v"(tid_e));
    if (AD)
        for (uint32_t i = tid_e; i < P.ad_lds * hs; i += WIDE_THREADS) {
            const uint32_t v = l_adf[i];
            if (!v) continue;
            const uint32_t a = i / hs, start = i % hs;
            atomicAdd(&P.ad_fwd[a * P.ad_cap + start], (unsigned long long)v);
            atomicAdd(&P.ad_rev[a * P.ad_cap + (U - 1 - start)], (unsigned long long)v);
        }
    const uint32_t ean = min(P.ea_len, U);
    for (uint32_t i = tid_e; i < hs * BASE_COLS; i += WIDE_THREADS) {
        const uint32_t v = l_hist_base[i], c = i / hs, pos = i % hs;
        if (!v || pos >= U) continue; /* columns behind U collected the padding */
        atomicAdd(&P.qc_base[(uint64_t)pos * 5 + c], (unsigned long long)v);
        if (pos >= U - ean) atomicAdd(&P.qc_ea_base[(uint64_t)(P.ea_len - U + pos) * 5 + c], (unsigned long long)v);
    }
    for (uint32_t i = tid_e; i < hs * PHRED_COLS; i += WIDE_THREADS) {
        const uint32_t v = l_hist_phred[i], c = i / hs, pos = i % hs;
        if (!v || pos >= U) continue;
        atomicAdd(&P.qc_phred[(uint64_t)pos * 12 + c], (unsigned long long)v);
        if (pos >= U - ean) atomicAdd(&P.qc_ea_phred[(uint64_t)(P.ea_len - U + pos) * 12 + c], (unsigned long long)v);
    }
    for (uint32_t i = tid_e; i < 101; i += WIDE_THREADS)
        if (l_gc[i]) atomicAdd(&P.qc_gc[i], (unsigned long long)l_gc[i]);
    for (uint32_t i = tid_e; i < 94; i += WIDE_THREADS)
        if (l_ps[i]) atomicAdd(&P.qc_ps[i], (unsigned long long)l_ps[i]);
}

/* ---- k_ptq: PerTileQuality alone on a batch of one read length ----
 * The pass needs the qualities only: per tile and position the sum of the error rates
 * (`:3196-3212`) and per tile the count of reads per length.  Like k_wide it visits 64 bytes per
 * row at a time (four lanes side by side on a row); the tile needs no swizzle (a row is staged
 * with one ds_write_b128 per lane, and the 64 lanes of the counting phase read 16 consecutive
 * dwords of one row, four lanes per dword).  Lane p owns position c0 + p of the chunk and sums
 * the error rates of the 64 rows in a register; a wave keeps the sums of the tile its groups are
 * in (and the number of reads it has seen of it) in LDS and hands them over when the tile
 * changes.  A group whose reads are not all of one tile (the seam between two tiles) adds row by
 * row to the device tables.  Full groups only; the walk follows P.order when the host sorted by
 * tile, with a contiguous run of groups per wave either way. */
constexpr int PTQ_THREADS = 512, PTQ_WAVES = PTQ_THREADS / 64, PTQ_DEPTH = 3;

size_t ptq_lds_bytes(uint32_t uniform_len)
{
    return 136 * 8 + (size_t)PTQ_WAVES * (64 * 64 + hist_stride(uniform_len) * 8) + 16;
}

__global__ void __launch_bounds__(PTQ_THREADS, 2) k_ptq(PassParams P)
{
    extern __shared__ __align__(16) uint8_t smem[];
    double *l_err = (double *)smem;                                  /* [136] by raw quality byte, 128 = padding */
    const uint32_t U = P.uniform_len, pts = hist_stride(U);
    double *l_pt = l_err + 136;                                      /* [waves][pts] */
    uint32_t *l_tiles = (uint32_t *)(l_pt + PTQ_WAVES * pts);        /* [waves][64 rows][16 dwords] */
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < 136; i += PTQ_THREADS) {
        double e;
        if (i >= 33 && i <= 33 + SQ_PHRED_MAX) e = __longlong_as_double((long long)c_error_rate_bits[i - 33]);
        else if (i >= 128) e = 0.0;
        else e = __longlong_as_double(0x7FF8000000000000LL);
        l_err[i] = e;
    }
    for (uint32_t i = tid; i < PTQ_WAVES * pts; i += PTQ_THREADS) l_pt[i] = 0.0;
    __syncthreads();
    double *w_pt = l_pt + wave * pts;
    uint32_t *w_tile = l_tiles + wave * 64 * 16;
    const uint32_t l_row = (uint32_t)lane >> 2, piece16 = ((uint32_t)lane & 3) * 16;

    int32_t acc_slot = -1;
    uint32_t acc_reads = 0;
    auto flush = [&]() {
        if (acc_slot < 0) return;
        if (lane == 0 && acc_reads)
            atomicAdd(&P.pt_len_counts[(uint64_t)acc_slot * P.pt_cap + (U - 1)], (unsigned long long)acc_reads);
        acc_reads = 0;
        for (uint32_t i = lane; i < pts; i += 64) {
            const double v = w_pt[i];
            if (v != 0.0) {
                unsafeAtomicAdd(&P.pt_errors[(uint64_t)acc_slot * P.pt_cap + i], v);
                w_pt[i] = 0.0;
            }
        }
    };

    const uint64_t ngroups = P.n / 64;
    const uint64_t n_waves = (uint64_t)gridDim.x * PTQ_WAVES, wave_id = (uint64_t)blockIdx.x * PTQ_WAVES + wave;
    /* tile-sorted walk: a contiguous run of groups per wave (few tiles per wave).  Stored order
       (the batch came ordered by tile): neighbouring waves take neighbouring groups, all waves
       move through the batch together (one tile at a time, and no two waves a large power-of-two
       stride apart in memory) */
    const uint64_t per_wave = (ngroups + n_waves - 1) / n_waves;
    const uint64_t g_begin = P.order ? wave_id * per_wave : wave_id;
    const uint64_t g_end = P.order ? min(ngroups, (wave_id + 1) * per_wave) : ngroups;
    const uint64_t g_step = P.order ? 1 : n_waves;
    for (uint64_t g = g_begin; g < g_end; g += g_step) {
        const uint64_t slot_index = g * 64 + lane;
        const uint64_t r = P.order ? P.order[slot_index] : slot_index;
        const sq_meta m = P.metas[r];
        const unsigned long long qoff = m.record_start + m.qualities_offset;
        int32_t slot = P.pt_slot[r];
        if (slot < 0 || P.first_read_index + r >= P.pt_first_bad) slot = -1;
        const int32_t g_slot = __builtin_amdgcn_readfirstlane(slot);
        const bool one_tile = g_slot >= 0 && __all(slot == g_slot);
        if (one_tile && g_slot != acc_slot) {
            flush();
            acc_slot = g_slot;
        }
        const uint8_t *qp[4];
#pragma unroll
        for (int it = 0; it < 4; it++)
            qp[it] = P.buf + (unsigned long long)__shfl(qoff, it * 16 + (int)l_row) + piece16;
        /* registers are plentiful here (one stream, no chains): three chunks of loads in flight */
        uint4 pf[PTQ_DEPTH][4];
#pragma unroll
        for (int k = 0; k < PTQ_DEPTH; k++)
#pragma unroll
            for (int it = 0; it < 4; it++) pf[k][it] = make_uint4(PAD4, PAD4, PAD4, PAD4);
        auto fetch = [&](uint32_t c0, uint4 (&f)[4]) { /* the buffer is the library's own: 64 readable bytes behind its end */
            if (c0 + piece16 < U) {
#pragma unroll
                for (int it = 0; it < 4; it++) f[it] = *(const uint4 *)(qp[it] + c0);
            }
        };
#pragma unroll
        for (int k = 0; k < PTQ_DEPTH; k++) fetch(k * WIDE_CW, pf[k]);
        for (uint32_t c0 = 0; c0 < U; c0 += WIDE_CW) {
            {
                const uint32_t p0 = c0 + piece16;
                const int nv = p0 < U ? (int)min(16u, U - p0) : 0;
#pragma unroll
                for (int it = 0; it < 4; it++) {
                    uint4 qv = pf[0][it];
                    if (nv < 16) {
                        qv.x = pad_tail(qv.x, nv, PAD4); qv.y = pad_tail(qv.y, nv - 4, PAD4);
                        qv.z = pad_tail(qv.z, nv - 8, PAD4); qv.w = pad_tail(qv.w, nv - 12, PAD4);
                    }
                    *(uint4 *)(w_tile + ((uint32_t)it * 16 + l_row) * 16 + (piece16 >> 2)) = qv;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k + 1 < PTQ_DEPTH; k++)
#pragma unroll
                for (int it = 0; it < 4; it++) pf[k][it] = pf[k + 1][it];
            fetch(c0 + PTQ_DEPTH * WIDE_CW, pf[PTQ_DEPTH - 1]);
            const uint32_t p = c0 + (uint32_t)lane;
            if (p < U) {
                const uint32_t dw = (uint32_t)lane >> 2, sh = 8 * ((uint32_t)lane & 3);
                if (one_tile) {
                    double run = 0.0;
#pragma unroll 8
                    for (uint32_t row = 0; row < 64; row++) {
                        const uint32_t qb = (w_tile[row * 16 + dw] >> sh) & 0xFF;
                        run += l_err[qb < 136 ? qb : 0];
                    }
                    w_pt[p] += run; /* this lane is the only one of the wave at position p */
                } else {
                    for (uint32_t row = 0; row < 64; row++) {
                        const int32_t s_row = __builtin_amdgcn_readlane(slot, (int)row);
                        if (s_row < 0) continue;
                        const uint32_t qb = (w_tile[row * 16 + dw] >> sh) & 0xFF;
                        unsafeAtomicAdd(&P.pt_errors[(uint64_t)s_row * P.pt_cap + p], l_err[qb < 136 ? qb : 0]);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (one_tile) acc_reads += 64;
        else if (slot >= 0) atomicAdd(&P.pt_len_counts[(uint64_t)slot * P.pt_cap + (U - 1)], 1ULL);
    }
    flush();
}

size_t ring_lds_bytes(uint32_t uniform_len, bool ad, uint32_t dfa_states, uint32_t ad_lds = 0)
{
    size_t b = FIXED_BYTES + (size_t)RING_WAVES * 2 * RING_TILE_WORDS * 4;
    b += (size_t)hist_stride(uniform_len) * (BASE_COLS + PHRED_COLS) * 4;
    if (ad) b += (size_t)dfa_states * 16 + (size_t)ad_lds * hist_stride(uniform_len) * 4;
    return b + 16;
}

size_t pass_lds_bytes(bool qc, uint32_t lds_len, uint32_t ea_rows, bool dfa_lds, uint32_t dfa_states,
                      uint32_t ad_lds = 0, uint32_t pt_len = 0)
{
    size_t b = FIXED_BYTES + (size_t)WAVES * WAVE_WORDS * 4;
    if (qc) b += (size_t)(hist_stride(lds_len) + hist_stride(ea_rows)) * (BASE_COLS + PHRED_COLS) * 4;
    if (dfa_lds) b += (size_t)dfa_states * 16 + (size_t)ad_lds * hist_stride(lds_len) * 4;
    if (pt_len) b += (size_t)WAVES * hist_stride(pt_len) * 8 + 8;
    return b + 16;
}

/* ---- PerTileQuality prepass: tile id and table slot of every record -------- */
struct TileMap {
    long long *keys; /* [TILE_MAP_SIZE], TILE_EMPTY when free */
    int *vals;       /* slot numbers, -1 until published */
    int *n_slots;
};

/* tile_id_of, tile_id_of_words: sq_pass.h (k_span<PT> parses headers too) */
/* pass 1: tile id of every record, first record whose header does not parse */
__global__ void k_tile_parse(const uint8_t *buf, uint64_t buf_len, const sq_meta *metas, uint64_t n,
                             uint64_t first_read_index, long long *tiles,
                             unsigned long long *first_bad)
{
    for (uint64_t r = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; r < n;
         r += (uint64_t)gridDim.x * blockDim.x) {
        const sq_meta m = metas[r];
        long long tile = -2;
        if (m.name_length <= 48 && m.record_start + 48 <= buf_len) {
            uint64_t w[6];
            const uint8_t *name = buf + m.record_start;
            uint4 a, b2, c;
            __builtin_memcpy(&a, name, 16);
            __builtin_memcpy(&b2, name + 16, 16);
            __builtin_memcpy(&c, name + 32, 16);
            w[0] = a.x | (uint64_t)a.y << 32;  w[1] = a.z | (uint64_t)a.w << 32;
            w[2] = b2.x | (uint64_t)b2.y << 32; w[3] = b2.z | (uint64_t)b2.w << 32;
            w[4] = c.x | (uint64_t)c.y << 32;  w[5] = c.z | (uint64_t)c.w << 32;
            tile = tile_id_of_words(w, m.name_length);
        }
        if (tile == -2) tile = tile_id_of(buf + m.record_start, m.name_length);
        if (tile < 0) atomicMin(first_bad, (unsigned long long)(first_read_index + r));
        tiles[r] = tile;
    }
}

/* pass 2: table slots, only for the records in front of the first bad header
 * (the reference stops creating tiles there, _qcmodule.c:3126,3137-3148) */
__global__ void k_tile_assign(const long long *tiles, uint64_t n, uint64_t first_read_index,
                              TileMap map, int32_t *slots, const unsigned long long *first_bad,
                              int *overflow)
{
    /* a flow cell has a few hundred tiles: a workgroup remembers the ones it has resolved in LDS
       (write-once entries) instead of asking the L2 for the same hot lines 10 M times */
    constexpr uint32_t CACHE = 1024;
    __shared__ long long c_key[CACHE];
    __shared__ int c_val[CACHE];
    for (uint32_t i = threadIdx.x; i < CACHE; i += blockDim.x) { c_key[i] = TILE_EMPTY; c_val[i] = -1; }
    __syncthreads();
    const unsigned long long stop = *first_bad;
    uint32_t changes = 0; /* neighbours in stored order with different tiles: overflow[1] */
    /* a workgroup takes a contiguous stretch of the batch: reads ordered by tile then ask it for
       one or two tiles, which it resolves once */
    const uint64_t per_wg = ((n + gridDim.x - 1) / gridDim.x + blockDim.x - 1) / blockDim.x * blockDim.x;
    const uint64_t r_end = min(n, (blockIdx.x + 1) * per_wg);
    for (uint64_t r = blockIdx.x * per_wg + threadIdx.x; r < r_end; r += blockDim.x) {
        const long long tile = tiles[r];
        if (r > 0 && tiles[r - 1] != tile) changes++;
        int slot = -1;
        /* reads ordered by tile: the 64 records of a wave ask for one tile, and with a stride of
           gridDim x 256 records between a workgroup's rounds it is a tile the workgroup has not
           resolved yet more often than not: one lane asks for all */
        const bool mine = tile >= 0 && first_read_index + r < stop;
        const long long tile0 = __shfl(tile, __ffsll((long long)__ballot(1)) - 1);
        const bool shared = __all(mine && tile == tile0);
        const int leader = __ffsll((long long)__ballot(1)) - 1;
        if (mine && (!shared || (int)(threadIdx.x & 63) == leader)) {
            const uint32_t ci = (uint32_t)(((unsigned long long)tile * 0x9E3779B97F4A7C15ULL) >> 40) & (CACHE - 1);
            bool cached = false;
            if (__hip_atomic_load(&c_key[ci], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == tile) {
                const int v = __hip_atomic_load(&c_val[ci], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (v >= 0) {
                    slot = v;
                    cached = true;
                }
            }
            if (!cached) {
            uint32_t idx = (uint32_t)(((unsigned long long)tile * 0x9E3779B97F4A7C15ULL) >> 48) & (TILE_MAP_SIZE - 1);
            bool done = false;
            for (uint32_t probes = 0; !done && probes < 4 * TILE_MAP_SIZE; probes++) {
                long long k = __hip_atomic_load(&map.keys[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (k == TILE_EMPTY) {
                    const long long old = (long long)atomicCAS((unsigned long long *)&map.keys[idx],
                                                               (unsigned long long)TILE_EMPTY,
                                                               (unsigned long long)tile);
                    if (old == TILE_EMPTY) {
                        slot = atomicAdd(map.n_slots, 1);
                        __hip_atomic_store(&map.vals[idx], slot, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                        done = true;
                    } else {
                        k = old;
                    }
                }
                if (!done) {
                    if (k == tile) {
                        const int v = __hip_atomic_load(&map.vals[idx], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                        if (v >= 0) { slot = v; done = true; }
                        /* else: the inserting lane publishes it this iteration */
                    } else {
                        idx = (idx + 1) & (TILE_MAP_SIZE - 1);
                    }
                }
            }
            if (!done) *overflow = 1;
            else if (atomicCAS((unsigned long long *)&c_key[ci], (unsigned long long)TILE_EMPTY,
                               (unsigned long long)tile) == (unsigned long long)TILE_EMPTY)
                __hip_atomic_store(&c_val[ci], slot, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        if (shared) slot = __shfl(slot, leader);
        slots[r] = slot;
    }
    for (int off = 32; off > 0; off >>= 1) changes += __shfl_xor(changes, off);
    if ((threadIdx.x & 63) == 0 && changes) atomicAdd((unsigned int *)&overflow[1], changes);
}

/* ---- phred_scores thresholds ----------------------------------------------------
 * bin = floor(-10 * log10(avg))  (_qcmodule.c:2127-2136) is evaluated with the
 * HOST libm: thr[k] is the largest double with bin >= k, found by bisection
 * over the bit patterns, so the device only compares.   */
int phred_bin_host(double avg) { return (int)(uint64_t)floor(-10.0 * log10(avg)); }

const double *phred_thresholds()
{
    static double thr[94];
    static bool ready = false;
    if (ready) return thr;
    thr[0] = INFINITY;
    for (int k = 1; k < 94; k++) {
        double lo_d = 1e-12, hi_d = 1.0; /* bin(lo) >= k, bin(hi) = 0 < k */
        uint64_t lo, hi;
        memcpy(&lo, &lo_d, 8);
        memcpy(&hi, &hi_d, 8);
        while (hi - lo > 1) {
            const uint64_t mid = lo + (hi - lo) / 2;
            double md;
            memcpy(&md, &mid, 8);
            if (phred_bin_host(md) >= k) lo = mid; else hi = mid;
        }
        memcpy(&thr[k], &lo, 8);
    }
    ready = true;
    return thr;
}

/* k_span bins a read by the SUM of its error rates (no f64 division per read): row U of this table holds,
 * for every k, the largest double S with fl(S / U) <= thr[k] -- x -> fl(x / U) does not fall when x grows, so
 * total <= S says exactly what total / U <= thr[k] says (NaN: neither).  [257][96], entries 94, 95 of a row 0. */
const double *phred_sum_thresholds()
{
    static std::vector<double> t;
    if (!t.empty()) return t.data();
    const double *thr = phred_thresholds();
    t.assign(257 * 96, 0.0);
    for (int U = 1; U <= 256; U++) {
        double *row = &t[(size_t)U * 96];
        const double d = (double)U;
        row[0] = INFINITY;
        for (int k = 1; k < 94; k++) {
            double c = thr[k] * d;
            while (c / d > thr[k]) c = nextafter(c, 0.0);
            for (;;) {
                const double n = nextafter(c, INFINITY);
                if (n / d <= thr[k]) c = n; else break;
            }
            row[k] = c;
        }
    }
    return t.data();
}

SQ_EXPORT void sq_phred_sum_thresholds(uint32_t length, double *thresholds, double *sums)
{
    memcpy(thresholds, phred_thresholds(), 94 * sizeof(double));
    if (length >= 1 && length <= 256) memcpy(sums, phred_sum_thresholds() + (size_t)length * 96, 94 * sizeof(double));
}

/* ---- processing order --------------------------------------------------------
 * The tables are sums over records, so the pass may visit the records in any
 * order.  Two orders pay: by length when a batch is ragged (a wave's 64 reads
 * then end together instead of idling until the longest is done), and by tile
 * for PerTileQuality (a lane then carries its position's error sum in a
 * register while the tile stays the same). */
__global__ void k_order_keys(const sq_meta *metas, const int32_t *slots, uint32_t missing_key,
                             uint64_t n, uint32_t *keys, uint32_t *vals)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t k;
        if (slots) k = slots[i] < 0 ? missing_key : (uint32_t)slots[i];
        else k = missing_key - metas[i].sequence_length; /* longest first: the long tail starts early */
        keys[i] = k;
        vals[i] = (uint32_t)i;
    }
}

/* returns the device array of record indices sorted by key (owned by the context) */
const uint32_t *sorted_order(sq_ctx *ctx, const sq_batch *b, const int32_t *slots, uint32_t max_key)
{
    const uint64_t n = b->n;
    uint32_t *keys_in = (uint32_t *)sq_scratch(ctx, 0, n * 4), *keys_out = (uint32_t *)sq_scratch(ctx, 1, n * 4);
    uint32_t *vals_in = (uint32_t *)sq_scratch(ctx, 2, n * 4), *vals_out = (uint32_t *)sq_scratch(ctx, 3, n * 4);
    if (!keys_in || !keys_out || !vals_in || !vals_out) return nullptr;
    int blocks = (int)std::min<uint64_t>((n + 255) / 256, 8192);
    hipLaunchKernelGGL(k_order_keys, dim3(blocks), dim3(256), 0, ctx->stream, b->d_metas, slots,
                       max_key, n, keys_in, vals_in);
    int bits = 1;
    while ((1ull << bits) <= max_key) bits++;
    size_t temp_bytes = 0;
    if (hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, keys_in, keys_out, vals_in, vals_out,
                                           (int)n, 0, bits, ctx->stream) != hipSuccess)
        return nullptr;
    void *temp = sq_scratch(ctx, 4, temp_bytes ? temp_bytes : 8);
    if (!temp) return nullptr;
    if (hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out,
                                           (int)n, 0, bits, ctx->stream) != hipSuccess)
        return nullptr;
    return vals_out;
}

/* records in `order` come longest first: counts[w] = how many are longer than w * stripe */
__global__ void k_stripe_counts(const sq_meta *metas, const uint32_t *order, uint64_t n, uint32_t stripe,
                                uint32_t n_stripes, unsigned long long *counts)
{
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_stripes) return;
    const uint64_t limit = (uint64_t)w * stripe;
    uint64_t lo = 0, hi = n; /* first index whose read is not longer than limit */
    while (lo < hi) {
        const uint64_t mid = (lo + hi) / 2;
        if (metas[order[mid]].sequence_length > limit) lo = mid + 1; else hi = mid;
    }
    counts[w] = lo;
}


/* ---- an invalid phred byte (_qcmodule.c:2073-2075, 2102-2105) ---------------------------------
 * The reference raises at the read that holds it: that read's bases (both tables) and GC bin are
 * counted, its phred counts in front of the byte too; nothing behind the byte, not its
 * end-anchored phred counts, not its phred_scores bin, no accumulated_error_rate; the reads
 * behind it in the array are never looked at.  The passes above run the whole batch (an invalid
 * byte counts in phred bin 11, its read's sum is NaN and lands in phred_scores[0]) and only
 * flag the read.  When the flag is found the host runs these two kernels over that batch:
 * k_first_invalid finds its first offending read, k_qc_uncount takes back, count by count, what
 * the pass added for the reads behind it and for the part of the offender the reference never
 * reached. */
__global__ void k_first_invalid(const uint8_t *buf, const sq_meta *metas, uint64_t start, uint64_t n, unsigned long long *first)
{
    for (uint64_t r = start + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; r < n; r += (uint64_t)gridDim.x * blockDim.x) {
        const sq_meta m = metas[r];
        const uint8_t *q = buf + m.record_start + m.qualities_offset;
        bool bad = false;
        for (uint32_t p = 0; p < m.sequence_length && !bad; p++) bad = (uint8_t)(q[p] - 33) > SQ_PHRED_MAX;
        if (bad) atomicMin(first, (unsigned long long)r);
    }
}

__global__ void k_qc_uncount(PassParams P, uint64_t first)
{
    const unsigned long long minus1 = ~0ULL;
    for (uint64_t r = first + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; r < P.n; r += (uint64_t)gridDim.x * blockDim.x) {
        const sq_meta m = P.metas[r];
        const uint32_t L = m.sequence_length;
        const uint8_t *s = P.buf + m.record_start + m.sequence_offset, *q = P.buf + m.record_start + m.qualities_offset;
        const uint32_t ean = min(P.ea_len, L), ea0 = P.ea_len - ean; /* :1971-1972 */
        const bool offender = r == first;
        uint32_t from = 0; /* phred counts are taken back from this position on */
        if (offender) while (from < L && (uint8_t)(q[from] - 33) <= SQ_PHRED_MAX) from++;
        uint32_t gc = 0, acgt = 0;
        for (uint32_t p = 0; p < L; p++) {
            const uint32_t c = sq_base_class(s[p]);
            gc += c == 1 || c == 2;
            acgt += c < 4;
            if (!offender) {
                atomicAdd(&P.qc_base[(uint64_t)p * 5 + c], minus1);
                if (p >= L - ean) atomicAdd(&P.qc_ea_base[(uint64_t)(ea0 + p - (L - ean)) * 5 + c], minus1);
            }
            const uint32_t bin = min((uint32_t)q[p] - 33u, 47u) >> 2; /* what the passes count, invalid bytes in bin 11 */
            if (p >= from) atomicAdd(&P.qc_phred[(uint64_t)p * 12 + bin], minus1);
            if (p >= L - ean) atomicAdd(&P.qc_ea_phred[(uint64_t)(ea0 + p - (L - ean)) * 12 + bin], minus1);
        }
        if (!offender && acgt > 0) atomicAdd(&P.qc_gc[(uint32_t)round((double)gc * 100.0 / (double)acgt)], minus1);
        if (L > 0) { /* the bin the pass put the read's sum in (NaN: bin 0) */
            const double avg = m.accumulated_error_rate / (double)L;
            uint32_t lo = 0, hi = 93;
            while (lo < hi) {
                const uint32_t mid = (lo + hi + 1) >> 1;
                if (avg <= P.thresholds[mid]) lo = mid; else hi = mid - 1;
            }
            atomicAdd(&P.qc_ps[lo], minus1);
        }
        P.metas[r].accumulated_error_rate = 0.0; /* never written by the reference: what the parser left */
    }
}

int grid_for(const sq_ctx *ctx, uint64_t n, int wgs_per_cu)
{
    uint64_t groups = (n + 63) / 64;
    uint64_t wgs = (groups + WAVES - 1) / WAVES;
    uint64_t cap = (uint64_t)ctx->num_cus * wgs_per_cu;
    if (wgs > cap) wgs = cap;
    return (int)(wgs ? wgs : 1);
}

} // namespace

/* =============================== modules =================================== */

struct sq_qcmetrics {
    sq_ctx *ctx;
    uint64_t end_anchor, max_length = 0, number_of_reads = 0;
    size_t cap = 0;              /* rows allocated in base/phred */
    unsigned long long *d_base = nullptr, *d_phred = nullptr;
    size_t cap_phred = 0;
    unsigned long long *d_ea_base = nullptr, *d_ea_phred = nullptr, *d_gc = nullptr, *d_ps = nullptr;
    double *d_thr = nullptr, *d_thr_sum = nullptr;
    unsigned long long *d_first_bad = nullptr;
    uint64_t records_seen = 0;
    /* batches since the last flush (by id: a freed batch's address may be handed out again) and
       their longest read: max_length after an invalid phred byte took the tail of one of them back */
    struct Seen { uint64_t id; uint64_t max_length; };
    std::vector<Seen> seen;
    uint64_t max_length_flushed = 0;
    /* sq_qcmetrics_poll: the flag as it was behind the passes queued when the poll was armed */
    uint64_t *polled = nullptr;     /* page-locked */
    hipEvent_t poll_event = nullptr;
    bool poll_armed = false;
};

struct sq_adaptercounter {
    sq_ctx *ctx;
    std::vector<std::string> adapters;
    uint64_t max_length = 0, number_of_sequences = 0;
    size_t cap = 0; /* row length of fwd / rev */
    unsigned long long *d_fwd = nullptr, *d_rev = nullptr;
    /* adapters are matched in groups of <= 64 (one automaton each) */
    struct Group {
        size_t first, count;
        uint32_t states;
        uint32_t accept_first;   /* states >= this one are hits */
        uint16_t *d_dfa = nullptr;
        unsigned long long *d_out = nullptr;
        uint8_t *d_len = nullptr;
        /* the same automaton walking two characters per step (k_span, build_pair_dfa) */
        uint32_t states2 = 0, accept2 = 0;
        uint16_t *d_dfa2 = nullptr;           /* [states2][36]: next state for (first class, second class) */
        unsigned long long *d_out2 = nullptr; /* [states2][2]: adapters ending on the second / on the first character */
    };
    std::vector<Group> groups;
};

struct sq_pertile {
    sq_ctx *ctx;
    uint32_t tile_changes = 0; /* of the batch in hand: neighbours in stored order with different tiles */
    bool skipped = false;
    std::string skipped_reason;
    uint64_t number_of_reads = 0, max_length = 0, records_seen = 0;
    TileMap map{};
    int n_slots = 0;
    size_t slot_cap = 0, len_cap = 0;
    unsigned long long *d_len_counts = nullptr;
    double *d_errors = nullptr;
    int32_t *d_slots = nullptr;
    long long *d_tiles = nullptr;
    size_t slots_cap = 0;
    unsigned long long *d_first_bad = nullptr;
    int *d_overflow = nullptr;
    uint64_t first_bad = UINT64_MAX;
    /* k_span<PT> (sq_pair.hip): PerTileQuality inside QCMetrics' pass */
    bool runs_ok = true;       /* the batches so far came in runs of one tile (as a sequencer writes): sums per run are staged in the pass */
    bool tiles_ready = false;  /* d_tiles holds the tile ids of the batch in hand (the pass wrote them): pertile_prepare skips k_tile_parse */
    hipEvent_t used = nullptr;   /* behind the kernels of the last call that read d_slots: the next call's pass over the headers, on another stream, overwrites them */
};

/* ---- QCMetrics ------------------------------------------------------------------ */

SQ_EXPORT sq_qcmetrics *sq_qcmetrics_new(sq_ctx *ctx, uint64_t end_anchor_length)
{
    if (end_anchor_length > UINT32_MAX) { /* _qcmodule.c:1832-1837 */
        sq_set_error("end_anchor_length must be between 0 and %lld, got %llu", (long long)UINT32_MAX,
                     (unsigned long long)end_anchor_length);
        return nullptr;
    }
    sq_qcmetrics *m = new sq_qcmetrics();
    m->ctx = ctx;
    m->end_anchor = end_anchor_length;
    size_t ea = end_anchor_length ? end_anchor_length : 1;
    SQ_HIP_NULL(hipMalloc((void **)&m->d_ea_base, ea * 5 * 8));
    SQ_HIP_NULL(hipMalloc((void **)&m->d_ea_phred, ea * 12 * 8));
    SQ_HIP_NULL(hipMalloc((void **)&m->d_gc, 101 * 8));
    SQ_HIP_NULL(hipMalloc((void **)&m->d_ps, 94 * 8));
    SQ_HIP_NULL(hipMalloc((void **)&m->d_thr, 94 * 8));
    SQ_HIP_NULL(hipMalloc((void **)&m->d_thr_sum, 257 * 96 * 8));
    SQ_HIP_NULL(hipMemcpyAsync(m->d_thr_sum, phred_sum_thresholds(), 257 * 96 * 8, hipMemcpyHostToDevice, ctx->stream));
    SQ_HIP_NULL(hipMalloc((void **)&m->d_first_bad, 8));
    SQ_HIP_NULL(hipMemsetAsync(m->d_ea_base, 0, ea * 5 * 8, ctx->stream));
    SQ_HIP_NULL(hipMemsetAsync(m->d_ea_phred, 0, ea * 12 * 8, ctx->stream));
    SQ_HIP_NULL(hipMemsetAsync(m->d_gc, 0, 101 * 8, ctx->stream));
    SQ_HIP_NULL(hipMemsetAsync(m->d_ps, 0, 94 * 8, ctx->stream));
    SQ_HIP_NULL(hipMemsetAsync(m->d_first_bad, 0xFF, 8, ctx->stream));
    SQ_HIP_NULL(hipMemcpyAsync(m->d_thr, phred_thresholds(), 94 * 8, hipMemcpyHostToDevice, ctx->stream));
    SQ_HIP_NULL(hipStreamSynchronize(ctx->stream));
    return m;
}

SQ_EXPORT void sq_qcmetrics_free(sq_qcmetrics *m)
{
    if (!m) return;
    (void)hipStreamSynchronize(m->ctx->stream);
    if (m->polled) (void)hipHostFree(m->polled);
    if (m->poll_event) (void)hipEventDestroy(m->poll_event);
    for (void *p : {(void *)m->d_base, (void *)m->d_phred, (void *)m->d_ea_base, (void *)m->d_ea_phred,
                    (void *)m->d_gc, (void *)m->d_ps, (void *)m->d_thr, (void *)m->d_thr_sum, (void *)m->d_first_bad})
        if (p) (void)hipFree(p);
    delete m;
}

SQ_EXPORT int sq_qcmetrics_reserve(sq_qcmetrics *m, uint64_t length)
{
    /* QCMetrics_resize, _qcmodule.c:1870-1906 */
    int rc = sq_grow_device(m->ctx, &m->d_base, &m->cap, (size_t)length * 5);
    if (rc) return rc;
    return sq_grow_device(m->ctx, &m->d_phred, &m->cap_phred, (size_t)length * 12);
}

/* ---- AdapterCounter: Aho-Corasick automaton over the 5 base classes ------------ */
namespace {

/* The reference scans with bit-parallel shift-AND words (_qcmodule.c:2675-2770);
 * what it reports is, per position, the set of adapters that end there.  A
 * dense automaton gives the same set with one table step per base however
 * many adapters there are.  Row = 8 u16: next-row byte offset | 1 when some
 * adapter ends in the target state; class 5 (padding) returns to the root. */
int build_dfa(const std::vector<std::string> &ads, size_t first, size_t count,
              std::vector<uint16_t> &dfa, std::vector<unsigned long long> &out, uint32_t *n_states,
              uint32_t *n_accept_first)
{
    struct Node { int next[5]; int fail; unsigned long long out; };
    std::vector<Node> t(1);
    for (int c = 0; c < 5; c++) t[0].next[c] = -1;
    t[0].fail = 0;
    t[0].out = 0;
    for (size_t a = 0; a < count; a++) {
        const std::string &s = ads[first + a];
        if (s.empty() || s.find('\0') != std::string::npos)
            continue; /* can never match (populate_bitmask skips NUL, :2455) */
        int cur = 0;
        for (unsigned char ch : s) {
            unsigned l = ch | 0x20u;
            int c = l == 'a' ? 0 : l == 'c' ? 1 : l == 'g' ? 2 : l == 't' ? 3 : 4;
            if (t[cur].next[c] < 0) {
                Node nn;
                for (int k = 0; k < 5; k++) nn.next[k] = -1;
                nn.fail = 0;
                nn.out = 0;
                t[cur].next[c] = (int)t.size();
                t.push_back(nn);
            }
            cur = t[cur].next[c];
        }
        t[cur].out |= 1ULL << a;
    }
    if (t.size() > 4095) return -1;
    /* breadth first: failure links, merged outputs, completed transitions */
    std::vector<int> queue;
    for (int c = 0; c < 5; c++) {
        int v = t[0].next[c];
        if (v < 0) t[0].next[c] = 0;
        else { t[v].fail = 0; queue.push_back(v); }
    }
    for (size_t qi = 0; qi < queue.size(); qi++) {
        int u = queue[qi];
        t[u].out |= t[t[u].fail].out;
        for (int c = 0; c < 5; c++) {
            int v = t[u].next[c];
            if (v < 0) t[u].next[c] = t[t[u].fail].next[c];
            else { t[v].fail = t[t[u].fail].next[c]; queue.push_back(v); }
        }
    }
    /* A state some adapter ends in that leads back to itself (a poly-G probe on the poly-G tail
       of a short insert) would report at every position of the run, and every report sends the
       lanes through the hit path for an adapter they have counted already: only the first hit
       of an adapter in a read counts (:2643-2672).  Such a state gets a twin that reports
       nothing and is where it (and the twin) goes on that character; everything else leaves
       the twin as it leaves the state. */
    for (size_t s = 0, n0 = t.size(); s < n0; s++) {
        if (!t[s].out) continue;
        for (int c = 0; c < 5; c++) {
            if (t[s].next[c] != (int)s) continue;
            Node twin = t[s];
            twin.out = 0;
            twin.next[c] = (int)t.size();
            t[s].next[c] = (int)t.size();
            queue.push_back((int)t.size());
            t.push_back(twin);
            break; /* a state has one character it can loop on: the one all its characters are */
        }
    }
    if (t.size() > 4095) return -1;
    /* outputs of a node are final only after its fail chain is: BFS order guarantees it.
       States some adapter ends in are numbered last, so that "is it a hit" can also be asked
       of a row's position (k_pass) instead of the flag bit */
    /* Rows are numbered breadth first: on random sequence nearly every lane of a wave sits in a
       state one or two characters deep, and rows next to each other lie in different LDS banks
       (numbered adapter by adapter the shallow states were 12 rows apart, every other one on
       the same banks). */
    std::vector<int> number(t.size());
    std::vector<int> bfs(1, 0);
    bfs.insert(bfs.end(), queue.begin(), queue.end());
    int next_number = 0;
    for (int s : bfs) if (!t[s].out) number[s] = next_number++;
    *n_accept_first = (uint32_t)next_number;
    for (int s : bfs) if (t[s].out) number[s] = next_number++;
    dfa.assign(t.size() * 8, 0);
    out.assign(t.size(), 0);
    for (size_t s = 0; s < t.size(); s++) {
        out[number[s]] = t[s].out;
        for (int c = 0; c < 5; c++) {
            int v = t[s].next[c];
            dfa[number[s] * 8 + c] = (uint16_t)((number[v] << 4) | (t[v].out ? 1 : 0));
        }
    }
    *n_states = (uint32_t)t.size();
    return 0;
}

/* The automaton of build_dfa taking two characters per step (k_span: half as many dependent
 * table reads per read).  State n on (c1, c2) goes to f = next[next[n][c1]][c2]; classes are the
 * five of build_dfa plus 5 = padding (back to the root).  What must not get lost is a report of
 * the state in between, m = next[n][c1]: such a step ends in an *alias* of f -- a state of its
 * own that reports the adapters of m one character back (out2[.][1]) besides f's own
 * (out2[.][0]) and goes on like f.  Numbering: build_dfa's states keep their numbers (the ones
 * some adapter ends in last), aliases follow: every state >= accept_first reports something.
 * dfa2[n][c1 + 6 c2] = number of the next state. */
void build_pair_dfa(const std::vector<uint16_t> &dfa, const std::vector<unsigned long long> &out, uint32_t states,
                    std::vector<uint16_t> &dfa2, std::vector<unsigned long long> &out2)
{
    auto next1 = [&](uint32_t n, int c) { return c == 5 ? 0u : (uint32_t)(dfa[n * 8 + c] >> 4); };
    std::vector<uint32_t> base;                                  /* state -> the build_dfa state it behaves like */
    std::vector<std::pair<unsigned long long, uint32_t>> alias;   /* (adapters of the state in between, f) */
    for (uint32_t n = 0; n < states; n++) base.push_back(n);
    out2.clear();
    for (uint32_t n = 0; n < states; n++) { out2.push_back(out[n]); out2.push_back(0); }
    dfa2.clear();
    for (uint32_t n = 0; n < base.size(); n++) {   /* grows while aliases are added: they get rows too */
        const uint32_t b = base[n];
        for (int k = 0; k < 36; k++) {
            const int c1 = k % 6, c2 = k / 6;
            const uint32_t m = next1(b, c1), f = next1(m, c2);
            uint32_t target = f;
            if (out[m]) {
                size_t i = 0;
                while (i < alias.size() && !(alias[i].first == out[m] && alias[i].second == f)) i++;
                if (i == alias.size()) {
                    alias.push_back({out[m], f});
                    base.push_back(f);
                    out2.push_back(out[f]);
                    out2.push_back(out[m]);
                }
                target = states + (uint32_t)i;
            }
            dfa2.push_back((uint16_t)target);
        }
    }
}

} // namespace

/* The tables the kernels walk for the first (up to 64) adapters, for inspection on the host (no
 * GPU involved; tests/test_boundary_cpu.py walks both over random text): dfa [states][8] with
 * entries next << 4 | reports, out [states]; dfa2 [states2][36] and out2 [states2][2] of
 * build_pair_dfa.  Returns the number of states, or -1 when a capacity is too small. */
SQ_EXPORT int64_t sq_adapter_automaton_tables(const char *const *adapters, const size_t *lengths, size_t n,
                                              uint16_t *dfa_o, uint64_t *out_o, size_t cap, uint32_t *accept_first,
                                              uint16_t *dfa2_o, uint64_t *out2_o, size_t cap2, uint32_t *states2)
{
    std::vector<std::string> ads;
    for (size_t i = 0; i < n && i < 64; i++) ads.emplace_back(adapters[i], lengths[i]);
    std::vector<uint16_t> dfa, dfa2;
    std::vector<unsigned long long> out, out2;
    uint32_t states = 0;
    if (build_dfa(ads, 0, ads.size(), dfa, out, &states, accept_first) != 0) return -1;
    build_pair_dfa(dfa, out, states, dfa2, out2);
    *states2 = (uint32_t)(out2.size() / 2);
    if (states > cap || *states2 > cap2) return -1;
    memcpy(dfa_o, dfa.data(), dfa.size() * 2);
    memcpy(out_o, out.data(), out.size() * 8);
    memcpy(dfa2_o, dfa2.data(), dfa2.size() * 2);
    memcpy(out2_o, out2.data(), out2.size() * 8);
    return (int64_t)states;
}

SQ_EXPORT sq_adaptercounter *sq_adaptercounter_new(sq_ctx *ctx, const char *const *adapters,
                                                   const size_t *lengths, size_t n)
{
    if (n < 1) { /* :2484 */
        sq_set_error("At least one adapter is expected");
        return nullptr;
    }
    sq_adaptercounter *a = new sq_adaptercounter();
    a->ctx = ctx;
    for (size_t i = 0; i < n; i++) {
        if (lengths[i] > SQ_MAX_SEQUENCE_SIZE) { /* :2506 */
            sq_set_error("Maximum adapter size is %d, got %zu for '%.*s'", SQ_MAX_SEQUENCE_SIZE,
                         lengths[i], (int)lengths[i], adapters[i]);
            delete a;
            return nullptr;
        }
        a->adapters.emplace_back(adapters[i], lengths[i]);
    }
    for (size_t first = 0; first < n;) {
        size_t count = 0, chars = 0;
        while (first + count < n && count < 64 && chars + lengths[first + count] <= 4000) {
            chars += lengths[first + count];
            count++;
        }
        sq_adaptercounter::Group g;
        g.first = first;
        g.count = count;
        std::vector<uint16_t> dfa;
        std::vector<unsigned long long> out;
        if (build_dfa(a->adapters, first, count, dfa, out, &g.states, &g.accept_first) != 0) {
            sq_set_error("adapter automaton too large");
            delete a;
            return nullptr;
        }
        std::vector<uint8_t> lens(64, 0);
        for (size_t k = 0; k < count; k++) lens[k] = (uint8_t)lengths[first + k];
        SQ_HIP_NULL(hipMalloc((void **)&g.d_dfa, dfa.size() * 2));
        SQ_HIP_NULL(hipMalloc((void **)&g.d_out, out.size() * 8));
        SQ_HIP_NULL(hipMalloc((void **)&g.d_len, 64));
        SQ_HIP_NULL(hipMemcpy(g.d_dfa, dfa.data(), dfa.size() * 2, hipMemcpyHostToDevice));
        SQ_HIP_NULL(hipMemcpy(g.d_out, out.data(), out.size() * 8, hipMemcpyHostToDevice));
        SQ_HIP_NULL(hipMemcpy(g.d_len, lens.data(), 64, hipMemcpyHostToDevice));
        {
            std::vector<uint16_t> dfa2;
            std::vector<unsigned long long> out2;
            build_pair_dfa(dfa, out, g.states, dfa2, out2);
            g.states2 = (uint32_t)(out2.size() / 2);
            g.accept2 = g.accept_first;
            SQ_HIP_NULL(hipMalloc((void **)&g.d_dfa2, dfa2.size() * 2));
            SQ_HIP_NULL(hipMalloc((void **)&g.d_out2, out2.size() * 8));
            SQ_HIP_NULL(hipMemcpy(g.d_dfa2, dfa2.data(), dfa2.size() * 2, hipMemcpyHostToDevice));
            SQ_HIP_NULL(hipMemcpy(g.d_out2, out2.data(), out2.size() * 8, hipMemcpyHostToDevice));
        }
        a->groups.push_back(g);
        first += count;
    }
    return a;
}

SQ_EXPORT void sq_adaptercounter_free(sq_adaptercounter *a)
{
    if (!a) return;
    (void)hipStreamSynchronize(a->ctx->stream);
    for (auto &g : a->groups) {
        (void)hipFree(g.d_dfa);
        (void)hipFree(g.d_dfa2);
        (void)hipFree(g.d_out2);
        (void)hipFree(g.d_out);
        (void)hipFree(g.d_len);
    }
    if (a->d_fwd) (void)hipFree(a->d_fwd);
    if (a->d_rev) (void)hipFree(a->d_rev);
    delete a;
}

/* row-major [rows][old_len] -> [rows][new_len], zero filled */
template <typename T>
static int regrow_rows(sq_ctx *ctx, T **ptr, size_t rows_old, size_t rows_new, size_t len_old,
                       size_t len_new)
{
    T *n = nullptr;
    SQ_HIP(hipMalloc((void **)&n, rows_new * len_new * sizeof(T)));
    SQ_HIP(hipMemsetAsync(n, 0, rows_new * len_new * sizeof(T), ctx->stream));
    if (*ptr && rows_old && len_old)
        SQ_HIP(hipMemcpy2DAsync(n, len_new * sizeof(T), *ptr, len_old * sizeof(T), len_old * sizeof(T),
                                rows_old, hipMemcpyDeviceToDevice, ctx->stream));
    SQ_HIP(hipStreamSynchronize(ctx->stream));
    if (*ptr) SQ_HIP(hipFree(*ptr));
    *ptr = n;
    return SQ_OK;
}

SQ_EXPORT int sq_adaptercounter_reserve(sq_adaptercounter *a, uint64_t length)
{
    /* AdapterCounter_resize, _qcmodule.c:2611-2641 */
    if (length <= a->cap) return SQ_OK;
    size_t want = std::max<size_t>(length, a->cap * 2);
    size_t rows = a->adapters.size();
    int rc = regrow_rows(a->ctx, &a->d_fwd, rows, rows, a->cap, want);
    if (rc) return rc;
    rc = regrow_rows(a->ctx, &a->d_rev, rows, rows, a->cap, want);
    if (rc) return rc;
    a->cap = want;
    return SQ_OK;
}

SQ_EXPORT uint64_t sq_adaptercounter_row_length(sq_adaptercounter *a) { return a->cap; }

SQ_EXPORT int sq_adaptercounter_set_row_length(sq_adaptercounter *a, uint64_t row_length)
{
    if (row_length < a->cap) { sq_set_error("sq_adaptercounter_set_row_length: the tables only grow"); return SQ_ERR_VALUE; }
    if (row_length == a->cap) return SQ_OK;
    size_t rows = a->adapters.size();
    int rc = regrow_rows(a->ctx, &a->d_fwd, rows, rows, a->cap, (size_t)row_length);
    if (rc) return rc;
    rc = regrow_rows(a->ctx, &a->d_rev, rows, rows, a->cap, (size_t)row_length);
    if (rc) return rc;
    a->cap = row_length;
    return SQ_OK;
}

/* ---- PerTileQuality ---------------------------------------------------------------- */

SQ_EXPORT sq_pertile *sq_pertile_new(sq_ctx *ctx)
{
    sq_pertile *p = new sq_pertile();
    p->ctx = ctx;
    SQ_HIP_NULL(hipMalloc((void **)&p->map.keys, TILE_MAP_SIZE * 8));
    SQ_HIP_NULL(hipMalloc((void **)&p->map.vals, TILE_MAP_SIZE * 4));
    SQ_HIP_NULL(hipMalloc((void **)&p->map.n_slots, 4));
    SQ_HIP_NULL(hipMalloc((void **)&p->d_first_bad, 8));
    SQ_HIP_NULL(hipMalloc((void **)&p->d_overflow, 8)); /* [0] map overflow, [1] tile changes of the batch */
    SQ_HIP_NULL(hipMemsetAsync(p->map.keys, 0xFF, TILE_MAP_SIZE * 8, ctx->stream));
    SQ_HIP_NULL(hipMemsetAsync(p->map.vals, 0xFF, TILE_MAP_SIZE * 4, ctx->stream));
    SQ_HIP_NULL(hipMemsetAsync(p->map.n_slots, 0, 4, ctx->stream));
    SQ_HIP_NULL(hipMemsetAsync(p->d_first_bad, 0xFF, 8, ctx->stream));
    SQ_HIP_NULL(hipMemsetAsync(p->d_overflow, 0, 8, ctx->stream));
    SQ_HIP_NULL(hipStreamSynchronize(ctx->stream));
    return p;
}

SQ_EXPORT void sq_pertile_free(sq_pertile *p)
{
    if (!p) return;
    (void)hipStreamSynchronize(p->ctx->stream);
    (void)hipStreamSynchronize(p->ctx->prep_stream);
    if (p->used) (void)hipEventDestroy(p->used);
    for (void *q : {(void *)p->map.keys, (void *)p->map.vals, (void *)p->map.n_slots,
                    (void *)p->d_first_bad, (void *)p->d_overflow, (void *)p->d_len_counts,
                    (void *)p->d_errors, (void *)p->d_slots, (void *)p->d_tiles})
        if (q) (void)hipFree(q);
    delete p;
}

namespace {

int fetch_bytes(sq_batch *b, uint64_t off, uint64_t len, std::string &out)
{
    out.resize(len);
    if (!len) return SQ_OK;
    if (!b->h_buf.empty()) {
        memcpy(&out[0], b->h_buf.data() + off, len);
        return SQ_OK;
    }
    SQ_HIP(hipMemcpy(&out[0], b->d_buf + off, len, hipMemcpyDeviceToHost));
    return SQ_OK;
}

int fetch_meta(sq_batch *b, uint64_t i, sq_meta *m)
{
    if (!b->h_metas.empty()) { *m = b->h_metas[i]; return SQ_OK; }
    SQ_HIP(hipMemcpy(m, b->d_metas + i, sizeof(sq_meta), hipMemcpyDeviceToHost));
    return SQ_OK;
}

/* room for the tile id and the table slot of every record of a batch */
int pertile_reserve_records(sq_pertile *p, uint64_t n)
{
    if (n > p->slots_cap) {
        SQ_HIP(hipStreamSynchronize(p->ctx->stream));
        if (p->ctx->prep_stream) SQ_HIP(hipStreamSynchronize(p->ctx->prep_stream));
        if (p->d_slots) SQ_HIP(hipFree(p->d_slots));
        if (p->d_tiles) SQ_HIP(hipFree(p->d_tiles));
        p->d_slots = nullptr;
        p->d_tiles = nullptr;
        p->slots_cap = 0;
        SQ_HIP(hipMalloc((void **)&p->d_slots, n * 4));
        SQ_HIP(hipMalloc((void **)&p->d_tiles, n * 8));
        p->slots_cap = n;
    }
    return SQ_OK;
}

/* the tables [slot][len] grow to need_slots rows of need_len entries */
int pertile_reserve_tables(sq_pertile *p, size_t need_slots, size_t need_len)
{
    if (need_slots > p->slot_cap || need_len > p->len_cap) {
        size_t ns = std::max(need_slots, p->slot_cap), nl = std::max(need_len, p->len_cap);
        if (need_slots > p->slot_cap) ns = std::max<size_t>(need_slots, p->slot_cap * 2);
        int rc = regrow_rows(p->ctx, &p->d_len_counts, p->slot_cap, ns, p->len_cap, nl);
        if (rc) return rc;
        rc = regrow_rows(p->ctx, &p->d_errors, p->slot_cap, ns, p->len_cap, nl);
        if (rc) return rc;
        p->slot_cap = ns;
        p->len_cap = nl;
    }
    return SQ_OK;
}

/* ---- PerTileQuality inside QCMetrics' pass (k_span<PT>, sq_pair.hip) ----
 * pt_ride_setup fills P.pt_*; the pass parses every header (the first that does not parse lands in p->d_first_bad)
 * and, while the batches come in runs of one tile, stages a sum per run and position.  pt_ride_finish reads back how
 * that went: the runs are folded into the tables -- or dropped, and *done stays false: the caller counts the batch by
 * the older route (a header that does not parse: that route knows where the reference stops, :3137-3148; more runs
 * than the staging area holds: the reads do not come tile by tile, and the following batches only take their tile
 * ids from the pass). */
struct PtRide {
    PtRun *runs = nullptr;
    double *sums = nullptr;
    unsigned int *nruns = nullptr;
    uint32_t cap = 0;
    bool launched = false;
};
/* the overlap scan of InsertSizeMetrics split over the passes of the two mates (sq_paired_add_batches): mode 1: this is
   read 2's pass, which leaves the ends of its reads in `ends`; mode 2: read 1's pass, which scans and leaves the insert
   sizes in `results`; covered: pairs the pass has taken (0: it did not run as k_span<PT>) */
struct PairRide {
    int mode = 0;
    uint8_t *ends = nullptr;
    uint32_t *results = nullptr;
    uint32_t L2 = 0;
    uint64_t covered = 0;
};

int pt_ride_setup(sq_pertile *p, sq_batch *b, PassParams &P, PtRide &R)
{
    sq_ctx *ctx = p->ctx;
    int rc = pertile_reserve_records(p, b->n);
    if (rc) return rc;
    P.pt_bad = p->d_first_bad;
    P.pt_first_index = p->records_seen;
    P.pt_tiles = nullptr;
    P.pt_runs = nullptr;
    if (p->runs_ok && sq_knobs().pt_fused == 1) {
        const uint32_t U = (uint32_t)b->max_length;
        R.cap = (uint32_t)std::min<uint64_t>(65536, std::max<uint64_t>(4096, b->n / 64));
        R.runs = (PtRun *)sq_scratch(ctx, 24, (size_t)R.cap * sizeof(PtRun));
        R.sums = (double *)sq_scratch(ctx, 25, (size_t)R.cap * U * 8);
        R.nruns = (unsigned int *)sq_scratch(ctx, 26, 16);
        if (!R.runs || !R.sums || !R.nruns) { sq_set_error("out of device memory for PerTileQuality's runs"); return SQ_ERR_MEMORY; }
        SQ_HIP(hipMemsetAsync(R.nruns, 0, 16, ctx->stream));
        P.pt_runs = R.runs;
        P.pt_run_sums = R.sums;
        P.pt_nruns = R.nruns;
        P.pt_runs_cap = R.cap;
    } else {
        P.pt_tiles = p->d_tiles;   /* the table by k_ptspan, from these tile ids */
    }
    /* the pass writes d_tiles / reads nothing of d_slots, but the kernels of this object's call before may still read them (on the work stream: in order) */
    return SQ_OK;
}

int pt_ride_finish(sq_pertile *p, sq_batch *b, const PtRide &R, bool *done)
{
    *done = false;
    sq_ctx *ctx = p->ctx;
    if (!R.runs) {             /* tile ids only: the older route goes on from them */
        p->tiles_ready = true;
        return SQ_OK;
    }
    SQ_HIP(hipMemcpyAsync(&ctx->pinned[10], R.nruns, 4, hipMemcpyDeviceToHost, ctx->stream));
    SQ_HIP(hipMemcpyAsync(&ctx->pinned[11], p->d_first_bad, 8, hipMemcpyDeviceToHost, ctx->stream));
    SQ_HIP(hipStreamSynchronize(ctx->stream));
    const uint32_t nruns = (uint32_t)ctx->pinned[10];
    if (ctx->pinned[11] != UINT64_MAX) return SQ_OK;        /* a header that does not parse: the older route, from scratch */
    if (nruns > R.cap) { p->runs_ok = false; return SQ_OK; } /* reads of mixed tiles */
    const uint32_t U = (uint32_t)b->max_length;
    int rc = sq_pt_runs_assign(ctx, R.runs, nruns, p->map.keys, p->map.vals, p->map.n_slots, p->d_overflow);
    if (rc) return rc;
    SQ_HIP(hipMemcpyAsync(&ctx->pinned[10], p->map.n_slots, 4, hipMemcpyDeviceToHost, ctx->stream));
    SQ_HIP(hipMemcpyAsync(&ctx->pinned[11], p->d_overflow, 4, hipMemcpyDeviceToHost, ctx->stream));
    SQ_HIP(hipStreamSynchronize(ctx->stream));
    p->n_slots = (int)(uint32_t)ctx->pinned[10];
    if ((uint32_t)ctx->pinned[11]) {
        sq_set_error("PerTileQuality: more than %u distinct tile ids", TILE_MAP_SIZE / 2);
        return SQ_ERR_MEMORY;
    }
    p->number_of_reads += b->n;
    p->max_length = std::max<uint64_t>(p->max_length, U);
    rc = pertile_reserve_tables(p, (size_t)p->n_slots, (size_t)std::max<uint64_t>(p->max_length, 1));
    if (rc) return rc;
    rc = sq_pt_fold(ctx, R.runs, R.sums, nruns, U, p->d_errors, p->d_len_counts, p->len_cap);
    if (rc) return rc;
    p->records_seen += b->n;
    *done = true;
    return SQ_OK;
}

/* stage 1 of a PerTileQuality add: tile ids, slots, first unparsable header */
int pertile_prepare(sq_pertile *p, sq_batch *b, bool *active)
{
    *active = false;
    if (p->skipped || b->n == 0) return SQ_OK; /* :3126 */
    sq_ctx *ctx = p->ctx;
    {
        int rc = pertile_reserve_records(p, b->n);
        if (rc) return rc;
    }
    int blocks = (int)std::min<uint64_t>((b->n + 255) / 256, 8192);
    /* The pass over the headers runs on a stream of its own: it reads the batch and writes this object's tile
       and slot arrays, nothing the kernels queued on ctx->stream for OTHER batches touch, so it goes beside
       them (the read-back below used to wait for all of them too).  It waits for the kernels of this
       object's call before (they read the slots) and for the batch's upload if that is still on its way. */
    const bool tiles_ready = p->tiles_ready;   /* QCMetrics' pass over this batch has parsed the headers (on ctx->stream) */
    p->tiles_ready = false;
    hipStream_t S = tiles_ready ? ctx->stream : ctx->prep_stream;
    if (S != ctx->stream) {
        if (p->used) SQ_HIP(hipStreamWaitEvent(S, p->used, 0));
        if (b->ready) SQ_HIP(hipStreamWaitEvent(S, b->ready, 0));
    }
    SQ_HIP(hipMemsetAsync(p->d_overflow + 1, 0, 4, S));
    if (!tiles_ready)
        hipLaunchKernelGGL(k_tile_parse, dim3(blocks), dim3(256), 0, S, b->d_buf, (uint64_t)b->buf_len,
                           b->d_metas, (uint64_t)b->n, p->records_seen, p->d_tiles, p->d_first_bad);
    /* a workgroup sets up a 12 KB LDS cache of resolved tiles first: fewer, longer-lived ones */
    const int ablocks = (int)std::min<uint64_t>((b->n + 255) / 256, (uint64_t)ctx->num_cus * 4);
    hipLaunchKernelGGL(k_tile_assign, dim3(ablocks), dim3(256), 0, S, p->d_tiles,
                       (uint64_t)b->n, p->records_seen, p->map, p->d_slots, p->d_first_bad,
                       p->d_overflow);
    SQ_HIP(hipMemcpyAsync(&ctx->pinned[0], p->d_first_bad, 8, hipMemcpyDeviceToHost, S));
    SQ_HIP(hipMemcpyAsync(&ctx->pinned[1], p->map.n_slots, 4, hipMemcpyDeviceToHost, S));
    SQ_HIP(hipMemcpyAsync(&ctx->pinned[2], p->d_overflow, 8, hipMemcpyDeviceToHost, S));
    SQ_HIP(hipStreamSynchronize(S));
    p->first_bad = ctx->pinned[0];
    p->n_slots = (int)(uint32_t)ctx->pinned[1];
    p->tile_changes = (uint32_t)(ctx->pinned[2] >> 32);
    if ((uint32_t)ctx->pinned[2]) {
        sq_set_error("PerTileQuality: more than %u distinct tile ids", TILE_MAP_SIZE / 2);
        return SQ_ERR_MEMORY;
    }
    uint64_t counted = b->n; /* records of this batch in front of the first bad header */
    if (p->first_bad != UINT64_MAX) {
        counted = p->first_bad - p->records_seen;
        sq_meta m;
        std::string name;
        int rc = fetch_meta(b, counted, &m);
        if (rc) return rc;
        rc = fetch_bytes(b, m.record_start, m.name_length, name);
        if (rc) return rc;
        p->skipped = true; /* :3137-3148 */
        p->skipped_reason = "Can not parse header: " + sq_py_repr(name);
    }
    p->number_of_reads += counted;
    /* max_length only moves for records that are counted (:3150) */
    uint64_t maxlen = b->max_length;
    if (counted < b->n) {
        maxlen = 0;
        if (!b->h_metas.empty()) {
            for (uint64_t i = 0; i < counted; i++) maxlen = std::max<uint64_t>(maxlen, b->h_metas[i].sequence_length);
        } else if (counted) {
            std::vector<sq_meta> tmp(counted);
            SQ_HIP(hipMemcpy(tmp.data(), b->d_metas, counted * sizeof(sq_meta), hipMemcpyDeviceToHost));
            for (auto &mm : tmp) maxlen = std::max<uint64_t>(maxlen, mm.sequence_length);
        }
    }
    p->max_length = std::max(p->max_length, maxlen);
    if (counted == 0) return SQ_OK;
    /* tables: [slot][len] */
    {
        int rc = pertile_reserve_tables(p, (size_t)p->n_slots, (size_t)std::max<uint64_t>(p->max_length, 1));
        if (rc) return rc;
    }
    *active = true;
    return SQ_OK;
}

template <bool QC, bool AD, bool PT>
void launch_pass(sq_ctx *ctx, const PassParams &P, bool dfa_lds, int grid, size_t lds)
{
    sq_route(ctx, "k_pass<%s%s%s>", QC ? "QC" : "", AD ? "AD" : "", PT ? "PT" : "");
    if (dfa_lds)
        hipLaunchKernelGGL((k_pass<QC, AD, PT, true>), dim3(grid), dim3(WG_THREADS), lds, ctx->stream, P);
    else
        hipLaunchKernelGGL((k_pass<QC, AD, PT, false>), dim3(grid), dim3(WG_THREADS), lds, ctx->stream, P);
}

void dispatch_pass(sq_ctx *ctx, const PassParams &P, bool qc, bool ad, bool pt, bool dfa_lds, int grid,
                   size_t lds)
{
    int key = (qc ? 4 : 0) | (ad ? 2 : 0) | (pt ? 1 : 0);
    switch (key) {
        case 1: launch_pass<false, false, true>(ctx, P, dfa_lds, grid, lds); break;
        case 2: launch_pass<false, true, false>(ctx, P, dfa_lds, grid, lds); break;
        case 3: launch_pass<false, true, true>(ctx, P, dfa_lds, grid, lds); break;
        case 4: launch_pass<true, false, false>(ctx, P, dfa_lds, grid, lds); break;
        case 5: launch_pass<true, false, true>(ctx, P, dfa_lds, grid, lds); break;
        case 6: launch_pass<true, true, false>(ctx, P, dfa_lds, grid, lds); break;
        case 7: launch_pass<true, true, true>(ctx, P, dfa_lds, grid, lds); break;
        default: break;
    }
}

} // namespace

static int fused_add_batch(sq_batch *b, sq_qcmetrics *m, sq_adaptercounter *a, sq_pertile *p, sq_pertile *ride = nullptr, bool *ride_done = nullptr,
                           PairRide *pair = nullptr);

SQ_EXPORT int sq_fused_add_batch(sq_batch *b, sq_qcmetrics *m, sq_adaptercounter *a, sq_pertile *p)
{
    const SqKnobs &K = sq_knobs();
    /* QCMetrics + AdapterCounter + PerTileQuality on a batch of one read length: k_wide for the
       first two and a PerTileQuality pass of its own (qualities only) read the batch twice and
       still beat the one k_pass that carries all three (3.5 against 3.7 ms per 10 M reads) */
    /* ... and since k_span and k_ptspan (sq_span.hip) QCMetrics + PerTileQuality without the
       adapters too: 2.5 + 1.5 ms per 25 M reads against 7.0 for k_pass<QC,PT> */
    if (m && p && !p->skipped && b->slack && b->n >= 4096 && b->min_length == b->max_length &&
        b->max_length > 0 && b->max_length <= LDS_HIST_MAX && !K.no_wide && !K.ring &&
        (a || (b->max_length <= 32u * SPAN_NW_MAX && K.span))) {
        /* without the adapters: PerTileQuality rides in QCMetrics' pass (sq_pair.hip) -- the tile ids from the header
           bytes the pass fetches anyway, and, while the reads come tile by tile, the table itself */
        bool pt_done = false;
        const bool ride = !a && K.pt_fused && K.span && b->max_length <= 32u * SPAN_NW_MAX;
        int rc = fused_add_batch(b, m, a, nullptr, ride ? p : nullptr, &pt_done);
        if (rc || pt_done) return rc;
        return fused_add_batch(b, nullptr, nullptr, p);
    }
    return fused_add_batch(b, m, a, p);
}

static int fused_add_batch(sq_batch *b, sq_qcmetrics *m, sq_adaptercounter *a, sq_pertile *p, sq_pertile *ride, bool *ride_done, PairRide *pair)
{
    const SqKnobs &K = sq_knobs();
    sq_ctx *ctx = b->ctx;
    bool pt_active = false;
    PtRide R;
    if (p) {
        int rc = pertile_prepare(p, b, &pt_active);
        if (rc) return rc;
    }
    /* wherever this call ends: the kernels it has queued are the last readers of p's slots (pertile_prepare) */
    struct SlotsUsed {
        sq_pertile *p;
        sq_ctx *ctx;
        ~SlotsUsed()
        {
            if (!p) return;
            if (!p->used && hipEventCreateWithFlags(&p->used, hipEventDisableTiming) != hipSuccess) { p->used = nullptr; return; }
            (void)hipEventRecord(p->used, ctx->stream);
        }
    } slots_used{pt_active ? p : nullptr, ctx};
    if (b->n == 0) return SQ_OK;
    PassParams P{};
    P.buf = b->d_buf;
    P.buf_len = b->buf_len;
    P.metas = b->d_metas;
    P.n = b->n;
    if (m) {
        if (b->max_length > m->max_length) {
            int rc = sq_qcmetrics_reserve(m, std::max<uint64_t>(b->max_length, m->max_length * 2));
            if (rc) return rc;
            m->max_length = b->max_length;
        }
        P.first_read_index = m->records_seen;
        P.qc_base = m->d_base; P.qc_phred = m->d_phred;
        P.qc_ea_base = m->d_ea_base; P.qc_ea_phred = m->d_ea_phred;
        P.qc_gc = m->d_gc; P.qc_ps = m->d_ps;
        P.ea_len = (uint32_t)m->end_anchor;
        P.ea_in_lds = m->end_anchor <= LDS_EA_MAX;
        P.thresholds = m->d_thr;
        P.thr_sum = m->d_thr_sum;
        P.qc_first_bad = m->d_first_bad;
    }
    if (p) {
        /* the pass compares first_read_index + r with pt_first_bad; both modules
           count records from their own creation, so carry the difference */
        P.pt_slot = p->d_slots;
        P.pt_len_counts = p->d_len_counts;
        P.pt_errors = p->d_errors;
        P.pt_cap = p->len_cap;
        uint64_t fb = p->first_bad == UINT64_MAX ? UINT64_MAX : p->first_bad - p->records_seen;
        P.pt_first_bad = fb == UINT64_MAX ? UINT64_MAX : fb + P.first_read_index;
    }
    uint32_t stripes = 0; /* 0: one launch holds every read */
    if (!m && pt_active && b->max_length <= LDS_HIST_MAX && b->min_length == b->max_length && b->max_length > 0)
        P.lds_len = P.uniform_len = (uint32_t)b->max_length; /* PerTileQuality alone: its one-tile groups */
    if (m) {
        if (b->max_length <= LDS_HIST_MAX) {
            P.lds_len = (uint32_t)b->max_length; /* every position has its LDS counters */
            if (b->min_length == b->max_length && b->max_length > 0) P.uniform_len = P.lds_len;
        } else {
            /* long reads: stripes of STRIPE positions, one launch each (see PassParams) */
            P.lds_len = STRIPE;
            stripes = (uint32_t)((b->max_length + STRIPE - 1) / STRIPE);
        }
    }
    /* reads of many lengths, none longer than k_span takes: sorted by length inside sq_span_launch_sorted */
    const bool span_sorted =
        m && !pt_active && !stripes && !P.uniform_len && b->slack && b->min_length >= 1 && b->n < (1ull << 31) &&
        (!a || a->groups[0].states <= DFA_LDS_MAX_STATES) && b->max_length <= 32u * (a ? (K.span_split ? SPAN_NW_AD_SPLIT : SPAN_NW_AD) : SPAN_NW_MAX) &&
        K.span && !K.ring && !K.no_ring && !K.no_wide &&
        (K.span_sorted >= 0 ? K.span_sorted != 0 : b->n >= 65536);
    /* PerTileQuality alone on a batch of one read length whose table fits LDS: k_ptspan streams the
       batch as it lies (no sort by tile) */
    uint64_t pt_covered = 0;
    const bool ptspan = !m && !a && pt_active && P.uniform_len && b->slack && b->n >= 64 && !stripes &&
                        p->n_slots > 0 && !K.no_ptq && K.span;
    if (ptspan) {
        int rc = sq_ptspan_launch(ctx, P, (uint32_t)p->n_slots, &pt_covered);
        if (rc) return rc;
        if (pt_covered == b->n) {   /* nothing left for the passes below; m and a are null here */
            p->records_seen += b->n;
            return SQ_OK;
        }
    }
    P.pos_end = UINT32_MAX;
    if (b->n >= 4096 && b->n < (1ull << 31)) {
        if (pt_active) {
            /* A sequencer writes its reads tile by tile: such a batch is walked as stored (a wave
               takes a contiguous run of groups, and nearly every group is of one tile).  Only a
               batch whose tiles are mixed (on average fewer than 256 reads between two changes)
               is walked in tile-sorted order, which costs the sort and makes every load a gather */
            if (!ptspan && (uint64_t)p->tile_changes * 256 > b->n)
                P.order = sorted_order(ctx, b, p->d_slots, (uint32_t)p->n_slots);
            P.blocked = P.order != nullptr; /* stored order: waves move through the batch together */
        }
        else if ((m || a) && b->max_length > 2 * b->min_length + 64 && !span_sorted)
            P.order = sorted_order(ctx, b, nullptr, (uint32_t)b->max_length);
    }
    const uint32_t ea_rows = (m && P.ea_in_lds && !P.uniform_len) ? P.ea_len : 0;
    if (a) {
        if (b->max_length > a->max_length) {
            int rc = sq_adaptercounter_reserve(a, b->max_length);
            if (rc) return rc;
            a->max_length = b->max_length;
        }
        P.ad_cap = a->cap;
    }
    /* long reads.  Sorted longest first and without PerTileQuality: the segment kernels
       (k_read_sums + k_seg + k_adapter_first).  Else stripes: one launch per 512 positions,
       the per-read state carried from launch to launch. */
    const bool segments = stripes && P.order && !P.blocked && !pt_active && !K.no_segments &&
                          (!a || a->groups[0].states <= DFA_LDS_MAX_STATES);
    const uint32_t span = segments ? SEG : STRIPE;
    if (stripes) stripes = (uint32_t)((b->max_length + span - 1) / span);
    std::vector<uint64_t> stripe_reads(std::max(stripes, 1u), b->n);
    unsigned long long *d_stripe_reads = nullptr;
    sq_carry *d_carry = nullptr;
    if (stripes) {
        if (!segments) {
            d_carry = (sq_carry *)sq_scratch(ctx, 5, b->n * sizeof(sq_carry));
            if (!d_carry) { sq_set_error("out of device memory for the long-read state"); return SQ_ERR_MEMORY; }
        }
        if (P.order && !P.blocked) { /* longest first: the reads that reach a stripe are a prefix */
            d_stripe_reads = (unsigned long long *)sq_scratch(ctx, 4, stripes * 8);
            if (!d_stripe_reads) { sq_set_error("out of device memory"); return SQ_ERR_MEMORY; }
            hipLaunchKernelGGL(k_stripe_counts, dim3((stripes + 255) / 256), dim3(256), 0, ctx->stream, b->d_metas,
                               P.order, (uint64_t)b->n, span, stripes, d_stripe_reads);
            SQ_HIP(hipMemcpyAsync(stripe_reads.data(), d_stripe_reads, stripes * 8, hipMemcpyDeviceToHost, ctx->stream));
            SQ_HIP(hipStreamSynchronize(ctx->stream));
        }
    }
    /* PerTileQuality alone on a batch of one read length: k_ptq for the full groups, k_pass for
       a trailing partial one (SQ_NO_PTQ=1: k_pass for all) */
    bool ptq_done = false;
    if (pt_covered) {   /* k_ptspan took the full spans: the last few records go through k_pass */
        P.order = nullptr;
        P.metas += pt_covered;
        P.pt_slot += pt_covered;
        P.first_read_index += pt_covered;
        P.n = b->n - pt_covered;
    } else if (!m && !a && pt_active && P.uniform_len && b->slack && b->n >= 64 && !stripes && !K.no_ptq &&
        ptq_lds_bytes(P.uniform_len) <= 80 * 1024) {
        PassParams C = P;
        C.n = (b->n / 64) * 64;
        static bool pattr = false;
        if (!pattr) {
            SQ_HIP(hipFuncSetAttribute((const void *)k_ptq, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
            pattr = true;
        }
        const uint64_t want = (C.n / 64 + PTQ_WAVES - 1) / PTQ_WAVES;
        const int pgrid = (int)std::max<uint64_t>(1, std::min<uint64_t>(want, (uint64_t)ctx->num_cus * 2));
        sq_route(ctx, "k_ptq");
        hipLaunchKernelGGL(k_ptq, dim3(pgrid), dim3(PTQ_THREADS), ptq_lds_bytes(P.uniform_len), ctx->stream, C);
        SQ_HIP(hipGetLastError());
        if (C.n == b->n) {
            ptq_done = true;
        } else if (P.order) {      /* the walk goes on behind the full groups */
            P.order += C.n;
            P.n = b->n - C.n;
        } else {
            P.metas += C.n;
            P.pt_slot += C.n;
            P.first_read_index += C.n;
            P.n = b->n - C.n;
        }
    }
    /* first automaton rides with the other modules; further groups get a pass of their own */
    size_t ngroups = a ? a->groups.size() : 0;
    for (size_t gi = 0; !ptq_done && (gi == 0 || gi < ngroups); gi++) {
        bool qc = m && gi == 0, pt = pt_active && gi == 0, ad = a != nullptr;
        bool dfa_lds = false;
        uint32_t states = 0;
        if (ad) {
            auto &g = a->groups[gi];
            P.dfa = g.d_dfa; P.dfa_states = g.states; P.dfa_accept = g.accept_first; P.dfa_out = g.d_out; P.ad_len = g.d_len;
            P.dfa2 = g.d_dfa2; P.dfa2_states = g.states2; P.dfa2_out = g.d_out2;
            P.ad_fwd = a->d_fwd + g.first * a->cap;
            P.ad_rev = a->d_rev + g.first * a->cap;
            states = g.states;
            dfa_lds = states <= DFA_LDS_MAX_STATES;
            /* one read length: hits are counted per workgroup in LDS (a handful of adapters x
               positions) and reach the device tables once, reverse positions derived there */
            P.ad_lds = 0;
            if (qc && !pt && dfa_lds && P.uniform_len && g.count * hist_stride(P.lds_len) * 4 <= 8192)
                P.ad_lds = (uint32_t)g.count;
            P.ad_maxlen = 0;
            for (size_t k = 0; k < g.count; k++)
                P.ad_maxlen = std::max<uint32_t>(P.ad_maxlen, (uint32_t)a->adapters[g.first + k].size());
        }
        if (!qc && !pt && !ad) break;
        /* Batches of one read length (<= 512) in stored order skip k_pass's general machinery; a
           trailing partial group of fewer than 64 records goes through k_pass.
           - QCMetrics + AdapterCounter: k_wide (64 bytes per row and visit).  Its 16-byte loads may
             run past the last record, so the buffer must be the library's own (64 bytes of slack);
             wrapped device memory keeps k_pass.  SQ_NO_WIDE=1: k_pass; SQ_WIDE=1: also QCMetrics alone.
           - QCMetrics alone: k_ring (every 64-byte sector fetched once), 8 % ahead of k_wide there.
             SQ_NO_RING=1: k_pass; SQ_RING=1: also with the automaton. */
        /* Reads of many lengths, none longer than k_span takes (what adapter trimming leaves of a
           file of one read length): sorted by length and cut into spans of 16 reads of one length,
           they are batches of one read length to k_span, one length after the other.  SQ_SPAN=0 or
           SQ_SPAN_SORTED=0: the general k_pass. */
        if (span_sorted && qc && !pt && (!ad || dfa_lds)) {
            PassParams S = P;
            S.order = nullptr;
            if (ad && dfa_lds && a->groups[gi].count * hist_stride(32u * (((uint32_t)b->max_length + 31) / 32)) * 4 <= 8192)
                S.ad_lds = (uint32_t)a->groups[gi].count;
            uint64_t covered = 0;
            int rc = sq_span_launch_sorted(ctx, S, ad, ad ? (uint32_t)a->groups[gi].count : 0, (uint32_t)b->min_length, (uint32_t)b->max_length,
                                           S.n == b->n && S.metas == b->d_metas && b->len_hist.size() == SQ_LEN_BINS ? b->len_hist.data() : nullptr, &covered);
            if (rc) return rc;
            if (covered == b->n) continue;
        }
        PassParams Pfull = P;
        const bool uniform_fast = qc && !pt && P.uniform_len && !P.order && b->n >= 64 && (!ad || dfa_lds);
        const size_t wlds = wide_lds_bytes(P.uniform_len, ad, states, ad ? P.ad_lds : 0);
        const bool wide = uniform_fast && b->slack && wlds <= 160 * 1024 && !K.no_wide &&
                          ad && !K.ring;
        const size_t rlds = ring_lds_bytes(P.uniform_len, ad, states, ad ? P.ad_lds : 0);
        const bool ring = uniform_fast && !wide && rlds <= 160 * 1024 && !K.no_ring &&
                          (!ad || K.ring);
        /* - k_span (sq_span.hip) in front of both: the records come through LDS by LDS-DMA, 16 per
             wave at a time, four lanes per read: QCMetrics alone up to 256 positions (1495
             Gbases/s at 150 against k_ring's 1020), with the automaton up to 160 (1015 against
             k_wide's 1000; adapters of up to 13 characters).  SQ_SPAN=0: the other two. */
        bool span_done = false;
        if (uniform_fast && b->slack && K.span && !K.ring && !K.no_ring && !K.no_wide) {
            uint64_t covered = 0;
            int rc = SQ_OK;
            if (ride && !ad && !R.launched) {
                PassParams C = P;
                rc = pt_ride_setup(ride, b, C, R);
                if (rc) return rc;
                if (pair && pair->mode) { C.pair_ends = pair->ends; C.pair_results = pair->results; C.pair_L2 = pair->L2; }
                rc = sq_span_launch_pt(ctx, C, pair ? pair->mode : 0, &covered);
                R.launched = covered != 0;
                if (!R.launched) R = PtRide();
                if (pair) pair->covered = covered;
            }
            if (!rc && !covered) rc = sq_span_launch(ctx, P, ad, ad ? (uint32_t)a->groups[gi].count : 0, &covered);
            if (rc) return rc;
            if (covered == b->n) continue;
            if (covered) {
                span_done = true;
                P.metas = b->d_metas + covered;
                P.first_read_index += covered;
                P.n = b->n - covered;
            }
        }
        if (wide && !span_done) {
            PassParams C = P;
            C.n = (b->n / 64) * 64;
            static bool wattr = false;
            if (!wattr) {
                SQ_HIP(hipFuncSetAttribute((const void *)k_wide<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                wattr = true;
            }
            const uint64_t want = (C.n / 64 + WIDE_WAVES - 1) / WIDE_WAVES;
            const int wgrid = (int)std::max<uint64_t>(1, std::min<uint64_t>(want, (uint64_t)ctx->num_cus));
            sq_route(ctx, "k_wide<AD>");   /* (QCMetrics alone never comes here: k_span, behind it k_ring) */
            hipLaunchKernelGGL((k_wide<true>), dim3(wgrid), dim3(WIDE_THREADS), wlds, ctx->stream, C);
            SQ_HIP(hipGetLastError());
            if (C.n == b->n) continue;
            P.metas = b->d_metas + C.n;
            P.first_read_index += C.n;
            P.n = b->n - C.n;
        }
        if (ring && !span_done) {
            PassParams C = P;
            C.n = (b->n / 64) * 64;
            const int per_cu = rlds <= 80 * 1024 ? 2 : 1;
            const uint64_t want = (C.n / 64 + RING_WAVES - 1) / RING_WAVES;
            const int rgrid = (int)std::max<uint64_t>(1, std::min<uint64_t>(want, (uint64_t)ctx->num_cus * per_cu));
            static bool attr_set = false;
            if (!attr_set) {
                SQ_HIP(hipFuncSetAttribute((const void *)k_ring<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                SQ_HIP(hipFuncSetAttribute((const void *)k_ring<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                attr_set = true;
            }
            sq_route(ctx, "k_ring<%s>", ad ? "AD" : "QC");
            if (ad) hipLaunchKernelGGL((k_ring<true>), dim3(rgrid), dim3(RING_THREADS), rlds, ctx->stream, C);
            else hipLaunchKernelGGL((k_ring<false>), dim3(rgrid), dim3(RING_THREADS), rlds, ctx->stream, C);
            SQ_HIP(hipGetLastError());
            if (C.n == b->n) continue;
            P.metas = b->d_metas + C.n;
            P.first_read_index += C.n;
            P.n = b->n - C.n;
        }
        size_t lds = pass_lds_bytes(qc, qc ? P.lds_len : 0, qc ? ea_rows : 0, dfa_lds, states, ad ? P.ad_lds : 0, pt ? P.lds_len : 0);
        int wgs_per_cu = (int)std::max<size_t>(1, std::min<size_t>(4, (160 * 1024) / lds));
        if (qc && segments) {
            /* (1) what is sequential per read; where k_span<LONG> takes the per-position pass it counts G/C per
               read on its way and this pass reads the qualities only */
            const bool try_long = K.long_spans && K.span && b->slack &&
                                  sq_span_long_takes(P, ad, ad ? (uint32_t)a->groups[0].count : 0, (uint32_t)b->max_length);
            sq_route(ctx, try_long ? "k_read_sums<qualities>" : "k_read_sums<GC>");
            if (try_long) {
                hipLaunchKernelGGL(k_read_sums<false>, dim3((unsigned)std::min<uint64_t>((b->n + 63) / 64, 4096)), dim3(256), 0,
                                   ctx->stream, P);
            } else
                hipLaunchKernelGGL(k_read_sums<true>, dim3((unsigned)std::min<uint64_t>((b->n + 63) / 64, 4096)), dim3(256), 0,
                                   ctx->stream, P);
            /* (2) everything per position.  k_span<LONG> (sq_span.hip) streams segments of 128 positions
               through LDS; it sends the padding behind a read's end to a histogram row of its own, where a
               quality byte 0x80 would land too: a batch k_read_sums has flagged (an invalid phred byte,
               one 8-byte read-back) keeps k_seg, whose counts k_qc_uncount knows how to take back.
               SQ_LONG=0: k_seg for all */
            const uint32_t n_ad_long = ad ? (uint32_t)a->groups[0].count : 0;
            if (try_long) {
                SQ_HIP(hipMemcpyAsync(&ctx->pinned[9], m->d_first_bad, 8, hipMemcpyDeviceToHost, ctx->stream));
                SQ_HIP(hipStreamSynchronize(ctx->stream));
                if (ctx->pinned[9] == UINT64_MAX) {
                    PassParams Lp = P;
                    if (ad) {
                        Lp.long_first = (unsigned int *)sq_scratch(ctx, 5, b->n * n_ad_long * 4);
                        if (!Lp.long_first) { sq_set_error("out of device memory for the adapter candidates"); return SQ_ERR_MEMORY; }
                        SQ_HIP(hipMemsetAsync(Lp.long_first, 0xFF, b->n * n_ad_long * 4, ctx->stream));
                    }
                    uint64_t covered = 0;
                    int rc = sq_span_launch_long(ctx, Lp, ad, n_ad_long, (uint32_t)b->max_length, &covered);
                    if (rc) return rc;
                    if (covered == b->n) {
                        if (ad)
                            hipLaunchKernelGGL(k_adapter_first, dim3((unsigned)std::min<uint64_t>((b->n * n_ad_long + 255) / 256, 8192)),
                                               dim3(256), 0, ctx->stream, Lp.long_first, b->d_metas, (uint64_t)b->n, n_ad_long, P.ad_len,
                                               P.ad_fwd, P.ad_rev, P.ad_cap);
                        SQ_HIP(hipGetLastError());
                        continue;
                    }
                }
                /* not taken after all (a flagged batch, or no memory for its tables): the G/C histogram the
                   pass above left out */
                hipLaunchKernelGGL(k_read_gc, dim3((unsigned)std::min<uint64_t>((b->n + 3) / 4, 4096)), dim3(256), 0, ctx->stream, P);
            }
            std::vector<uint32_t> table;
            for (uint32_t w = 0; w < stripes && stripe_reads[w]; w++) {
                const uint64_t groups = (stripe_reads[w] + 63) / 64;
                for (uint64_t g0 = 0; g0 < groups; g0 += 64) {
                    table.push_back(w);
                    table.push_back((uint32_t)g0);
                    table.push_back((uint32_t)std::min<uint64_t>(64, groups - g0));
                }
            }
            uint32_t *d_table = (uint32_t *)sq_scratch(ctx, 0, table.size() * 4 + 16);
            if (!d_table) { sq_set_error("out of device memory"); return SQ_ERR_MEMORY; }
            SQ_HIP(hipMemcpyAsync(d_table, table.data(), table.size() * 4, hipMemcpyHostToDevice, ctx->stream));
            SegParams S{};
            S.wg_table = d_table;
            S.seg_reads = d_stripe_reads;
            const uint32_t n_ad = ad ? (uint32_t)a->groups[0].count : 0;
            if (ad) {
                S.n_adapters = n_ad;
                S.first = (unsigned int *)sq_scratch(ctx, 5, b->n * n_ad * 4);
                if (!S.first) { sq_set_error("out of device memory for the adapter candidates"); return SQ_ERR_MEMORY; }
                SQ_HIP(hipMemsetAsync(S.first, 0xFF, b->n * n_ad * 4, ctx->stream));
            }
            const size_t slds = seg_lds_bytes(ea_rows, ad, states);
            const unsigned sgrid = (unsigned)(table.size() / 3);
            sq_route(ctx, "k_seg<%s>", ad ? "AD" : "QC");
            if (ad) hipLaunchKernelGGL((k_seg<true>), dim3(sgrid), dim3(WG_THREADS), slds, ctx->stream, P, S);
            else hipLaunchKernelGGL((k_seg<false>), dim3(sgrid), dim3(WG_THREADS), slds, ctx->stream, P, S);
            SQ_HIP(hipGetLastError());
            /* (3) the first occurrence of every adapter in every read */
            if (ad)
                hipLaunchKernelGGL(k_adapter_first, dim3((unsigned)std::min<uint64_t>((b->n * n_ad + 255) / 256, 8192)),
                                   dim3(256), 0, ctx->stream, S.first, b->d_metas, (uint64_t)b->n, n_ad, P.ad_len,
                                   P.ad_fwd, P.ad_rev, P.ad_cap);
            /* the host vector goes out of scope: the copy must have left it */
            SQ_HIP(hipStreamSynchronize(ctx->stream));
        } else if (qc && stripes) {
            const uint64_t n_all = P.n;
            for (uint32_t w = 0; w < stripes && stripe_reads[w]; w++) {
                P.pos_base = w * STRIPE;
                P.pos_end = (w + 1) * STRIPE;
                P.carry = d_carry;
                P.n = std::min<uint64_t>(stripe_reads[w], n_all);
                dispatch_pass(ctx, P, qc, ad, pt, dfa_lds, grid_for(ctx, P.n, wgs_per_cu), lds);
                SQ_HIP(hipGetLastError());
            }
            P.n = n_all;
            P.pos_base = 0;
            P.pos_end = UINT32_MAX;
            P.carry = nullptr;
        } else {
            dispatch_pass(ctx, P, qc, ad, pt, dfa_lds, grid_for(ctx, P.n, wgs_per_cu), lds);
            SQ_HIP(hipGetLastError());
        }
        if (ring || wide || span_done) P = Pfull;
    }
    if (m) {
        m->number_of_reads += b->n; m->records_seen += b->n;
        if (m->seen.size() >= 4096) { m->max_length_flushed = m->max_length; m->seen.clear(); } /* a caller that never flushes */
        m->seen.push_back({b->id, b->max_length});
    }
    if (a) a->number_of_sequences += b->n;
    if (p) p->records_seen += b->n;
    if (R.launched) return pt_ride_finish(ride, b, R, ride_done);
    return SQ_OK;
}

/* One call for what the reference's driver does with a pair of arrays (__main__.py:279-306): QCMetrics + PerTileQuality on
 * read 1, the same on read 2, InsertSizeMetrics on the pair -- the results of the five add_record_array calls in that
 * order.  Batches of one read length each (what a sequencer writes) take two passes instead of seven: read 2's pass
 * leaves the first and the last 16 bases of every read behind (32 bytes per pair), read 1's pass scans the sequences it
 * holds in LDS anyway against them (calculate_insert_size :5667-5707), both carry PerTileQuality (sq_pair.hip).  Anything
 * else -- other lengths, a module that is NULL or has stopped, SQ_PT_FUSED != 1 -- is the five calls.  Any of m1, p1, m2,
 * p2, z may be NULL.
 * On an error (out of device memory, a HIP error) the call returns at once and the modules' state is that of SOME prefix of
 * the work, not of the reference's five calls cut at the same point: the two-pass route counts read 2 first (its ends are
 * read 1's needles), so read 2's modules may have counted a batch read 1's have not.  Results after a non-zero return are
 * undefined here as they are partial in the reference (_qcmodule.c:2196-2203 returns NULL mid-array too). */
SQ_EXPORT int sq_paired_add_batches(sq_batch *b1, sq_batch *b2, sq_qcmetrics *m1, sq_pertile *p1, sq_qcmetrics *m2,
                                    sq_pertile *p2, sq_insertsize *z)
{
    const SqKnobs &K = sq_knobs();
    auto side = [](sq_batch *b, sq_qcmetrics *m, sq_pertile *p) -> int { return (m || p) ? sq_fused_add_batch(b, m, nullptr, p) : SQ_OK; };
    auto uniform = [](const sq_batch *b) { return b->slack && b->n >= 4096 && b->min_length == b->max_length && b->max_length >= 16 && b->max_length <= 32u * SPAN_NW_MAX; };
    const bool fuse = K.pt_fused == 1 && K.span && !K.no_wide && !K.ring && m1 && p1 && m2 && p2 && z && !p1->skipped && !p2->skipped &&
                      b1->n == b2->n && uniform(b1) && uniform(b2) && b1->n < (1ull << 32);
    if (!fuse) {
        int rc = side(b1, m1, p1);
        if (rc) return rc;
        rc = side(b2, m2, p2);
        if (rc) return rc;
        return z ? sq_insertsize_add_batch_pair(z, b1, b2) : SQ_OK;
    }
    sq_ctx *ctx = b1->ctx;
    int rc = sq_insertsize_reserve_for(z, b1, b2);
    if (rc) return rc;
    PairRide r2, r1;
    r2.mode = 1;
    r2.ends = (uint8_t *)sq_scratch(ctx, 28, (size_t)b2->n * 32);
    r1.mode = 2;
    r1.ends = r2.ends;
    r1.results = sq_insertsize_scan_results(z, b1->n);
    r1.L2 = (uint32_t)b2->max_length;
    if (!r2.ends || !r1.results) { sq_set_error("out of device memory for the paired pass"); return SQ_ERR_MEMORY; }
    /* read 2 first: its ends are read 1's needles.  (The modules see the same records as in the reference's order; none
       of their results depends on which mate was counted first.) */
    bool pt_done = false;
    rc = fused_add_batch(b2, m2, nullptr, nullptr, p2, &pt_done, &r2);
    if (rc) return rc;
    if (!pt_done) { rc = fused_add_batch(b2, nullptr, nullptr, p2); if (rc) return rc; }
    if (!r2.covered) r1.mode = 0;   /* read 2's pass did not run as k_span<PT>: no ends, the scan keeps its own kernel */
    pt_done = false;
    rc = fused_add_batch(b1, m1, nullptr, nullptr, p1, &pt_done, &r1);
    if (rc) return rc;
    if (!pt_done) { rc = fused_add_batch(b1, nullptr, nullptr, p1); if (rc) return rc; }
    if (r1.mode == 2 && r1.covered && r1.covered <= r2.covered)
        return sq_insertsize_add_batch_pair_scanned(z, b1, b2, r1.results, r1.covered);
    return sq_insertsize_add_batch_pair(z, b1, b2);
}

SQ_EXPORT int sq_qcmetrics_add_batch(sq_qcmetrics *m, sq_batch *b) { return sq_fused_add_batch(b, m, nullptr, nullptr); }
SQ_EXPORT int sq_adaptercounter_add_batch(sq_adaptercounter *a, sq_batch *b) { return sq_fused_add_batch(b, nullptr, a, nullptr); }
SQ_EXPORT int sq_pertile_add_batch(sq_pertile *p, sq_batch *b) { return sq_fused_add_batch(b, nullptr, nullptr, p); }

/* ---- QCMetrics: flush, errors, getters ----------------------------------------------- */

SQ_EXPORT int sq_qcmetrics_flush(sq_qcmetrics *m)
{
    sq_ctx *ctx = m->ctx;
    SQ_HIP(hipMemcpyAsync(&ctx->pinned[8], m->d_first_bad, 8, hipMemcpyDeviceToHost, ctx->stream));
    SQ_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->pinned[8] != UINT64_MAX) {
        /* _qcmodule.c:2102-2105, deferred: every batch since the last flush has been applied in
           full.  The caller names the batches (sq_batch_first_invalid_phred) and has their tails
           taken back (sq_qcmetrics_uncount_tail); the flag is rearmed, the object stays usable as
           the reference's does */
        const unsigned long long bad = ctx->pinned[8];
        SQ_HIP(hipMemsetAsync(m->d_first_bad, 0xFF, 8, ctx->stream));
        sq_set_error("Not a valid phred character in record %llu", bad);
        return SQ_ERR_VALUE;
    }
    m->max_length_flushed = m->max_length;
    m->seen.clear();
    return SQ_OK;
}

/* Without waiting: have the passes that were queued when the poll was armed ended, and with no invalid phred character
 * flagged?  The first call arms the poll (0); later calls answer 0 (not yet), 1 (yes: the arrays handed in before the
 * arming call are counted for good, the caller may let go of them; the poll is disarmed) or -1 (a character is flagged:
 * sq_qcmetrics_flush and what follows it see to that; disarmed).  Nothing else changes: a caller that never polls keeps
 * its arrays until it flushes, as before. */
SQ_EXPORT int sq_qcmetrics_poll(sq_qcmetrics *m)
{
    sq_ctx *ctx = m->ctx;
    if (!m->poll_armed) {
        if (!m->polled && hipHostMalloc((void **)&m->polled, 8, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); m->polled = nullptr; return 0; }
        if (!m->poll_event && hipEventCreateWithFlags(&m->poll_event, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); m->poll_event = nullptr; return 0; }
        if (hipMemcpyAsync(m->polled, m->d_first_bad, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipEventRecord(m->poll_event, ctx->stream) != hipSuccess) { (void)hipGetLastError(); return 0; }
        m->poll_armed = true;
        return 0;
    }
    if (hipEventQuery(m->poll_event) != hipSuccess) { (void)hipGetLastError(); return 0; }
    m->poll_armed = false;
    return *m->polled == UINT64_MAX ? 1 : -1;
}

/* index of the first record among [start, end) of b with a quality byte outside 33 .. 126, -1: none */
SQ_EXPORT int64_t sq_batch_first_invalid_phred(sq_batch *b, uint64_t start, uint64_t end)
{
    if (end > b->n) end = b->n;
    if (start >= end) return -1;
    sq_ctx *ctx = b->ctx;
    unsigned long long *d = (unsigned long long *)sq_scratch(ctx, 0, 64);
    if (!d) { sq_set_error("out of device memory"); return SQ_ERR_MEMORY; }
    SQ_HIP(hipMemsetAsync(d, 0xFF, 8, ctx->stream));
    hipLaunchKernelGGL(k_first_invalid, dim3((unsigned)std::min<uint64_t>((end - start + 255) / 256, 4096)), dim3(256), 0,
                       ctx->stream, b->d_buf, b->d_metas, start, end, d);
    SQ_HIP(hipMemcpyAsync(&ctx->pinned[9], d, 8, hipMemcpyDeviceToHost, ctx->stream));
    SQ_HIP(hipStreamSynchronize(ctx->stream));
    return ctx->pinned[9] == UINT64_MAX ? -1 : (int64_t)ctx->pinned[9];
}

/* Records [first, end) of b went through this object since its last successful flush as the tail
 * of one call, and record `first` holds an invalid phred byte: leaves the tables as the
 * reference's are behind that call's ValueError (see k_qc_uncount).  kept_max_length: the
 * longest read of b among the records that stay counted.  Call sq_qcmetrics_flush again when
 * every such stretch has been handled. */
SQ_EXPORT int sq_qcmetrics_uncount_tail(sq_qcmetrics *m, sq_batch *b, uint64_t first, uint64_t end, uint64_t kept_max_length)
{
    if (end > b->n) end = b->n;
    if (first >= end) { sq_set_error("sq_qcmetrics_uncount_tail: record index out of range"); return SQ_ERR_VALUE; }
    sq_ctx *ctx = m->ctx;
    PassParams P{};
    P.buf = b->d_buf; P.buf_len = b->buf_len; P.metas = b->d_metas; P.n = end;
    P.qc_base = m->d_base; P.qc_phred = m->d_phred; P.qc_ea_base = m->d_ea_base; P.qc_ea_phred = m->d_ea_phred;
    P.qc_gc = m->d_gc; P.qc_ps = m->d_ps; P.ea_len = (uint32_t)m->end_anchor; P.thresholds = m->d_thr; P.thr_sum = m->d_thr_sum;
    const uint64_t tail = end - first;
    hipLaunchKernelGGL(k_qc_uncount, dim3((unsigned)std::min<uint64_t>((tail + 255) / 256, 4096)), dim3(256), 0, ctx->stream, P, first);
    SQ_HIP(hipGetLastError());
    SQ_HIP(hipStreamSynchronize(ctx->stream));
    m->number_of_reads -= tail - 1;
    uint64_t ml = m->max_length_flushed;
    bool found = false;
    for (auto &sn : m->seen) {
        if (sn.id == b->id && !found) { sn.max_length = kept_max_length; found = true; }
        ml = std::max(ml, sn.max_length);
    }
    if (found) m->max_length = ml;
    return SQ_OK;
}

SQ_EXPORT int sq_qcmetrics_add(sq_qcmetrics *m, const uint8_t *buf, size_t buf_len, sq_meta *metas, size_t n)
{
    sq_batch *b = sq_batch_upload(m->ctx, buf, buf_len, metas, n);
    if (!b) return SQ_ERR_MEMORY;
    int rc = sq_qcmetrics_add_batch(m, b);
    if (rc == SQ_OK) {
        /* surface an invalid phred byte now, with the reference's message and its partial state */
        sq_ctx *ctx = m->ctx;
        hipError_t e = hipMemcpyAsync(&ctx->pinned[8], m->d_first_bad, 8, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess && ctx->pinned[8] != UINT64_MAX) {
            (void)hipMemsetAsync(m->d_first_bad, 0xFF, 8, ctx->stream);
            const int64_t idx = sq_batch_first_invalid_phred(b, 0, n);
            char bad = '?';
            if (idx >= 0) {
                uint64_t kept = 0;
                for (int64_t i = 0; i <= idx; i++) kept = std::max<uint64_t>(kept, metas[i].sequence_length);
                rc = sq_qcmetrics_uncount_tail(m, b, (uint64_t)idx, n, kept);
                const uint8_t *q = buf + metas[idx].record_start + metas[idx].qualities_offset;
                for (uint32_t k = 0; k < metas[idx].sequence_length; k++)
                    if ((uint8_t)(q[k] - 33) > SQ_PHRED_MAX) { bad = (char)q[k]; break; }
            }
            if (rc == SQ_OK) {
                sq_set_error("Not a valid phred character: %c", bad);
                rc = SQ_ERR_VALUE;
            }
        }
    }
    if (n && (rc == SQ_OK || rc == SQ_ERR_VALUE)) {
        std::vector<double> errs(n);
        int rc2 = sq_batch_error_rates(b, errs.data(), n);
        for (size_t i = 0; i < n && rc2 == SQ_OK; i++) metas[i].accumulated_error_rate = errs[i];
        if (rc == SQ_OK) rc = rc2;
    }
    sq_batch_free(b);
    return rc;
}

SQ_EXPORT uint64_t sq_qcmetrics_number_of_reads(sq_qcmetrics *m) { return m->number_of_reads; }
SQ_EXPORT uint64_t sq_qcmetrics_max_length(sq_qcmetrics *m) { return m->max_length; }
SQ_EXPORT uint64_t sq_qcmetrics_end_anchor_length(sq_qcmetrics *m) { return m->end_anchor; }

static int64_t copy_out(sq_ctx *ctx, const void *d, size_t count, void *out, size_t cap, size_t elem)
{
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) { sq_set_error("stream sync failed"); return SQ_ERR_HIP; }
    if (out && cap >= count && count) {
        if (hipMemcpy(out, d, count * elem, hipMemcpyDeviceToHost) != hipSuccess) {
            sq_set_error("device to host copy failed");
            return SQ_ERR_HIP;
        }
    }
    return (int64_t)count;
}

SQ_EXPORT int64_t sq_qcmetrics_base_count_table(sq_qcmetrics *m, uint64_t *out, size_t cap)
{ return copy_out(m->ctx, m->d_base, m->max_length * 5, out, cap, 8); }
SQ_EXPORT int64_t sq_qcmetrics_phred_count_table(sq_qcmetrics *m, uint64_t *out, size_t cap)
{ return copy_out(m->ctx, m->d_phred, m->max_length * 12, out, cap, 8); }
SQ_EXPORT int64_t sq_qcmetrics_end_anchored_base_count_table(sq_qcmetrics *m, uint64_t *out, size_t cap)
{ return copy_out(m->ctx, m->d_ea_base, m->end_anchor * 5, out, cap, 8); }
SQ_EXPORT int64_t sq_qcmetrics_end_anchored_phred_count_table(sq_qcmetrics *m, uint64_t *out, size_t cap)
{ return copy_out(m->ctx, m->d_ea_phred, m->end_anchor * 12, out, cap, 8); }
SQ_EXPORT int64_t sq_qcmetrics_gc_content(sq_qcmetrics *m, uint64_t *out, size_t cap)
{ return copy_out(m->ctx, m->d_gc, 101, out, cap, 8); }
SQ_EXPORT int64_t sq_qcmetrics_phred_scores(sq_qcmetrics *m, uint64_t *out, size_t cap)
{ return copy_out(m->ctx, m->d_ps, 94, out, cap, 8); }

SQ_EXPORT int64_t sq_qcmetrics_device_tables(sq_qcmetrics *m, void **ptrs, uint64_t *counts, size_t cap)
{
    void *p[6] = {m->d_base, m->d_phred, m->d_ea_base, m->d_ea_phred, m->d_gc, m->d_ps};
    uint64_t c[6] = {m->cap, m->cap_phred, m->end_anchor * 5, m->end_anchor * 12, 101, 94};
    for (size_t i = 0; i < 6 && i < cap; i++) { ptrs[i] = p[i]; counts[i] = c[i]; }
    return 6;
}

SQ_EXPORT int sq_qcmetrics_set_totals(sq_qcmetrics *m, uint64_t number_of_reads, uint64_t max_length)
{
    int rc = sq_qcmetrics_reserve(m, max_length);
    if (rc) return rc;
    m->number_of_reads = number_of_reads;
    m->max_length = max_length;
    return SQ_OK;
}

/* ---- the job's exchange step without torch: RCCL through the C ABI (sq_dist.hip) ----------------------
 * What sequali_amd/dist.py::merge_qcmetrics / merge_adaptercounter do through torch.distributed: the ranks agree on
 * the longest read (all-reduce max) and on the row length, pad their tables to it, sum them in place, and set the
 * scalars the host keeps.  Afterwards every rank holds the job's tables.  SURVEY 8e; _qcmodule.c:1870-1906 (resize). */
extern "C" int sq_rccl_allreduce_tables(sq_ctx *ctx, void *comm, void *const *ptrs, const uint64_t *counts, size_t n, int op);

namespace {
/* {a, b} -> {max over ranks of a, sum over ranks of b} */
int rccl_max_and_sum(sq_ctx *ctx, void *comm, uint64_t *a_max, uint64_t *b_sum)
{
    unsigned long long *d = (unsigned long long *)sq_scratch(ctx, 27, 16);
    if (!d) { sq_set_error("out of device memory"); return SQ_ERR_MEMORY; }
    ctx->pinned[12] = *a_max;
    ctx->pinned[13] = *b_sum;
    SQ_HIP(hipMemcpyAsync(d, &ctx->pinned[12], 16, hipMemcpyHostToDevice, ctx->stream));
    void *p0 = d, *p1 = d + 1;
    const uint64_t one = 1;
    int rc = sq_rccl_allreduce_tables(ctx, comm, &p0, &one, 1, 2);
    if (rc) return rc;
    rc = sq_rccl_allreduce_tables(ctx, comm, &p1, &one, 1, 0);
    if (rc) return rc;
    SQ_HIP(hipMemcpyAsync(&ctx->pinned[12], d, 16, hipMemcpyDeviceToHost, ctx->stream));
    SQ_HIP(hipStreamSynchronize(ctx->stream));
    *a_max = ctx->pinned[12];
    *b_sum = ctx->pinned[13];
    return SQ_OK;
}
}  // namespace

SQ_EXPORT int sq_qcmetrics_allreduce(sq_qcmetrics *m, void *comm)
{
    sq_ctx *ctx = m->ctx;
    uint64_t ml = m->max_length, reads = m->number_of_reads;
    int rc = rccl_max_and_sum(ctx, comm, &ml, &reads);
    if (rc) return rc;
    rc = sq_qcmetrics_reserve(m, ml);   /* every rank: the job's rows, zero where this shard saw nothing */
    if (rc) return rc;
    void *ptrs[6] = {m->d_base, m->d_phred, m->d_ea_base, m->d_ea_phred, m->d_gc, m->d_ps};
    const uint64_t counts[6] = {ml * 5, ml * 12, m->end_anchor * 5, m->end_anchor * 12, 101, 94};
    rc = sq_rccl_allreduce_tables(ctx, comm, ptrs, counts, 6, 0);
    if (rc) return rc;
    SQ_HIP(hipStreamSynchronize(ctx->stream));
    m->number_of_reads = reads;
    m->max_length = ml;
    return SQ_OK;
}

SQ_EXPORT int sq_adaptercounter_allreduce(sq_adaptercounter *a, void *comm)
{
    sq_ctx *ctx = a->ctx;
    uint64_t ml = a->max_length, seqs = a->number_of_sequences;
    int rc = rccl_max_and_sum(ctx, comm, &ml, &seqs);
    if (rc) return rc;
    rc = sq_adaptercounter_reserve(a, ml);
    if (rc) return rc;
    /* rows grow geometrically: ranks with different batch histories hold different row lengths */
    uint64_t row = a->cap, unused = 0;
    rc = rccl_max_and_sum(ctx, comm, &row, &unused);
    if (rc) return rc;
    rc = sq_adaptercounter_set_row_length(a, row);
    if (rc) return rc;
    void *ptrs[2] = {a->d_fwd, a->d_rev};
    const uint64_t counts[2] = {a->adapters.size() * a->cap, a->adapters.size() * a->cap};
    rc = sq_rccl_allreduce_tables(ctx, comm, ptrs, counts, 2, 0);
    if (rc) return rc;
    SQ_HIP(hipStreamSynchronize(ctx->stream));
    a->number_of_sequences = seqs;
    a->max_length = ml;
    return SQ_OK;
}

/* ---- AdapterCounter: add, getters -------------------------------------------------------- */

SQ_EXPORT int sq_adaptercounter_add(sq_adaptercounter *a, const uint8_t *buf, size_t buf_len,
                                    const sq_meta *metas, size_t n)
{
    sq_batch *b = sq_batch_upload(a->ctx, buf, buf_len, metas, n);
    if (!b) return SQ_ERR_MEMORY;
    int rc = sq_adaptercounter_add_batch(a, b);
    sq_batch_free(b);
    return rc;
}

SQ_EXPORT int sq_adaptercounter_flush(sq_adaptercounter *a) { return sq_synchronize(a->ctx); }
SQ_EXPORT uint64_t sq_adaptercounter_number_of_sequences(sq_adaptercounter *a) { return a->number_of_sequences; }
SQ_EXPORT uint64_t sq_adaptercounter_max_length(sq_adaptercounter *a) { return a->max_length; }
SQ_EXPORT uint64_t sq_adaptercounter_number_of_adapters(sq_adaptercounter *a) { return a->adapters.size(); }

SQ_EXPORT int64_t sq_adaptercounter_get_counts(sq_adaptercounter *a, size_t i, uint64_t *forward,
                                               uint64_t *reverse, size_t cap)
{
    if (i >= a->adapters.size()) { sq_set_error("adapter index out of range"); return SQ_ERR_VALUE; }
    int64_t r = copy_out(a->ctx, a->d_fwd + i * a->cap, a->max_length, forward, cap, 8);
    if (r < 0) return r;
    return copy_out(a->ctx, a->d_rev + i * a->cap, a->max_length, reverse, cap, 8);
}

SQ_EXPORT int64_t sq_adaptercounter_device_tables(sq_adaptercounter *a, void **ptrs, uint64_t *counts, size_t cap)
{
    if (cap > 0) { ptrs[0] = a->d_fwd; counts[0] = a->adapters.size() * a->cap; }
    if (cap > 1) { ptrs[1] = a->d_rev; counts[1] = a->adapters.size() * a->cap; }
    return 2;
}

SQ_EXPORT int sq_adaptercounter_set_totals(sq_adaptercounter *a, uint64_t number_of_sequences, uint64_t max_length)
{
    int rc = sq_adaptercounter_reserve(a, max_length);
    if (rc) return rc;
    a->number_of_sequences = number_of_sequences;
    a->max_length = max_length;
    return SQ_OK;
}

/* ---- PerTileQuality: add, getters ---------------------------------------------------------- */

SQ_EXPORT int sq_pertile_add(sq_pertile *p, const uint8_t *buf, size_t buf_len, const sq_meta *metas, size_t n)
{
    if (p->skipped) { p->records_seen += n; return SQ_OK; }
    sq_batch *b = sq_batch_upload(p->ctx, buf, buf_len, metas, n);
    if (!b) return SQ_ERR_MEMORY;
    int rc = sq_pertile_add_batch(p, b);
    sq_batch_free(b);
    return rc;
}

SQ_EXPORT int sq_pertile_flush(sq_pertile *p) { return sq_synchronize(p->ctx); }
SQ_EXPORT uint64_t sq_pertile_number_of_reads(sq_pertile *p) { return p->number_of_reads; }
SQ_EXPORT uint64_t sq_pertile_max_length(sq_pertile *p) { return p->max_length; }
SQ_EXPORT const char *sq_pertile_skipped_reason(sq_pertile *p) { return p->skipped ? p->skipped_reason.c_str() : nullptr; }
SQ_EXPORT uint64_t sq_pertile_number_of_tiles(sq_pertile *p) { return (uint64_t)p->n_slots; }

SQ_EXPORT int64_t sq_pertile_get_tile_counts(sq_pertile *p, int64_t *tile_ids, double *errors,
                                             uint64_t *counts, size_t cap_tiles, size_t cap_len)
{
    sq_ctx *ctx = p->ctx;
    SQ_HIP(hipStreamSynchronize(ctx->stream));
    const size_t nt = (size_t)p->n_slots, ml = (size_t)p->max_length;
    if (!tile_ids || cap_tiles < nt || cap_len < ml || nt == 0) return (int64_t)nt;
    std::vector<long long> keys(TILE_MAP_SIZE);
    std::vector<int> vals(TILE_MAP_SIZE);
    SQ_HIP(hipMemcpy(keys.data(), p->map.keys, TILE_MAP_SIZE * 8, hipMemcpyDeviceToHost));
    SQ_HIP(hipMemcpy(vals.data(), p->map.vals, TILE_MAP_SIZE * 4, hipMemcpyDeviceToHost));
    std::vector<std::pair<long long, int>> order;
    for (size_t i = 0; i < TILE_MAP_SIZE; i++)
        if (keys[i] != TILE_EMPTY && vals[i] >= 0) order.emplace_back(keys[i], vals[i]);
    std::sort(order.begin(), order.end()); /* :3317 ascending tile id */
    std::vector<unsigned long long> lc(p->slot_cap * p->len_cap);
    std::vector<double> er(p->slot_cap * p->len_cap);
    if (!lc.empty()) {
        SQ_HIP(hipMemcpy(lc.data(), p->d_len_counts, lc.size() * 8, hipMemcpyDeviceToHost));
        SQ_HIP(hipMemcpy(er.data(), p->d_errors, er.size() * 8, hipMemcpyDeviceToHost));
    }
    for (size_t k = 0; k < order.size() && k < nt; k++) {
        const size_t slot = (size_t)order[k].second;
        tile_ids[k] = order[k].first;
        uint64_t running = 0; /* :3336-3347 reads reaching position j */
        for (size_t j = ml; j-- > 0;) {
            running += lc[slot * p->len_cap + j];
            errors[k * cap_len + j] = er[slot * p->len_cap + j];
            counts[k * cap_len + j] = running;
        }
    }
    return (int64_t)nt;
}

/* ---- PerTileQuality across shards ------------------------------------------ */
/* index, among the records this object was given, of the first record whose header has no
 * tile id (the module stopped counting there, :3137-3148); -1 while active */
SQ_EXPORT int64_t sq_pertile_first_unparsable(sq_pertile *p)
{
    return p->first_bad == UINT64_MAX ? -1 : (int64_t)p->first_bad;
}

/* replaces the state by merged tables: tile k owns row k of errors / length_counts
 * ([n_tiles][len]; length_counts[k][j] = reads of tile k that are j + 1 long, the raw form
 * of what get_tile_counts reverse-cumulates); skipped_reason NULL keeps the module active */
SQ_EXPORT int sq_pertile_install(sq_pertile *p, const int64_t *tile_ids, size_t n_tiles, const double *errors,
                                 const uint64_t *length_counts, size_t len, uint64_t number_of_reads,
                                 const char *skipped_reason)
{
    sq_ctx *ctx = p->ctx;
    if (n_tiles > TILE_MAP_SIZE / 2) {
        sq_set_error("PerTileQuality: more than %u distinct tile ids", TILE_MAP_SIZE / 2);
        return SQ_ERR_MEMORY;
    }
    SQ_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<long long> keys(TILE_MAP_SIZE, TILE_EMPTY);
    std::vector<int> vals(TILE_MAP_SIZE, -1);
    for (size_t k = 0; k < n_tiles; k++) {
        uint32_t idx = (uint32_t)(((unsigned long long)tile_ids[k] * 0x9E3779B97F4A7C15ULL) >> 48) & (TILE_MAP_SIZE - 1);
        while (keys[idx] != TILE_EMPTY) idx = (idx + 1) & (TILE_MAP_SIZE - 1);
        keys[idx] = tile_ids[k];
        vals[idx] = (int)k;
    }
    const int n_slots = (int)n_tiles;
    SQ_HIP(hipMemcpy(p->map.keys, keys.data(), TILE_MAP_SIZE * 8, hipMemcpyHostToDevice));
    SQ_HIP(hipMemcpy(p->map.vals, vals.data(), TILE_MAP_SIZE * 4, hipMemcpyHostToDevice));
    SQ_HIP(hipMemcpy(p->map.n_slots, &n_slots, 4, hipMemcpyHostToDevice));
    if (p->d_len_counts) { SQ_HIP(hipFree(p->d_len_counts)); p->d_len_counts = nullptr; }
    if (p->d_errors) { SQ_HIP(hipFree(p->d_errors)); p->d_errors = nullptr; }
    p->slot_cap = n_tiles;
    p->len_cap = len;
    if (n_tiles && len) {
        SQ_HIP(hipMalloc((void **)&p->d_len_counts, n_tiles * len * 8));
        SQ_HIP(hipMalloc((void **)&p->d_errors, n_tiles * len * 8));
        SQ_HIP(hipMemcpy(p->d_len_counts, length_counts, n_tiles * len * 8, hipMemcpyHostToDevice));
        SQ_HIP(hipMemcpy(p->d_errors, errors, n_tiles * len * 8, hipMemcpyHostToDevice));
    } else {
        p->slot_cap = 0;
        p->len_cap = 0;
    }
    p->n_slots = n_slots;
    p->max_length = len;
    p->number_of_reads = number_of_reads;
    p->records_seen = number_of_reads;
    p->skipped = skipped_reason != nullptr;
    p->skipped_reason = skipped_reason ? skipped_reason : "";
    p->first_bad = p->skipped ? number_of_reads : UINT64_MAX;
    const unsigned long long fb = p->first_bad;
    SQ_HIP(hipMemcpy(p->d_first_bad, &fb, 8, hipMemcpyHostToDevice));
    return SQ_OK;
}
