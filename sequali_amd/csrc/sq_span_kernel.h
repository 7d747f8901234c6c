/*
 * sq_span_kernel.h -- k_span itself (see sq_span.hip for what it does and why): the kernel template and the device
 * helpers it is written with, in a header so that its builds can be spread over translation units (sq_span.hip: the
 * builds of QCMetrics / AdapterCounter alone; sq_pair.hip: the builds that carry PerTileQuality and the overlap scan
 * of InsertSizeMetrics along).  Everything has internal linkage.
 */
#ifndef SQ_SPAN_KERNEL_H
#define SQ_SPAN_KERNEL_H

#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include <algorithm>
#include <type_traits>

#include "sq_pass.h"
#include "sq_span.h"

namespace {

constexpr uint32_t SPAN_R = 16;              /* records per span */
constexpr uint32_t CLS6_PAD4 = 0x1E1E1E1Eu;  /* code 30 */
constexpr int SPAN_W4 = 3;  /* dwords an automaton is restarted in front of its piece: adapters of up to 13 characters */
#define SPAN_STATES(P) ((P).dfa2_states)   /* k_span walks the two-character automaton (build_pair_dfa) */
#ifndef SPAN_S
#define SPAN_S 1   /* pieces a lane cuts its quarter into, one automaton each (more pieces: shorter chains, more table reads; the reads cost more) */
#endif

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

/* The same four class codes for the bytes sequencers write -- A C G T N in either case, and the bytes 0x00 / 0x20 --, one
 * v_perm instead of ten instructions; cls6_unusual says whether a dword holds anything else (an IUPAC letter, a dot, a byte
 * >= 0x80): then cls6_of_dword is the answer.  For a byte b with idx = b & 7: A C G T sit at 1 3 7 4 and N at 6; `want` is
 * the upper-case letter idx stands for (0 for the others, which the bytes 0x00 and 0x20 "match": class N either way). */
__device__ __forceinline__ uint32_t cls6_usual(uint32_t w)
{
    return __builtin_amdgcn_perm(0x0C181812u, 0x06180018u, w & 0x07070707u);
}
__device__ __forceinline__ uint32_t cls6_unusual(uint32_t w)   /* != 0: some byte of w is not of the usual ones */
{
    const uint32_t want = __builtin_amdgcn_perm(0x474E0054u, 0x43004100u, w & 0x07070707u);
    return (w ^ want) & 0xDFDFDFDFu;
}
/* what QCMetrics counts per read besides the classes (:1997-2060): 0x01 for C / G, 0x10 for a base that is none of A C G T,
 * per byte, from the byte's idx (usual bytes) or from the selector cls6_of_dword ends on (any byte) */
__device__ __forceinline__ uint32_t gcn_usual(uint32_t w)
{
    return __builtin_amdgcn_perm(0x01101000u, 0x01100010u, w & 0x07070707u);
}
__device__ __forceinline__ uint32_t cls6_gcn_of_dword(uint32_t w, uint32_t *gcn)
{
    const uint32_t idx = w & 0x07070707u;
    const uint32_t lut = __builtin_amdgcn_perm(0x0C181812u, 0x06180018u, idx);
    const uint32_t flg = __builtin_amdgcn_perm(0x01101000u, 0x01100010u, idx);
    const uint32_t want = __builtin_amdgcn_perm(0x47000054u, 0x43004100u, idx);
    const uint32_t d = (w & 0xDFDFDFDFu) ^ want;
    const uint32_t ne4 = ((d + 0x7F7F7F7Fu) & 0x80808080u) >> 5;
    *gcn = __builtin_amdgcn_perm(flg, 0x10101010u, 0x07060504u - ne4);
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
/* byte offset of automaton state n from the root, and back (k_span): three states share
   SPAN_DFA_ROW bytes, their entries interleaved (span_lds_layout knows the same number) */
__host__ __device__ constexpr uint32_t span_dfa_offset(uint32_t n) { return SPAN_DFA_ROW * (n / 3) + 2 * (n % 3); }
__device__ __forceinline__ uint32_t span_dfa_state(uint32_t off) { return 3 * (off / SPAN_DFA_ROW) + (off % SPAN_DFA_ROW) / 2; }
/* a + byte J of w */
template <int J> __device__ __forceinline__ uint32_t add_byte(uint32_t a, uint32_t w)
{
    uint32_t r;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_%3"
        : "=v"(r) : "v"(a), "v"(w), "i"(J));
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

/* LDS reads the compiler neither counts nor waits for: the rounds of phase S keep a dozen of them
 * in flight behind the automaton's dependent reads and wait by hand (lgkmcnt, LDS answers in
 * order; the counter has four bits).  A value is used only behind a wait_* that names it. */
template <int OFF> __device__ __forceinline__ uint32_t rd_u8(uint32_t a)
{
    uint32_t r;
    asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(r) : "v"(a), "i"(OFF) : "memory");
    return r;
}
template <int OFF> __device__ __forceinline__ uint32_t rd_u16(uint32_t a)
{
    uint32_t r;
    asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(r) : "v"(a), "i"(OFF) : "memory");
    return r;
}
template <int OFF> __device__ __forceinline__ uint32_t rd_b32(uint32_t a)
{
    uint32_t r;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r) : "v"(a), "i"(OFF) : "memory");
    return r;
}
typedef uint32_t sq_u32x4v __attribute__((ext_vector_type(4)));
template <int OFF> __device__ __forceinline__ sq_u32x4v rd_b128(uint32_t a)   /* a + OFF: 16-byte aligned */
{
    sq_u32x4v r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(a), "i"(OFF) : "memory");
    return r;
}
__device__ __forceinline__ double rd_f64(uint32_t a)
{
    double r;
    asm volatile("ds_read_b64 %0, %1" : "=v"(r) : "v"(a) : "memory");
    return r;
}
/* ds_read_b64_tr_b8: in a group of 16 lanes, lane 2 q + p hands in the address of 8 bytes
 * (8-byte aligned): chunk p of row q; lane i < 8 gets byte i of chunk 0 of the 8 rows (row q in
 * byte q of the 64 bits), lane 8 + i byte i of chunk 1 (scripts/ubench_tr8.hip,
 * profiles/r2b/ubench_tr8.txt).  All 64 lanes must be active. */
typedef uint32_t sq_u32x2 __attribute__((ext_vector_type(2)));
template <int OFF> __device__ __forceinline__ sq_u32x2 rd_tr8(uint32_t a)
{
    sq_u32x2 r;
    asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(r) : "v"(a), "i"(OFF) : "memory");
    return r;
}
/* c + (1 << byte J of w) for bytes below 32 (the class codes: 0 .. 30).  A shift takes its amount from the low five bits of its
   operand, so byte 0 needs no extraction and the others one v_lshrrev; v_lshl_add_u32 does the rest.  (Until round 5 this was
   v_lshlrev_b32_sdwa + v_add_u32: gfx950 wants a wait state between an SDWA instruction and whatever uses its result, hipcc put
   an s_nop behind every pair, and an s_nop costs a wave as much as an instruction: 120 issue slots per span for 40 cells, now 70.) */
template <int J> __device__ __forceinline__ uint32_t add_one_shl_byte(uint32_t w, uint32_t one, uint32_t c)
{
    /* plain C, not asm: behind an asm statement whose result the next instruction uses hipcc puts an s_nop (it cannot know
       what the statement holds; gfx950 wants a wait state behind SDWA and transcendental results), and these statements are
       chains on one counter.  The empty statement keeps the order: left to itself hipcc collects the shifted ones of a whole
       round in registers */
    c = (one << ((J ? w >> (8 * J) : w) & 31u)) + c;
    asm volatile("" :: "v"(c));   /* (an INPUT only: a statement that defines c would get an s_nop in front of c's next use) */
    return c;
}
/* byte J of w, times 2 */
template <int J> __device__ __forceinline__ uint32_t shl1_byte(uint32_t w, uint32_t one)
{
    uint32_t t;
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_%3"
        : "=v"(t) : "v"(one), "v"(w), "i"(J));
    return t;
}
template <int OFF> __device__ __forceinline__ void inc_u32(uint32_t a, uint32_t one)
{
    asm volatile("ds_add_u32 %0, %1 offset:%2" :: "v"(a), "v"(one), "i"(OFF) : "memory");
}

__device__ __forceinline__ void add_f64_lds(uint32_t a, double v)
{
    asm volatile("ds_add_f64 %0, %1" :: "v"(a), "v"(v) : "memory");
}
/* 16-bit half J of w, times 8 */
template <int J> __device__ __forceinline__ uint32_t shl3_word_of(uint32_t w, uint32_t three)
{
    uint32_t t;
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_%3"
        : "=v"(t) : "v"(three), "v"(w), "i"(J));
    return t;
}
/* byte J of w, times 8 */
template <int J> __device__ __forceinline__ uint32_t shl3_byte_of(uint32_t w, uint32_t three)
{
    uint32_t t;
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_%3"
        : "=v"(t) : "v"(three), "v"(w), "i"(J));
    return t;
}

__device__ __forceinline__ void tie(uint32_t &x) { asm volatile("" : "+v"(x)); }   /* x is used behind this point only */
__device__ __forceinline__ void tie_f64(double &x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void tie2(sq_u32x2 &x) { asm volatile("" : "+v"(x)); }
template <int N> __device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" :: "i"(N) : "memory"); }
template <int I, int N, class F> __device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
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

/* waves per workgroup.  One wave for both streams of a span: LDS leaves at most 12 of them from 128
 * positions per read on, which is 168 registers per lane instead of 128.  SPLIT (a wave holds one
 * stream of a span at a time): 16 waves of 128 registers up to 160 positions (128 when the batch
 * holds many lengths), 12 beyond */
constexpr int span_max_waves(int nw, bool split = false, bool seg = false, bool lng = false, bool pt = false)
{
    if (pt && nw >= 6) return 8;   /* PerTileQuality's sums per position ride in registers: beyond 160 positions the 12-wave build spills */
    /* 225-256 positions (and 193-224 of the builds for sorted rows): 8 waves of up to 256 registers -- at 12
       waves these builds spilled, and a build that spills is not used (span_waves): such reads went to the round-1
       kernels without a word (tests/test_gpu_routes.py) */
    /* (round 5: the builds' registers fell by a third when the chain's tail stopped reading bytes one by one -- the 8-window build
       of a wave per stream from 195 to 98 --, so batches of one read length with a wave per stream are bounded by LDS now: 16
       waves up to 7 windows (LDS takes 7 windows down to 14), 12 at 8) */
    if (!lng && split && !seg) return nw >= 8 ? 12 : 16;
    if (!lng && (nw >= 8 || (nw == 7 && seg))) return 8;
    return nw <= (lng ? 6 : split ? 5 : 3) ? 16 : 12;   /* (sorted rows, a wave per stream: 16 waves up to 5 windows since the registers allow it, round 5) */
}

#if defined(SQ_SPAN_MARK)   /* comments in the ISA listing (hipcc -S): instructions per phase can be counted */
#define SPAN_PHASE(k) do { asm volatile("; SPAN_PHASE " #k ::: "memory"); } while (0)
#else
#define SPAN_PHASE(k) do { } while (0)
#endif
/* SEG: the batch holds reads of many lengths; P.span_rows lists them sorted by length, 16 reads of
   one length per span, P.span_segs the lengths in span order.  A workgroup takes a contiguous
   stretch of the spans (a handful of lengths at most) and merges its histograms whenever the
   length changes, so that inside a stretch everything is as for a batch of one read length: the
   end-anchored tables a window of the positional ones, no question asked per row.

   SPLIT: what is counted of a read's bases (class codes, base counts, GC, the automaton) and what
   of its qualities (the f64 chains, the phred histogram, the bins of the average) have nothing
   to do with each other, so a wave takes ONE stream of a span at a time: half the LDS per slot
   and half the live registers, 16 waves per CU instead of 12.  The waves of a workgroup work in
   pairs on one sequence of spans: wave 2 j takes the bases of the pair's spans 0, 2, 4 .. and the
   qualities of 1, 3, 5 .., wave 2 j + 1 the other halves -- every wave alternates between the two
   roles, so the two kinds of work need no balancing. */
/* LONG (with SEG and SPLIT): the rows are SEGMENTS of long reads -- positions [pos_base, pos_base +
   32 NW) of the reads that are longer than pos_base, pos_base a per-stretch scalar like U -- for what
   can be counted segment by segment: the positional histograms and the automaton (restarted in
   front of the segment like in front of a lane's quarter, the 12 bases in front of the segment
   come along in a piece of their own; a match is a candidate for the read's first occurrence,
   P.long_first).  The rows of a span may end inside the segment (the read's last one): what lies
   behind a row's end is turned into padding in LDS.  Nothing per read is done here (the f64
   chains, the bins, GC: k_read_sums) nor the end-anchored tables (k_long_ea). */
/* round(gc * 100.0 / acgt) of _qcmodule.c:2058 (C round: halves away from zero) for 0 <= gc <= acgt <= 4096 in
 * integers: the quotient is a multiple of 1 / (2 acgt) away from every half unless it IS one, far more than the
 * rounding of the f64 division moves it, so floor((200 gc + acgt) / (2 acgt)) is the same number.  The
 * division by a float reciprocal, corrected by the remainder. */
__device__ __forceinline__ uint32_t gc_percent(uint32_t gc, uint32_t acgt)
{
    const uint32_t n = 200u * gc + acgt, d = 2u * acgt;
    uint32_t q = (uint32_t)((float)n * __builtin_amdgcn_rcpf((float)d));
    int32_t r = (int32_t)(n - q * d);
    if (r < 0) { q--; r += (int32_t)d; }
    if (r >= (int32_t)d) q++;
    return q;
}

/* PT: PerTileQuality_add_meta (:3123-3222) rides along (PassParams::pt_*): lanes c == 0 parse their read's tile id from
   the first 64 bytes of its header (illumina_header_to_tile_id :3088-3121; four plain 16-byte loads per read that hit
   the lines the DMA has just fetched), lane (h, pl) looks up the error rates of the qualities phase H hands it anyway
   and keeps their sums per position in registers while the wave's reads stay in one tile.  A workgroup takes a
   CONTIGUOUS stretch of the batch then, so that a wave meets a tile change once, not once per workgroup. */
/* PAIR (with PT): the overlap scan of InsertSizeMetrics (calculate_insert_size :5667-5707) rides along too, split over
   the passes of the two mates: read 2's pass (PAIR = 1) leaves the first and the last 16 bases of every read in
   P.pair_ends; read 1's pass (PAIR = 2) brings a span's 512 bytes of them in by one LDS-DMA and scans the sequences it
   holds in LDS anyway, four lanes per pair as k_isz_span does (sq_span.hip), before the class pass turns them into
   codes.  The insert size of every pair goes to P.pair_results; the histogram and the adapter remainders are
   k_isz_adapters' work. */
__device__ __forceinline__ uint32_t pair_nonzero_bytes_of(uint32_t v)
{
    return (uint32_t)__popc((((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v) & 0x80808080u);
}
/* NUCLEOTIDE_COMPLEMENT, _qcmodule.c:5613-5631: reverse complement of 8 bases (0 for what is no base).  Four bases per
   v_perm, as cls6_of_dword looks its classes up: the low three bits of a letter name the complement and the letter itself
   (A 1, C 3, G 7, T 4); what is not that letter (either case) gives 0; the last v_perm turns the four bytes round.  (Byte by
   byte with compares this took 112 instructions per lane and span, a seventh of the scan.) */
__device__ __forceinline__ uint32_t pair_revcomp4(uint32_t w)
{
    const uint32_t idx = w & 0x07070707u;
    const uint32_t comp = __builtin_amdgcn_perm(0x43000041u, 0x47005400u, idx);
    const uint32_t want = __builtin_amdgcn_perm(0x47000054u, 0x43004100u, idx);
    const uint32_t d = (w & 0xDFDFDFDFu) ^ want;
    const uint32_t ne4 = ((((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d) & 0x80808080u) >> 5;
    const uint32_t c = __builtin_amdgcn_perm(comp, 0u, 0x07060504u - ne4);
    return __builtin_amdgcn_perm(0u, c, 0x00010203u);
}
__device__ __forceinline__ unsigned long long pair_revcomp8(unsigned long long a)
{
    return ((unsigned long long)pair_revcomp4((uint32_t)a) << 32) | pair_revcomp4((uint32_t)(a >> 32));
}

/* The prefilter of the overlap scan (calculate_insert_size :5667-5707) for one lane of a pair's quad: the WQ = 8 NW positions
   that start at LDS address `ra` (8-byte aligned; the lane's WQ + 16 bytes must be readable).  Is the 8-base word at i a
   needle's first half, or the one at i + 8 its second half (:5695-5698: at most one byte of the 16 may differ, so one of the
   halves is exact)?  Whole halves, not their low dwords as in round 4 (a dword matches somewhere in every other lane by
   chance, and every chance match sent the wave through the byte-by-byte look that follows), and a half as ONE word,
   f(i) = dword(i) + 8 dword(i + 4) (not injective: what it lets through, the look behind it turns away).  No compares:
   v_sad_u8 of two words is 0 exactly when they are equal (a compare into a scalar pair costs 2.3 plain instructions here,
   scripts/ubench_qsad.hip); the four sums of a position meet in one v_min and one v_min3, `- 1` turns "zero" into the sign
   bit and v_alignbit shifts it into the mask: 9 instructions per position where the sliding compare of rounds 3-4 took 20
   (read 1's pass of config 3: 6.14 -> 4.77 ms per 25 M pairs).  The bytes wait in registers as dwords (upper case: & 0xDF);
   the dword at position 4 m + j is v_alignbyte(d[m + 1], d[m], j).  Returns the candidates: bit 63 - k = position k.
   hl / hl2: low / high dword of the first half of the head needle, hh / hh2: of its second half; tl .. th2: the tail needle. */
template <int NW>
__device__ __forceinline__ unsigned long long pair_scan_candidates(uint32_t ra, uint32_t hl, uint32_t hl2, uint32_t hh, uint32_t hh2,
                                                                   uint32_t tl, uint32_t tl2, uint32_t th, uint32_t th2)
{
    constexpr int WQ = 8 * NW, NDW = 2 * NW + 4;
    const uint32_t UP4 = 0xDFDFDFDFu;
    uint32_t dm[NDW];
#pragma unroll
    for (int k = 0; k < NDW / 2; k++) {
        const unsigned long long v = *(SQ_LDS const unsigned long long *)(uintptr_t)(ra + 8 * k);
        dm[2 * k] = (uint32_t)v & UP4;
        dm[2 * k + 1] = (uint32_t)(v >> 32) & UP4;
    }
    const uint32_t f_hl = hl + (hl2 << 3), f_hh = hh + (hh2 << 3), f_tl = tl + (tl2 << 3), f_th = th + (th2 << 3);
    uint32_t fold[WQ + 8];
    {
        uint32_t at[WQ + 12];   /* the dword at every position the lane looks at */
        static_for<0, WQ + 12>([&](auto pc) {
            constexpr int pp = decltype(pc)::value, m = pp / 4, j = pp % 4;
            at[pp] = j ? __builtin_amdgcn_alignbyte(dm[m + 1], dm[m], j) : dm[m];
        });
        static_for<0, WQ + 8>([&](auto pc) { constexpr int pp = decltype(pc)::value; fold[pp] = at[pp] + (at[pp + 4] << 3); });
    }
    uint32_t acc_lo = 0, acc_hi = 0;
    static_for<0, WQ>([&](auto pc) {
        constexpr int pp = decltype(pc)::value;
        const uint32_t f = fold[pp], f8 = fold[pp + 8];
        const uint32_t second = min(__builtin_amdgcn_sad_u8(f8, f_hh, 0u), __builtin_amdgcn_sad_u8(f8, f_th, 0u));
        const uint32_t any = min(min(__builtin_amdgcn_sad_u8(f, f_hl, 0u), __builtin_amdgcn_sad_u8(f, f_tl, 0u)), second);
        if constexpr (pp < 32) acc_lo = __builtin_amdgcn_alignbit(acc_lo, any - 1u, 31);
        else acc_hi = __builtin_amdgcn_alignbit(acc_hi, any - 1u, 31);
    });
    constexpr int N_LO = WQ < 32 ? WQ : 32, N_HI = WQ - N_LO;
    unsigned long long cand = (unsigned long long)acc_lo << (64 - N_LO);
    if constexpr (N_HI > 0) cand |= (unsigned long long)acc_hi << (32 - N_HI);
    return cand;
}

template <int NW, bool AD, bool SEG = false, int W4T = SPAN_W4, bool SPLIT = false, bool LONG = false, bool PT = false, int PAIR = 0>
__global__ void __launch_bounds__(64 * span_max_waves(NW, SPLIT, SEG, LONG, PT)) k_span(PassParams P, uint32_t n_ad)
{
    static_assert(PAIR == 0 || PT, "the passes over pairs are builds of the pass that carries PerTileQuality");
    static_assert(!LONG || (SEG && SPLIT), "segments of long reads come as sorted rows, a wave per stream");
    static_assert(!PT || (!AD && !SEG && !SPLIT && !LONG), "PerTileQuality rides with QCMetrics alone on batches of one read length, one wave for both streams");
    /* Row r of a slot: sequence at r * ROWB, qualities at r * ROWB + QOFF (SPLIT: the slot holds
       one of the two streams, at r * ROWB).  A row is an odd number of 16-byte pieces (the last one
       is never loaded) and a lane's quarter an odd number of dwords, so that the 32 lanes of an LDS
       instruction of phase S (8 rows x 4 quarters) fall into 32 different banks */
    constexpr uint32_t PRE = LONG ? 16 : 0;   /* LONG: the row's piece 0 holds the 16 bytes in front of the segment */
    constexpr uint32_t SB = 32 * NW, PR = (SPLIT ? 2 : 4) * NW + 1 + (LONG ? 2 : 0), ROWB = 16 * PR, SLOT = SPAN_R * ROWB;
    constexpr uint32_t QOFF = SPLIT ? 0 : SB;
    constexpr uint32_t DW = 8 * NW, Q4 = 2 * NW + 1, ND = (SPAN_R * PR + 63) / 64;
    extern __shared__ __align__(16) uint8_t smem[];
    uint32_t U = SEG ? 32 * NW : P.uniform_len;   /* SEG: the length of the stretch being counted */
    const uint32_t hs = hist_stride(U);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, T = blockDim.x, W = T >> 6;
    const SpanLds L = span_lds_layout(NW, U, AD ? SPAN_STATES(P) : 0, AD ? n_ad : 0, AD ? P.ad_lds : 0, W, SEG, SPLIT, LONG, PAIR == 2);
    double *l_err = (double *)smem;                        /* [SPAN_ERR_N] by raw quality byte; [SPAN_ERR_PAD]: +0.0 */
    uint16_t *l_bin = (uint16_t *)(smem + SPAN_BIN_OFF);   /* [256] byte offset of a quality byte's row in the phred histogram */
    double *l_thr = (double *)(smem + L.thr);              /* [96] */
    uint32_t *l_gc = (uint32_t *)(smem + L.gc);            /* [104] */
    uint32_t *l_ps = (uint32_t *)(smem + L.ps);            /* [96] */
    /* The automaton: the entry of state n for the class with shift code k (0, 6, .. 30) is the
       16-bit LDS address of the next state and lives at address(n) + k, so a step is one SDWA add
       and one ds_read_u16.  Three states share 36 bytes (address(n) = root + 36 (n / 3) +
       2 (n % 3): their entries interleave without a gap), which keeps the dozen shallow states
       nearly every lane sits in (numbered first, build_dfa) in banks of their own: in rows of 32
       bytes, four to the 32 banks, a table read took 5 extra LDS cycles on average
       (SQ_LDS_BANK_CONFLICT, profiles/r2b). */
    /* The automaton takes TWO characters per step (build_pair_dfa, sq_qc.hip): the entry
       of state n for the classes with codes (k1, k2) lives at address(n) + k1 + 6 k2 (multiples of
       6 up to 210: three states share 216 bytes), half as many dependent reads per read; a state
       that reports (>= dfa_hit) names the adapters that end on the step's second character
       (l_out[.][0]) and on its first (l_out[.][1]) */
    uint16_t *l_dfa = (uint16_t *)(smem + L.dfa);
    unsigned long long *l_out = (unsigned long long *)(smem + L.out); /* [states][2] adapters ending on the second / on the first character of the step into that state */
    uint8_t *l_adlen = smem + L.adlen;                     /* [64] */
    uint32_t *l_hist_base = (uint32_t *)(smem + L.hist);   /* [5][hs] */
    constexpr uint32_t PROWS = PHRED_COLS + (SEG ? 1 : 0);   /* SEG: one more row takes the qualities of filler rows */
    uint32_t *l_hist_phred = l_hist_base + hs * BASE_COLS; /* [PROWS][hs] */
    uint32_t *l_adf = l_hist_phred + hs * PROWS;           /* [ad_lds][hs] */
    uint32_t *l_first = (uint32_t *)(smem + L.first) + wave * SPAN_R * (AD ? n_ad : 0); /* [16][n_ad] */
    uint32_t *l_rows = (uint32_t *)(smem + L.rows) + wave * (SEG ? 4 : 2) * SPAN_R;     /* [16][2] (SEG: [16][2] of 64 bits) */
    /* PT: the wave's own addresses as scalars (hipcc does not know that `wave` is uniform and keeps them in vector registers, which those builds lack) */
    const uint32_t slot_base0 = lds_addr(smem + L.slots) + wave * 2 * SLOT;
    const uint32_t slot_base = PT ? (uint32_t)__builtin_amdgcn_readfirstlane(slot_base0) : slot_base0;

    if (lds_addr(l_err) != 0) __builtin_trap(); /* quality byte << 3 is the address of its error rate */
    const uint32_t dfa_root = AD ? lds_addr(l_dfa) : 0, dfa_hit = dfa_root + span_dfa_offset(P.dfa_accept);
    for (int i = tid; i < (int)SPAN_ERR_N; i += T) {   /* every byte that is no phred character: NaN (:2073-2075), also >= 128 (BAM qualities + 33) */
        double e = __longlong_as_double(0x7FF8000000000000LL);
        if (i >= 33 && i <= 33 + SQ_PHRED_MAX) e = __longlong_as_double((long long)c_error_rate_bits[i - 33]);
        else if (i >= 256) e = 0.0;
        l_err[i] = e;
    }
    /* SEG: filler rows hold quality 0x80, which goes to a row of its own (a real 0x80 too: the merge tells them apart) */
    for (int i = tid; i < 256; i += T) l_bin[i] = (uint16_t)((SEG && i == 0x80 ? PHRED_COLS : min((uint32_t)i - 33u, 47u) >> 2) * hs * 4);
    /* thresholds for the SUM of a read's error rates (phred_sum_thresholds(), sq_qc.hip): row U of the table */
    if constexpr (!SEG) for (int i = tid; i < 96; i += T) l_thr[i] = P.thr_sum[U * 96 + i];
    for (int i = tid; i < 104; i += T) l_gc[i] = 0;
    for (int i = tid; i < 96; i += T) l_ps[i] = 0;
    uint32_t *l_prog = (uint32_t *)(smem + L.prog);   /* [16] spans started by each wave */
    if (tid < 16) l_prog[tid] = 0;
    /* PT: the pointers the pass needs now and then wait in LDS (the 64 bytes of l_prog: only the builds of a wave per
       stream use those) instead of in scalar registers across the whole loop -- the builds of 168 registers keep their
       loop invariants in VECTOR registers once the scalar ones are gone, and spill */
    enum { PAR_RESULTS = 0, PAR_ENDS, PAR_SUMS, PAR_RUNS, PAR_NRUNS, PAR_TILES, PAR_BAD, PAR_N,
           /* PAIR = 2: QCMetrics' tables too (used once, at the end: twelve scalar registers less across the loop) */
           PAR_QC_BASE = 8, PAR_QC_PHRED, PAR_QC_EA_BASE, PAR_QC_EA_PHRED, PAR_QC_GC, PAR_QC_PS, PAR_QC_BAD };
    const uint32_t par_base = lds_addr(l_prog), par_more = lds_addr(smem + L.ends) + W * 512;
    auto par = [&](int k) -> uint8_t * {   /* wave-uniform */
        const unsigned long long v = *(volatile SQ_LDS unsigned long long *)(uintptr_t)(k < 8 ? par_base + 8 * k : par_more + 8 * (k - 8));
        return (uint8_t *)(((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) |
                           (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v));
    };
    for (uint32_t i = tid; i < hs * (BASE_COLS + PROWS); i += T) l_hist_base[i] = 0;
    if constexpr (PT) {
        static_assert(!SPLIT, "l_prog holds the parameter table");
        __syncthreads();   /* behind the zeroes above */
        if (tid == 0) {
            unsigned long long *l_par = (unsigned long long *)l_prog;
            l_par[PAR_RESULTS] = (unsigned long long)P.pair_results;
            l_par[PAR_ENDS] = (unsigned long long)P.pair_ends;
            l_par[PAR_SUMS] = (unsigned long long)P.pt_run_sums;
            l_par[PAR_RUNS] = (unsigned long long)P.pt_runs;
            l_par[PAR_NRUNS] = (unsigned long long)P.pt_nruns;
            l_par[PAR_TILES] = (unsigned long long)P.pt_tiles;
            l_par[PAR_BAD] = (unsigned long long)P.pt_bad;
            if constexpr (PAIR == 2) {
                unsigned long long *l_more = (unsigned long long *)(smem + L.ends + W * 512);
                l_more[PAR_QC_BASE - 8] = (unsigned long long)P.qc_base;
                l_more[PAR_QC_PHRED - 8] = (unsigned long long)P.qc_phred;
                l_more[PAR_QC_EA_BASE - 8] = (unsigned long long)P.qc_ea_base;
                l_more[PAR_QC_EA_PHRED - 8] = (unsigned long long)P.qc_ea_phred;
                l_more[PAR_QC_GC - 8] = (unsigned long long)P.qc_gc;
                l_more[PAR_QC_PS - 8] = (unsigned long long)P.qc_ps;
                l_more[PAR_QC_BAD - 8] = (unsigned long long)P.qc_first_bad;
            }
        }
    }
    if (AD) {
        for (uint32_t i = tid; i < P.dfa2_states * 36; i += T) {
            const uint32_t st = i / 36, k = i % 36;   /* k = first class + 6 * second class */
            l_dfa[(span_dfa_offset(st) + 6 * k) >> 1] = (uint16_t)(dfa_root + span_dfa_offset(P.dfa2[i]));
        }
        for (uint32_t i = tid; i < 2 * P.dfa2_states; i += T) l_out[i] = P.dfa2_out[i];
        for (uint32_t i = tid; i < 64; i += T) l_adlen[i] = P.ad_len[i];
        for (uint32_t i = tid; i < P.ad_lds * hs; i += T) l_adf[i] = 0;
        for (uint32_t i = lane; i < SPAN_R * n_ad; i += 64) l_first[i] = 0xFFFFFFFFu;
    }
    __syncthreads();

    const uint32_t q = (uint32_t)lane >> 2, c = (uint32_t)lane & 3;
    /* which bytes of the lane's last two sequence dwords (j = 0, 1: dword 2 NW - 2 + j; positions 16 (2 NW - 2 + j) + 4 c ..)
       are bases of a read of U_ positions: behind the end of a read lies text of the record */
    auto tail_keep = [](uint32_t U_, uint32_t c_, int j) {
        const uint32_t p0 = 16 * (2 * NW - 2 + j) + 4 * c_;
        const uint32_t left = U_ - p0;   /* (mod 2^32) */
        const uint32_t nvalid = p0 < U_ ? (left > 4u ? 4u : left) : 0u;
        return nvalid >= 4 ? 0xFFFFFFFFu : (1u << (8 * nvalid)) - 1u;
    };
    /* One read length, no PerTileQuality riding: the two masks are made once, from an OPAQUE copy of U.  Written as
       `p0 < U ? min(4u, U - p0) : 0u` with the kernel's own U, hipcc (ROCm 7.2) concludes that p0 < U holds for every
       lane -- U > 44 at two windows -- and deletes the tests U - 1 > 35 / 39 / 43 that guard the chain's last steps far below:
       k_span<1|2,QC,uniform,both> then summed the error rates of the text behind a short read's qualities (NaN: 'Not a
       valid phred character' for every read of 1-15 and 33-47 bases; profiles/r5/exp_late_kernel_ab.txt, item 4).  The
       padding code always worked on opaque copies; now the reason is known.
       Round 6, the root cause (scripts/hipcc_speculation_noundef/): a compiler bug, not undefined behaviour here.  clang marks
       the result of min() `noundef`; InstCombine makes the U - p0 inside the `p0 < U` arm a `sub nuw`; AMD LLVM 22's
       SpeculativeExecutionPass hoists `call noundef @llvm.umin(4, sub nuw U, p0)` above the branch WITHOUT dropping the
       `noundef` -- for U < p0 the hoisted call now returns poison where it promised not to, which is undefined behaviour the
       source never had -- and ConstraintElimination, entitled to by that IR, concludes U >= p0 >= 48 and folds the guards.
       The library is built with -Xclang -no-enable-noundef-analysis since (build.py: no such promises, nothing to keep while
       hoisting; same-box A/B of every configuration: no difference); the opaque copy stays. */
    uint32_t keep_u0 = 0xFFFFFFFFu, keep_u1 = 0xFFFFFFFFu;
    if constexpr (!SEG && !PT && !LONG) {
        uint32_t Uo = U;
        asm volatile("" : "+s"(Uo));
        keep_u0 = tail_keep(Uo, c, 0);
        keep_u1 = tail_keep(Uo, c, 1);
    }
    /* DMA: piece i = 64 k + lane of a slot is 16 bytes of row i / PR: of its sequence (the first
       2 NW pieces), of its qualities (the next 2 NW; SPLIT: the stream the wave's role names), or
       the unused last one; where a row's streams start (relative to the span's first record) is
       read from l_rows */
    /* per piece (l_dma[k][lane]): byte offset into l_rows | offset inside the stream << 8 | loaded at all << 31 */
    uint32_t *l_dma = (uint32_t *)(smem + L.dma);
    if (wave == 0) {
#pragma unroll
        for (int k = 0; k < (int)ND; k++) {
            const uint32_t i = 64 * k + lane, row = i / PR, pir = i % PR, stream = !SPLIT && pir >= 2 * NW;
            const bool on = i < SPAN_R * PR && pir < (LONG ? 2 * NW + 1 : PR - 1);
            /* LONG: the row starts 16 bytes in front of the segment (l_rows points there); bit 30: that piece */
            l_dma[64 * k + lane] = on ? (row * (SEG ? 16 : 8) + stream * (SEG ? 8 : 4)) | (((pir - stream * 2 * NW) * 16) << 8) | 0x80000000u |
                                        (LONG && pir == 0 ? 0x40000000u : 0u) : 0;
        }
    }
    __syncthreads();
    /* the spans of this wave (SPLIT: of this pair of waves): s, s + stride, ... < s_end (SEG: set per length below) */
    constexpr int WPS = SPLIT ? 2 : 1;   /* waves per sequence of spans */
    const int seqs = W / WPS, my_seq = wave / WPS;
    uint64_t s_end = P.n / SPAN_R;
    uint64_t s_last = ~0ull, seg_first = 0, seg_span0 = 0;   /* SEG: the last span of the length (last_rows reads in it), its first read and span */
    uint32_t last_rows = SPAN_R;
    const uint64_t stride = SEG || PT ? (uint64_t)seqs : (uint64_t)gridDim.x * seqs;
    uint64_t s = (uint64_t)blockIdx.x * seqs + my_seq;
    if constexpr (PT) {   /* a contiguous stretch per workgroup */
        const uint64_t chunk = (s_end + gridDim.x - 1) / gridDim.x, lo = min(s_end, blockIdx.x * chunk);
        s = lo + my_seq;
        s_end = min(s_end, lo + chunk);
    }
    uint32_t role = SPLIT ? (uint32_t)wave & 1 : 0;   /* SPLIT: 0 = the bases of the span, 1 = its qualities; flips with every span */

    /* The 640 bytes of a span's metas come through LDS too (one more DMA of 40 lanes, two buffers
       per wave): the loop below holds no load hipcc counts, or its waits for one (vmcnt counts in
       order) would wait for the DMA issued in front of it.  Meta buffer k goes with slot k. */
    const uint32_t meta_base0 = lds_addr(smem + L.meta) + wave * (SEG ? 256 : SPAN_META_LDS);
    const uint32_t meta_base = PT ? (uint32_t)__builtin_amdgcn_readfirstlane(meta_base0) : meta_base0;
    auto issue_meta = [&](uint64_t sp, uint32_t maddr) {
        if constexpr (SEG) {   /* row k of span sp: read first + 16 (sp - span0) + k of the length, the last one again behind the end */
            if (lane < (int)SPAN_R) {
                const uint32_t valid = sp == s_last ? last_rows : SPAN_R;
                dma16((const uint8_t *)(P.span_rows + seg_first + (sp - seg_span0) * SPAN_R + min((uint32_t)lane, valid - 1)),
                      __builtin_amdgcn_readfirstlane(maddr));
            }
        } else {
            uint32_t lv = (uint32_t)lane;   /* opaque: the lane's part of the address is made here, not kept in two registers across the span */
            asm volatile("" : "+v"(lv));
            if (lv < 2 * SPAN_R)   /* bytes 0 .. 31 of every meta: record_start, sequence_offset, qualities_offset */
                dma16((const uint8_t *)(P.metas + sp * SPAN_R) + 40 * (lv >> 1) + 16 * (lv & 1), __builtin_amdgcn_readfirstlane(maddr));
        }
    };
    typedef uint32_t sq_u32x4 __attribute__((ext_vector_type(4)));
    sq_u32x4 name_next = {0, 0, 0, 0}, name_cur = {0, 0, 0, 0};   /* PT: 16 bytes of the read's header (the span in flight / the span being counted) */
    uint32_t nlen_next = 0, nlen_cur = 0;                          /* PT: its name_length */
    uint32_t rec_next = 0;   /* SEG: the record behind row q of the span issue() was last called for */
    uint32_t urow_next = 0, urow_cur = 0, pos_base = 0;   /* LONG: positions of row q inside the segment (0: a filler row); where the segment starts */
    /* rl: SPLIT: the stream to fetch (the role the wave has in that span) */
    const uint32_t ends_addr = __builtin_amdgcn_readfirstlane(lds_addr(smem + L.ends) + wave * 512);   /* PAIR = 2 */
    auto issue = [&](uint32_t slot_addr, uint32_t maddr, uint32_t rl, uint64_t sp) {
        if constexpr (SEG) {
            const uint32_t ma = maddr + 16 * q;
            unsigned long long seq = *(SQ_LDS const unsigned long long *)(uintptr_t)ma;
            const uint32_t qd = lds_u32(ma + 8);
            rec_next = lds_u32(ma + 12);
            if constexpr (LONG) {   /* the row: sequence start | read length << 40; the segment starts pos_base behind it */
                const uint32_t Lr = (uint32_t)(seq >> 40);
                urow_next = Lr > pos_base ? min(U, Lr - pos_base) : 0;
                seq = (seq & ((1ull << 40) - 1)) + pos_base - PRE;
            }
            if (c == 0) {
                *(SQ_LDS unsigned long long *)(uintptr_t)(lds_addr(l_rows) + 16 * q) = seq;
                *(SQ_LDS unsigned long long *)(uintptr_t)(lds_addr(l_rows) + 16 * q + 8) = seq + qd;
            }
            const uint32_t roff = lds_addr(l_rows) + (SPLIT ? 8 * rl : 0);
            uint32_t pk[ND];
            unsigned long long rr[ND];
#pragma unroll
            for (int k = 0; k < (int)ND; k++) pk[k] = l_dma[64 * k + lane];
#pragma unroll
            for (int k = 0; k < (int)ND; k++) rr[k] = *(SQ_LDS const unsigned long long *)(uintptr_t)(roff + (pk[k] & 0xFFu));
#pragma unroll
            for (int k = 0; k < (int)ND; k++)
                if ((int32_t)pk[k] < 0 && !(LONG && (pk[k] & 0x40000000u) && pos_base == 0))   /* nothing lies in front of a read's first segment */
                    dma16(P.buf + rr[k] + ((pk[k] >> 8) & 0xFFFu), __builtin_amdgcn_readfirstlane(slot_addr + 1024 * k));
            return;
        }
        const uint32_t ma = maddr + 32 * q;
        const unsigned long long m_rs = *(SQ_LDS const unsigned long long *)(uintptr_t)ma; /* record_start */
        const uint32_t m_so = lds_u32(ma + 12), m_qo = lds_u32(ma + 20);             /* sequence_offset, qualities_offset */
        const unsigned long long base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(m_rs >> 32)) << 32) |
                                        (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)m_rs);
        const uint32_t rel = (uint32_t)(m_rs - base);
        if (c == 0) { l_rows[2 * q] = rel + m_so; l_rows[2 * q + 1] = rel + m_qo; }
        const uint8_t *g0 = P.buf + base;
        if constexpr (PT) {   /* lane c: bytes 16 c .. 16 c + 15 of its read's header; used behind the wait at the top of the loop */
            nlen_next = lds_u32(ma + 8);
            /* a load hipcc KNOWS of (unlike the DMA): it may move or spill the four registers only behind a wait of its
               own, which it puts in front of their first use -- right behind the wait at the top of the loop, where
               nothing is in flight any more.  (As inline asm the load was consumed early now and then: the compiler
               takes the output of an asm statement for valid at once and is free to copy it.) */
            const uint8_t *np = g0 + (long long)(int32_t)rel + 16 * c;
            __builtin_memcpy(&name_next, np, 16);
        }
        if constexpr (PAIR == 2) {   /* the ends of read 2 of the span's 16 pairs: 32 lanes x 16 bytes, contiguous */
            uint32_t lv = (uint32_t)lane;   /* opaque: the lane's part of the address is made here */
            asm volatile("" : "+v"(lv));
            if (lv < 32) dma16(par(PAR_ENDS) + sp * (SPAN_R * 32) + 16 * lv, __builtin_amdgcn_readfirstlane(ends_addr));
        }
        const uint32_t roff = lds_addr(l_rows) + (SPLIT ? 4 * rl : 0);
        uint32_t pk[ND];
        int32_t rr[ND];
        int lane_i = lane;   /* PT: opaque, so that the address of the lane's table entries is made here and not kept across the spans */
        if constexpr (PT) asm volatile("" : "+v"(lane_i));
#pragma unroll
        for (int k = 0; k < (int)ND; k++) pk[k] = l_dma[64 * k + lane_i];
#pragma unroll
        for (int k = 0; k < (int)ND; k++) rr[k] = (int32_t)lds_u32(roff + (pk[k] & 0xFFu));
#pragma unroll
        for (int k = 0; k < (int)ND; k++)
            if ((int32_t)pk[k] < 0)
                dma16(g0 + (long long)rr[k] + ((pk[k] >> 8) & 0xFFFu), __builtin_amdgcn_readfirstlane(slot_addr + 1024 * k));
    };

    uint32_t Lmain = 4 * ((U - 1) / 4), nsteps = Lmain / 4; /* _qcmodule.c:2062,2068 */
    const uint32_t h = (uint32_t)lane >> 5, pl = (uint32_t)lane & 31;
    uint32_t cnt[NW];
#pragma unroll
    for (int w = 0; w < NW; w++) cnt[w] = 0;
    uint32_t since_flush = 0;
    auto flush_counts = [&]() {
        uint32_t plv = pl;   /* PT: opaque -- the positions 32 w + pl are made here, not kept in NW registers across the spans */
        if constexpr (PT) asm volatile("" : "+v"(plv));
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const uint32_t p = 32 * w + plv;
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

    /* the workgroup's positional histograms to the device tables (end-anchored = a window of the
       positional ones: every read counted since the last merge has length U), then zeroed again.
       fill: SEG: filler rows the workgroup counted in this stretch (their qualities sit in row 12
       of every position; what else is there are real 0x80 bytes, bin 11 like every invalid byte) */
    auto merge_hist = [&](bool zero, uint32_t fill) {
        int tid_m = tid;   /* PT builds make the thread's index again here instead of keeping it across the loop (their registers are gone) */
        if constexpr (PT) { tid_m = (int)threadIdx.x; asm volatile("" : "+v"(tid_m)); }
        if (AD)
            for (uint32_t i = tid_m; i < P.ad_lds * hs; i += T) {
                const uint32_t v = l_adf[i];
                if (!v) continue;
                if (zero) l_adf[i] = 0;
                const uint32_t a = i / hs, start = i % hs;
                atomicAdd(&P.ad_fwd[a * P.ad_cap + start], (unsigned long long)v);
                atomicAdd(&P.ad_rev[a * P.ad_cap + (U - 1 - start)], (unsigned long long)v);
            }
        const uint32_t ean = LONG ? 0 : min(P.ea_len, U);   /* LONG: the end-anchored tables are k_long_ea's */
        unsigned long long *t_base = P.qc_base, *t_phred = P.qc_phred, *t_ea_base = P.qc_ea_base, *t_ea_phred = P.qc_ea_phred;
        if constexpr (PAIR == 2) {
            t_base = (unsigned long long *)par(PAR_QC_BASE); t_phred = (unsigned long long *)par(PAR_QC_PHRED);
            t_ea_base = (unsigned long long *)par(PAR_QC_EA_BASE); t_ea_phred = (unsigned long long *)par(PAR_QC_EA_PHRED);
        }
        for (uint32_t i = tid_m; i < hs * BASE_COLS; i += T) {
            const uint32_t v = l_hist_base[i], cc = i / hs, pos = i % hs;
            if (zero) l_hist_base[i] = 0;
            if (!v || pos >= U) continue;
            atomicAdd(&t_base[(uint64_t)(pos_base + pos) * 5 + cc], (unsigned long long)v);
            if (pos >= U - ean) atomicAdd(&t_ea_base[(uint64_t)(P.ea_len - U + pos) * 5 + cc], (unsigned long long)v);
        }
        for (uint32_t i = tid_m; i < hs * PROWS; i += T) {
            uint32_t v = l_hist_phred[i], cc = i / hs;
            const uint32_t pos = i % hs;
            if (zero) l_hist_phred[i] = 0;
            if (pos >= U) continue;
            if (cc >= PHRED_COLS) {   /* the row of the padding: LONG: nothing else is there (the host sends batches with an invalid byte elsewhere) */
                if (LONG) continue;
                v -= fill;
                cc = PHRED_COLS - 1;
            }
            if (!v) continue;
            atomicAdd(&t_phred[(uint64_t)(pos_base + pos) * 12 + cc], (unsigned long long)v);
            if (pos >= U - ean) atomicAdd(&t_ea_phred[(uint64_t)(P.ea_len - U + pos) * 12 + cc], (unsigned long long)v);
        }
    };

    /* PT: the run this wave is in: reads of ONE tile, back to back; pt_acc[w]: the sum of their error rates at position
       32 w + pl over the rows 8 h .. 8 h + 7 of every span of the run (the two halves of the wave are added when the run
       is written out) */
    double pt_acc[PT ? NW : 1];
    uint32_t pt_lo = 0, pt_hi = 0, pt_reads = 0;   /* wave-uniform: the run's tile id, its reads so far */
    bool pt_full = false;                          /* the staging area has overflowed (reads of mixed tiles): nothing more is staged, the host counts the batch by the older route */
    if constexpr (PT) {
#pragma unroll
        for (int w = 0; w < NW; w++) pt_acc[w] = 0.0;
    }
    /* the run goes to the staging area (k_pt_fold adds it to the table row of its tile) */
    auto pt_flush = [&](uint32_t tlo, uint32_t thi, uint32_t reads) {
        if constexpr (PT) {
            uint32_t idx = 0;
            uint32_t lv = (uint32_t)lane;   /* opaque: nothing of the lane's part is made outside and kept across the spans */
            asm volatile("" : "+v"(lv));
            if (lv == 0) idx = atomicAdd((unsigned int *)par(PAR_NRUNS), 1u);
            idx = __builtin_amdgcn_readfirstlane(idx);
            if (idx >= P.pt_runs_cap) pt_full = true;
            if (idx < P.pt_runs_cap) {
                if (lv == 0) {
                    PtRun r;
                    r.tile = (long long)(((unsigned long long)thi << 32) | tlo);
                    r.reads = reads;
                    r.pad = 0;
                    ((PtRun *)par(PAR_RUNS))[idx] = r;
                }
                double *sums = (double *)par(PAR_SUMS);
#pragma unroll
                for (int w = 0; w < NW; w++) {
                    const unsigned long long mine = (unsigned long long)__double_as_longlong(pt_acc[w]);
                    const uint32_t olo = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lv ^ 32) << 2), (int)(uint32_t)mine);
                    const uint32_t ohi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lv ^ 32) << 2), (int)(uint32_t)(mine >> 32));
                    const double other = __longlong_as_double((long long)(((unsigned long long)ohi << 32) | olo));
                    const uint32_t pos = 32 * w + (lv & 31);
                    if (lv < 32 && pos < U) sums[(uint64_t)idx * U + pos] = pt_acc[w] + other;
                }
            }
#pragma unroll
            for (int w = 0; w < NW; w++) pt_acc[w] = 0.0;
        }
    };

    /* ---- one span, or one stream of it: DS: what is counted of the bases, DQ: of the qualities ---- */
    int cur = 0;
    uint32_t rec_cur = 0;
    auto body = [&](auto ds_c, auto dq_c) {
        constexpr bool DS = decltype(ds_c)::value, DQ = decltype(dq_c)::value, ADr = AD && DS;
        const uint32_t sa = slot_base + cur * SLOT;
        const uint64_t r = SEG ? (uint64_t)rec_cur : s * SPAN_R + q;
        const uint32_t nv = SEG && s == s_last ? last_rows : SPAN_R;   /* rows q >= nv are filler */
        const uint32_t seq_row = sa + q * ROWB + PRE, qual_row = seq_row + QOFF;
        const uint32_t urow = LONG ? (q < nv ? urow_cur : 0u) : 0u;   /* LONG: positions of this row inside the segment */

        if constexpr (PAIR == 1) {
            /* the first and the last 16 bases of the row, raw (the scan of read 1's pass compares raw bytes, :5699-5703): lanes
               0, 1 of the quad the two halves of the head, lanes 2, 3 of the tail (any alignment: byte reads) */
            unsigned long long v;
            if (c < 2) {
                v = *(SQ_LDS const unsigned long long *)(uintptr_t)(seq_row + 8 * c);
            } else {
                const uint32_t a = seq_row + U - 16 + 8 * (c - 2);
                uint32_t lo = 0, hi = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) { lo |= lds_u8(a + k) << (8 * k); hi |= lds_u8(a + 4 + k) << (8 * k); }
                v = ((unsigned long long)hi << 32) | lo;
            }
            uint8_t *dst = par(PAR_ENDS) + r * 32 + 8 * c;
            asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(dst), "v"(v) : "memory");
            __builtin_amdgcn_sched_barrier(0);   /* done before the next block starts: the builds of 168 registers have none to spare for an overlap */
        }
        /* PT: the tile of every row (lanes c == 0), illumina_header_to_tile_id :3088-3121 */
#ifdef SQ_SPAN_MARK
        asm volatile("; PT_PARSE_BEGIN" ::: "memory");
#endif
        uint32_t t0lo = 0, t0hi = 0;   /* wave-uniform: the tile of row 0 */
        bool one_tile = false;         /* every row has it */
        if constexpr (PT) {
            /* every lane of the quad on its own 16 bytes, five small values exchanged by DPP (sq_pass.h: quad_*; the same
               functions run on the host under tests/test_boundary_cpu.py).  What they decline -- a header of more than 64
               bytes whose fifth colon lies behind byte 64, a tile of 9 .. 18 digits -- lane 0 parses byte by byte. */
            const uint32_t wv[4] = {name_cur.x, name_cur.y, name_cur.z, name_cur.w};
            const uint32_t m16 = quad_colon_mask(wv, c, nlen_cur);
            const uint32_t cnt16 = (uint32_t)__builtin_popcount(m16);
            const uint32_t k0 = quad_bcast<0x00>(cnt16), k1 = quad_bcast<0x55>(cnt16), k2 = quad_bcast<0xAA>(cnt16);
            const uint32_t before = (c > 0 ? k0 : 0u) + (c > 1 ? k1 : 0u) + (c > 2 ? k2 : 0u);
            uint32_t p4 = quad_kth_colon(m16, before, 4, c), p5 = quad_kth_colon(m16, before, 5, c);
            p4 = min(p4, quad_bcast<0xB1>(p4)); p4 = min(p4, quad_bcast<0x4E>(p4));   /* the quad's minimum: quad_perm [1,0,3,2], [2,3,0,1] */
            p5 = min(p5, quad_bcast<0xB1>(p5)); p5 = min(p5, quad_bcast<0x4E>(p5));
            const unsigned long long part = quad_window8(wv, c, p4 == 0xFFu ? 0u : p4 + 1);
            uint32_t d_lo = (uint32_t)part, d_hi = (uint32_t)(part >> 32);
            d_lo |= quad_bcast<0xB1>(d_lo); d_lo |= quad_bcast<0x4E>(d_lo);
            d_hi |= quad_bcast<0xB1>(d_hi); d_hi |= quad_bcast<0x4E>(d_hi);
            const long long quad_tile = quad_tile_value(p4, p5, nlen_cur, ((unsigned long long)d_hi << 32) | d_lo);
            uint32_t tile_lo = 0, tile_hi = 0;
            if (c == 0) {
                long long tile = quad_tile;
                if (tile == QUAD_TILE_SLOW) tile = tile_id_of(P.buf + P.metas[r].record_start, nlen_cur);
                tile_lo = (uint32_t)tile;
                tile_hi = (uint32_t)((unsigned long long)tile >> 32);
                /* the rows' tiles wait in l_rows (free between two calls of issue()) for the span a tile ends in */
                l_rows[2 * q] = tile_lo;
                l_rows[2 * q + 1] = tile_hi;
                if (tile < 0) atomicMin((unsigned long long *)par(PAR_BAD), (unsigned long long)(P.pt_first_index + r));   /* :3137-3148: PerTileQuality ends here */
                long long *tiles_out = (long long *)par(PAR_TILES);
                if (tiles_out) {
                    long long *dst = tiles_out + r;
                    asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(dst), "v"(tile) : "memory");
                }
            }
            t0lo = __builtin_amdgcn_readfirstlane(tile_lo);
            t0hi = __builtin_amdgcn_readfirstlane(tile_hi);
            one_tile = __builtin_amdgcn_ballot_w64(c == 0 && (tile_lo != t0lo || tile_hi != t0hi)) == 0 && (int32_t)t0hi >= 0;
            __builtin_amdgcn_sched_barrier(0);   /* the parse is over before the class pass asks for its registers */
#ifdef SQ_SPAN_MARK
            asm volatile("; PT_PARSE_END" ::: "memory");
#endif
        }
        /* ---------------- phase S: four lanes per read ----------------
           (1) class codes: lane c of a quad takes dwords c, c + 4, ... of its read's sequence (8 rows
           x 4 lanes of an instruction in 32 banks), writes the codes back in place (phase H and the
           automaton read them) and counts G/C and non-ACGT bases */
        /* G/C bases (low nibble) and bases that are none of A C G T (high nibble) of the lane's dwords, per byte lane; padding
           counts as nothing.  A nibble holds 15: a second register from the 16th dword on */
        constexpr int NF = (2 * NW + 14) / 15;
        uint32_t gacc = 0, nacc = 0;   /* LONG: the same per byte lane as 4 x / 8 x a count, padding counted in both */
        bool long_usual = false;       /* LONG: the span took the short way (its counts are in facc) */
        uint32_t facc[NF];
#pragma unroll
        for (int k = 0; k < NF; k++) facc[k] = 0;
        auto gcn_sum = [&](const uint32_t (&f)[NF]) {   /* the lane's G/C count | its count of other bases << 16 */
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int k = 0; k < NF; k++) { lo = sum_bytes(f[k] & 0x0F0F0F0Fu, lo); hi = sum_bytes((f[k] >> 4) & 0x0F0F0F0Fu, hi); }
            return lo | (hi << 16);
        };
        if constexpr (DS) {
            const uint32_t cb = seq_row + 4 * c;
            uint32_t Uv = U, cv = c; /* opaque: what is made of them is made per span, not kept across the spans */
            if constexpr (LONG) asm volatile("" : "+v"(cv)); else asm volatile("" : "+s"(Uv), "+v"(cv));
            uint32_t raw[2 * NW];
#pragma unroll
            for (int t = 0; t < 2 * NW; t++) raw[t] = lds_u32(cb + 16 * t);
            /* LONG: every row has an end of its own, but the rows that end inside their segment are the
               last ones of a stretch (the reads are sorted by length): most spans hold full rows only */
            const bool ragged_rows = LONG && __builtin_amdgcn_ballot_w64(urow < U) != 0;
            /* Behind the end of a read lies text of the record, not bases: U > 32 (NW - 1), so only the lane's last two dwords can
               reach there; `keep` masks their bytes that are bases (LONG: every row has an end of its own, see below) */
            uint32_t keep[2] = {keep_u0, keep_u1};
            if constexpr (!LONG && (SEG || PT)) {   /* a length per stretch / no register to spare across the spans: made per span */
                keep[0] = tail_keep(Uv, cv, 0);
                keep[1] = tail_keep(Uv, cv, 1);
            }
            auto classes = [&](auto exact_c) {
#pragma unroll
                for (int t = 0; t < 2 * NW; t++) {
                    uint32_t cl, fl;
                    if constexpr (decltype(exact_c)::value) cl = cls6_gcn_of_dword(lds_u32(cb + 16 * t), &fl);   /* (read again: this path holds no registers the usual one lacks) */
                    else { cl = cls6_usual(raw[t]); fl = gcn_usual(raw[t]); }
                    if (t >= 2 * NW - 2) {
                        const uint32_t k = keep[t - (2 * NW - 2) < 0 ? 0 : t - (2 * NW - 2)];
                        cl = (cl & k) | (CLS6_PAD4 & ~k);
                        fl &= k;
                    }
                    lds_store_u32(cb + 16 * t, cl);
                    facc[t / 15] += fl;
                }
            };
            /* Is there a byte in the wave's rows that a sequencer does not write?  (Nearly never: then every dword is one v_perm
               for the classes and one for the counts instead of cls6_of_dword's ten instructions and six for the counts.) */
            auto unusual = [&]() {
                uint32_t bad = 0;
#pragma unroll
                for (int t = 0; t < 2 * NW; t++) {
                    uint32_t x = cls6_unusual(raw[t]);
                    if (t >= 2 * NW - 2) x &= keep[t - (2 * NW - 2) < 0 ? 0 : t - (2 * NW - 2)];
                    bad |= x;
                }
                return __builtin_amdgcn_ballot_w64(bad != 0 && q < nv) != 0;
            };
            if constexpr (LONG) {
                /* (the builds for long reads keep cls6_of_dword for spans with rows that end inside the segment: all four code paths do not fit their registers) */
                auto classes_long = [&](auto ragged_c) {
#pragma unroll
                    for (int t = 0; t < 2 * NW; t++) {
                        uint32_t cl = cls6_of_dword(raw[t]);
                        if constexpr (decltype(ragged_c)::value) {
                            const uint32_t p0 = 16 * t + 4 * cv;
                            const uint32_t left = urow - p0;   /* (mod 2^32; spelled out: see tail_keep) */
                            cl = pad_tail(cl, p0 < urow ? (int)(left > 4u ? 4u : left) : 0, CLS6_PAD4);
                        }
                        lds_store_u32(cb + 16 * t, cl);
                        gacc += cl & 0x04040404u;                 /* C, G and padding */
                        nacc += cl & (cl >> 1) & 0x08080808u;     /* N and padding */
                    }
                };
                /* a span of full segments (nearly all of them: the reads are sorted by length) whose rows hold the usual letters
                   only: the two v_perms of the other builds; the counts go to facc */
                if (!ragged_rows) {
                    uint32_t bad = 0;
#pragma unroll
                    for (int t = 0; t < 2 * NW; t++) bad |= cls6_unusual(raw[t]);
                    long_usual = __builtin_amdgcn_ballot_w64(bad != 0) == 0;
                }
                if (long_usual) {
#pragma unroll
                    for (int t = 0; t < 2 * NW; t++) {
                        lds_store_u32(cb + 16 * t, cls6_usual(raw[t]));
                        facc[t / 15] += gcn_usual(raw[t]);
                    }
                } else if (ragged_rows) classes_long(std::true_type{}); else classes_long(std::false_type{});
            } else {
                if (unusual()) classes(std::true_type{}); else classes(std::false_type{});
            }
            if constexpr (LONG && AD) {   /* the 12 bases in front of the segment: lanes 1 .. 3 of the quad, a dword each */
                if (cv > 0) {
                    const uint32_t a = seq_row - 16 + 4 * cv;
                    lds_store_u32(a, pos_base && urow ? cls6_of_dword(lds_u32(a)) : CLS6_PAD4);
                }
            }
            if (!LONG && SEG && nv < SPAN_R) {   /* the last span of a length: filler rows become padding */
                if (q >= nv) {
#pragma unroll
                    for (int t = 0; t < 2 * NW; t++) lds_store_u32(cb + 16 * t, CLS6_PAD4);
                }
            }
        }
        if constexpr (DS && LONG) {   /* the read's G/C and A/C/G/T counts, a segment at a time (:1997-2049; the bin: k_long_gc_bins) */
            /* G/C | other << 16 of the lane's dwords, padding counted in both (the short way meets none) */
            uint32_t gn = long_usual ? gcn_sum(facc) : (sum_bytes(gacc, 0) >> 2) | ((sum_bytes(nacc, 0) >> 3) << 16);
            gn += quad_bcast<0xB1>(gn);
            gn += quad_bcast<0x4E>(gn);
            if (c == 0 && urow) {
                const uint32_t gc_cnt = (gn & 0xFFFFu) - (SB - urow), acgt_cnt = SB - (gn >> 16);
                unsigned int *dst = P.long_gc + 2 * (uint64_t)rec_cur;
                asm volatile("global_atomic_add %0, %1, off\n\tglobal_atomic_add %0, %2, off offset:4" :: "v"(dst), "v"(gc_cnt), "v"(acgt_cnt) : "memory");
            }
        }
        if constexpr (DQ && SEG && !LONG) {
            if (nv < SPAN_R && q >= nv) {
#pragma unroll
                for (int t = 0; t < 2 * NW; t++) lds_store_u32(qual_row + 4 * c + 16 * t, PAD4);
            }
        }
        if constexpr (DQ && LONG) {   /* rows that end inside the segment (and filler rows): padding behind the end */
            if (__builtin_amdgcn_ballot_w64(urow < U)) {
#pragma unroll
                for (int t = 0; t < 2 * NW; t++) {
                    const uint32_t a = qual_row + 4 * c + 16 * t, p0 = 16 * t + 4 * c;
                    if (p0 + 4 > urow) lds_store_u32(a, pad_tail(lds_u32(a), p0 < urow ? (int)(urow - p0) : 0, PAD4));
                }
            }
        }
        SPAN_PHASE(0);   /* class codes */
        /* (2) the automaton.  One table read per base that depends on the read before it is the
           only chain of dependent LDS round trips in the span (a round trip under this load is a
           few hundred cycles), so a lane cuts its quarter into S pieces of D dwords and walks S
           automatons side by side, each restarted W4T dwords in front of its piece: W4T + D rounds
           of four dependent steps instead of W4T + Q4.  Everything else that is left to do for
           the span has no chain and is spread over those rounds, hand scheduled into the waits:
           the lane's f64 chain (positions c, c + 4, ...; CG groups of four steps per round) and
           phase H (lane = position: lane (h, pl) counts position pl of every window of 32 for
           the rows 8 h .. 8 h + 7, whose bytes at that position one transposing read per window
           and stream hands it; HI of the 8 NW cells per round).  What a round consumes was loaded in the round before it; the
           loads are asm the compiler does not wait for (rd_*, wait_lgkm + tie). */
        bool any_hit = false;
        double acc = 0.0, tail0 = 0.0, tail1 = 0.0, tail2 = 0.0, tail3 = 0.0;
        {
            constexpr int S = SPAN_S, D = ((int)Q4 + S - 1) / S, WT = ADr ? W4T : 0;
            constexpr int NR = ADr ? WT + D : NW;   /* without the automaton: rounds of eight items and two groups */
            constexpr int HALF = (int)SPAN_R / 2, ITEMS = NW * HALF, HI = (ITEMS + NR - 1) / NR;
            static_assert(HI <= HALF, "a round stays inside two windows");
            constexpr int KRG = LONG ? 0 : 2 * (NW - 1), CG = (KRG + NR - 1) / NR;   /* groups of four chain steps that exist whatever U is (LONG: the chains are k_read_sums' work) */
            constexpr int KR4 = 4 * KRG;                                  /* chain steps the rounds carry */
            constexpr int TRN = (DS ? 1 : 0) + (DQ ? 1 : 0);              /* transposing reads per window */
            uint32_t co = c;   /* opaque: the padding masks of the rounds are made per span, not kept across spans */
            asm volatile("" : "+v"(co));
            const uint32_t abase = seq_row + 4u * (Q4 * c) - 4u * WT;     /* dword (piece s, round t): abase + 4 (s D + t) */
            const uint32_t qp = qual_row + c;
            /* transposing reads: lane 2 q + p of a group of 16 hands in row 8 h + q, bytes 8 p .. 8 p + 7 of
               the group's 16 positions; window w and the quality stream by immediate offset */
            const uint32_t trb = sa + PRE + (8 * h + (((uint32_t)lane & 15) >> 1)) * ROWB + 16 * (((uint32_t)lane >> 4) & 1) + 8 * ((uint32_t)lane & 1);
            const uint32_t hpp = lds_addr(l_hist_phred + pl);
            const uint32_t one = 1;
            uint32_t rec = 0, rec2 = 0, st0 = dfa_root; /* a lane's first two matches of the span: row of the automaton | end position << 12 | 1 << 31 */
            bool multi = false;                  /* a third one: the wave walks its quarters again, one base at a time */
            wait_lgkm<0>();
            /* what round 0 consumes */
            uint32_t qc[4 * (CG > 0 ? CG : 1)], cl[S], st[S];
            sq_u32x2 ts[NW], tq[NW];   /* the 8 rows' class codes / qualities at the lane's position of window w */
            if constexpr (DS) ts[0] = rd_tr8<0>(trb);
            if constexpr (DQ) tq[0] = rd_tr8<(int)QOFF>(trb);
            /* The chain's quality bytes: a group of four steps of the quad's four chains is the 16 bytes at 16 G of the row --
               ONE ds_read_b128 (all four lanes the same address), lane c takes byte c of each dword.  Until round 5: four
               ds_read_u8 per group (30 LDS instructions more per span for 40 vector ones less: 2 % slower). */
            /* byte c of two dwords, times 8 (the addresses of their error rates), in three instructions: a v_perm puts byte c of
               both into the halves of one register (selector bytes c, zero, 4 + c, zero), an SDWA shift makes an address of each
               half.  (Until the end of round 5 a v_bfe and a shift per dword: + 1 % on the headline, same box.) */
            const uint32_t selc = 0x0C040C00u + co * 0x00010001u, three_c = 3;
            auto chain_pair8 = [&](uint32_t w0, uint32_t w1, uint32_t &a0, uint32_t &a1) {
                const uint32_t p = __builtin_amdgcn_perm(w1, w0, selc);
                a0 = shl3_word_of<0>(p, three_c);
                a1 = shl3_word_of<1>(p, three_c);
            };
            auto load_group = [&](auto gc, auto Gc) {   /* qc[4 g .. 4 g + 3] = the dwords of group G */
                constexpr int g = decltype(gc)::value, G = decltype(Gc)::value;
                const sq_u32x4v v = rd_b128<16 * G>(qual_row);
                qc[4 * g] = v.x; qc[4 * g + 1] = v.y; qc[4 * g + 2] = v.z; qc[4 * g + 3] = v.w;
            };
            if constexpr (DQ)
                static_for<0, CG>([&](auto gc) {
                    constexpr int g = decltype(gc)::value;
                    if constexpr (g < KRG) load_group(gc, std::integral_constant<int, g>{});
                });
            static_for<0, S>([&](auto sc) {
                constexpr int sI = decltype(sc)::value;
                st[sI] = dfa_root;
                cl[sI] = 0;
                if constexpr (ADr) cl[sI] = rd_b32<4 * (sI * D)>(abase);
            });
            wait_lgkm<0>();
            if constexpr (DS) tie2(ts[0]);
            if constexpr (DQ) {
                tie2(tq[0]);
                static_for<0, 4 * (CG > 0 ? CG : 1)>([&](auto ic) { tie(qc[decltype(ic)::value]); });
            }
            static_for<0, S>([&](auto sc) { tie(cl[decltype(sc)::value]); });

            static_for<0, NR>([&](auto tc) {
                /* one round = one dword of the lane's quarter = two steps of the two-character automaton */
                constexpr int t = decltype(tc)::value;
                constexpr bool proper = t >= WT;
                constexpr auto cap = [](int n) { return n < 15 ? n : 15; };
                constexpr auto items_of = [](int round) { return ITEMS - round * HI < 0 ? 0 : ITEMS - round * HI < HI ? ITEMS - round * HI : HI; };
                constexpr auto groups_of = [](int round) { return KRG - round * CG < 0 ? 0 : KRG - round * CG < CG ? KRG - round * CG : CG; };
                constexpr int n_l = items_of(t), n_lq = DQ ? n_l : 0;
                constexpr auto windows_through = [](int round) {
                    const int cells = (round + 1) * HI;
                    return ((cells < ITEMS ? cells : ITEMS) + HALF - 1) / HALF;
                };
                constexpr int w_lo = windows_through(t), w_hi = t + 1 < NR ? windows_through(t + 1) : w_lo;
                constexpr int n_nx = w_hi - w_lo;
                constexpr int g_now = DQ ? groups_of(t) : 0, g_nx = DQ && t + 1 < NR ? groups_of(t + 1) : 0;
                constexpr int nC1 = g_now >= 1 ? 4 : 0, nC2 = g_now >= 2 ? 4 : 0, SA = ADr ? S : 0;
                static_assert(CG <= 2, "a round carries at most two groups of chain steps");
                static_assert(S == 1, "the two-character automaton walks one piece per lane");
                uint32_t e0 = 0, e1 = 0, pc = 0;   /* pc: byte 0 / byte 2 = code of the first + 6 * code of the second character of the dword's two pairs */
                if constexpr (ADr) {
                    if (t == WT) st0 = st[0];
                    constexpr int idx = t - WT;   /* dword of the quarter, < 0: in front of it */
                    if constexpr (proper && idx >= (int)Q4) cl[0] = CLS6_PAD4;
                    else {
                        if constexpr (idx < 0 && !LONG) cl[0] = co == 0 ? CLS6_PAD4 : cl[0];   /* LONG: the bases in front of the segment are in the row (padding in front of a read) */
                        if constexpr (3 * (int)Q4 + idx >= (int)DW) cl[0] = Q4 * co + idx < DW ? cl[0] : CLS6_PAD4;
                    }
                    pc = __umul24(cl[0] >> 8, 6u) + cl[0];
                    e0 = rd_u16<0>(add_byte<0>(st[0], pc));
                }
                uint32_t l[HI];
                if constexpr (DQ)
                    static_for<0, HI>([&](auto mc) {
                        constexpr int m = decltype(mc)::value, cell = t * HI + m, w = cell / HALF, k = cell % HALF;
                        if constexpr (m < n_l) l[m] = rd_u16<SPAN_BIN_OFF>(shl1_byte<k % 4>(k < 4 ? tq[w].x : tq[w].y, one));
                    });
                double d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                if constexpr (nC1 && t > 0) { /* the quality bytes the round before asked for */
                    wait_lgkm<cap(SA + n_lq)>();
                    static_for<0, 4 * CG>([&](auto ic) { tie(qc[decltype(ic)::value]); });
                }
                if constexpr (nC1) { uint32_t a0, a1, a2, a3; chain_pair8(qc[0], qc[1], a0, a1); chain_pair8(qc[2], qc[3], a2, a3); d[0] = rd_f64(a0); d[1] = rd_f64(a1); d[2] = rd_f64(a2); d[3] = rd_f64(a3); }
                if constexpr (DS)
                    static_for<0, HI>([&](auto mc) {
                        constexpr int m = decltype(mc)::value, cell = t * HI + m, w = cell / HALF, k = cell % HALF;
                        if constexpr (m < n_l) cnt[w] = add_one_shl_byte<k % 4>(k < 4 ? ts[w].x : ts[w].y, one, cnt[w]);
                    });
                if constexpr (n_lq > 0) {
                    wait_lgkm<cap(nC1)>();   /* the histogram rows (and, older, the first step) */
                    static_for<0, HI>([&](auto mc) {
                        constexpr int m = decltype(mc)::value;
                        if constexpr (m < n_l) { tie(l[m]); inc_u32<128 * ((t * HI + m) / HALF)>(hpp + l[m], one); }
                    });
                }
                if constexpr (nC2) { uint32_t a4, a5, a6, a7; chain_pair8(qc[4], qc[5], a4, a5); chain_pair8(qc[6], qc[7], a6, a7); d[4] = rd_f64(a4); d[5] = rd_f64(a5); d[6] = rd_f64(a6); d[7] = rd_f64(a7); }
                if constexpr (ADr) {   /* the second step: behind the first one are d[0..3], the increments, d[4..7] */
                    wait_lgkm<cap(nC1 + n_lq + nC2)>();
                    tie(e0);
                    e1 = rd_u16<0>(add_byte<2>(e0, pc));
                }
                if constexpr (nC1) {
                    wait_lgkm<cap(n_lq + nC2 + SA)>();
                    tie_f64(d[0]); tie_f64(d[1]); tie_f64(d[2]); tie_f64(d[3]); acc += d[0]; acc += d[1]; acc += d[2]; acc += d[3];
                }
                /* what the next round consumes: the bytes of its phase H items ... */
                static_for<0, NW>([&](auto wc) {
                    constexpr int w = decltype(wc)::value;
                    if constexpr (w >= w_lo && w < w_hi) {
                        if constexpr (DS) ts[w] = rd_tr8<32 * w>(trb);
                        if constexpr (DQ) tq[w] = rd_tr8<32 * w + (int)QOFF>(trb);
                    }
                });
                if constexpr (nC2) {
                    wait_lgkm<cap(SA + TRN * n_nx)>();
                    tie_f64(d[4]); tie_f64(d[5]); tie_f64(d[6]); tie_f64(d[7]); acc += d[4]; acc += d[5]; acc += d[6]; acc += d[7];
                }
                /* ... its class dword and the quality bytes of its chain steps */
                uint32_t cln = 0;
                if constexpr (ADr && t + 1 < NR) cln = rd_b32<4 * (t + 1)>(abase);
                if constexpr (DQ)
                    static_for<0, CG>([&](auto gc) {
                        constexpr int g = decltype(gc)::value;
                        if constexpr (g < g_nx) load_group(gc, std::integral_constant<int, (t + 1) * CG + g>{});
                    });
                constexpr int n_cln = ADr && t + 1 < NR ? 1 : 0;
                if constexpr (ADr) {
                    wait_lgkm<cap(TRN * n_nx + n_cln + g_nx)>();
                    tie(e1);
                    st[0] = e1;
                    if constexpr (proper) {
                        if (max(e0, e1) >= dfa_hit) {
                            constexpr int idx = t - WT;
                            const uint32_t ee[2] = {e0, e1};
#pragma unroll
                            for (uint32_t j = 0; j < 2; j++) {
                                if (ee[j] < dfa_hit) continue;   /* position: of the step's second character */
                                const uint32_t v = 0x80000000u | ((4 * (Q4 * co + (uint32_t)idx) + 2 * j + 1) << 12) | span_dfa_state(ee[j] - dfa_root);
                                if (!rec) rec = v;
                                else if (!rec2) rec2 = v;
                                else multi = true;
                            }
                        }
                    }
                }
                wait_lgkm<cap(g_nx)>();
                tie(cln);
                cl[0] = cln;
                static_for<0, NW>([&](auto wc) {
                    constexpr int w = decltype(wc)::value;
                    if constexpr (w >= w_lo && w < w_hi) {
                        if constexpr (DS) tie2(ts[w]);
                        if constexpr (DQ) tie2(tq[w]);
                    }
                });
            });
            wait_lgkm<0>();
            SPAN_PHASE(1);   /* the rounds */
            if constexpr (PT && DQ) {
                if (P.pt_runs && !pt_full) {
                    /* error rates of the eight qualities the lane holds per window (tq[w]: rows 8 h .. 8 h + 7 at position
                       32 w + pl); rows: bit 8 h + k set = row 8 h + k is added.  The lookups of window w + 1 are in flight
                       while those of window w are added */
                    auto pt_add = [&]() {
                        const uint32_t three = 3;
                        static_for<0, NW>([&](auto wc) { tie2(tq[decltype(wc)::value]); });   /* the addresses are made here, not in the rounds */
                        /* four lookups at a time (half a window: the rows 8 h + 4 j .. 8 h + 4 j + 3), the next four in flight while these are added */
                        double e[2][4];
                        auto look = [&](auto gc) {
                            constexpr int g = decltype(gc)::value, w = g / 2, j = g % 2;
                            static_for<0, 4>([&](auto kc) {
                                constexpr int k = decltype(kc)::value;
                                e[g & 1][k] = rd_f64(shl3_byte_of<k>(j ? tq[w].y : tq[w].x, three));
                            });
                        };
                        look(std::integral_constant<int, 0>{});
                        static_for<0, 2 * NW>([&](auto gc) {
                            constexpr int g = decltype(gc)::value;
                            if constexpr (g + 1 < 2 * NW) look(std::integral_constant<int, g + 1>{});
                            if constexpr (g + 1 < 2 * NW) wait_lgkm<4>(); else wait_lgkm<0>();
                            static_for<0, 4>([&](auto kc) {
                                constexpr int k = decltype(kc)::value;
                                tie_f64(e[g & 1][k]);
                                pt_acc[g / 2] += e[g & 1][k];
                            });
                        });
                    };
                    if (one_tile) {
                        if (pt_reads && (t0lo != pt_lo || t0hi != pt_hi)) { pt_flush(pt_lo, pt_hi, pt_reads); pt_reads = 0; }
                        pt_lo = t0lo; pt_hi = t0hi;
                        pt_add();
                        pt_reads += SPAN_R;
                    } else {   /* the span a tile ends in: every row a run of its own, its error rates straight from the slot (a header that does not parse: no run) */
                        if (pt_reads) { pt_flush(pt_lo, pt_hi, pt_reads); pt_reads = 0; }
#pragma unroll 1
                        for (uint32_t row = 0; row < SPAN_R; row++) {
                            const uint32_t rlo = __builtin_amdgcn_readfirstlane(l_rows[2 * row]), rhi = __builtin_amdgcn_readfirstlane(l_rows[2 * row + 1]);
                            if ((int32_t)rhi < 0) continue;
                            uint32_t idx = 0;
                            uint32_t lv = (uint32_t)lane;
                            asm volatile("" : "+v"(lv));
                            if (lv == 0) idx = atomicAdd((unsigned int *)par(PAR_NRUNS), 1u);
                            idx = __builtin_amdgcn_readfirstlane(idx);
                            if (idx >= P.pt_runs_cap) { pt_full = true; break; }
                            if (lv == 0) {
                                PtRun run;
                                run.tile = (long long)(((unsigned long long)rhi << 32) | rlo);
                                run.reads = 1;
                                run.pad = 0;
                                ((PtRun *)par(PAR_RUNS))[idx] = run;
                            }
                            const uint32_t qrow = sa + row * ROWB + PRE + QOFF;
                            double *sums = (double *)par(PAR_SUMS);
                            for (uint32_t pos = lv; pos < U; pos += 64)
                                sums[(uint64_t)idx * U + pos] = lds_f64(lds_u8(qrow + pos) << 3);
                        }
                    }
                }
            }
            if (ADr && __builtin_amdgcn_ballot_w64(rec != 0 || multi)) { /* update_adapter_count_array, :2643-2672 */
                any_hit = true;
                auto matches = [&](uint32_t row, uint32_t pos) { /* the adapters that end in that row of the automaton: on the step's second character (pos), on its first */
#pragma unroll
                    for (uint32_t back = 0; back < 2; back++) {
                        unsigned long long hits = l_out[2 * row + back];
                        while (hits) {
                            const int a = __ffsll((long long)hits) - 1;
                            hits &= hits - 1;
                            lds_min(lds_addr(l_first + q * n_ad + a), pos - back);
                        }
                    }
                };
                if (__builtin_amdgcn_ballot_w64(multi)) {
                    uint32_t s2 = st0;
#pragma unroll 1
                    for (uint32_t tt = 0; tt < Q4; tt++) {
                        const uint32_t dw = Q4 * c + tt;
                        uint32_t cl2 = lds_u32(seq_row + 4 * dw);
                        cl2 = dw < DW ? cl2 : CLS6_PAD4;
                        const uint32_t pc2 = __umul24(cl2 >> 8, 6u) + cl2;
#pragma unroll 1
                        for (uint32_t j = 0; j < 2; j++) {
                            s2 = lds_u16(s2 + ((pc2 >> (16 * j)) & 0xFFu));
                            if (s2 >= dfa_hit) matches(span_dfa_state(s2 - dfa_root), 4 * dw + 2 * j + 1);
                        }
                    }
                } else if (rec) {
                    matches(rec & 0xFFFu, (rec >> 12) & 0xFFFu);
                    if (rec2) matches(rec2 & 0xFFFu, (rec2 >> 12) & 0xFFFu);
                }
            }
            /* the chain steps the rounds did not carry (at most 8: U <= 32 NW) and the 1-4 qualities
               behind the chains (:2100-2112): all their bytes first, then all their error rates --
               two round trips to LDS instead of two per step; what does not exist reads the
               padding entry of the table, +0.0 */
            if constexpr (DQ && !LONG) {
                uint32_t lb[8], tb[4];
                /* (three LDS reads instead of twelve: the eight steps are two groups of 16 bytes, lane c's byte of each dword; the
                   1-4 tail qualities lie in the one dword at Lmain, a multiple of four) */
                const sq_u32x4v g0 = *(SQ_LDS const sq_u32x4v *)(uintptr_t)(qual_row + 16 * KRG), g1 = *(SQ_LDS const sq_u32x4v *)(uintptr_t)(qual_row + 16 * (KRG + 1));
                const uint32_t tw = lds_u32(qual_row + Lmain);
                const uint32_t gw[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
                for (uint32_t j = 0; j < 8; j += 2) {
                    uint32_t a0, a1;
                    chain_pair8(gw[j], gw[j + 1], a0, a1);
                    lb[j] = KR4 + j < nsteps ? a0 : SPAN_ERR_PAD << 3;
                    lb[j + 1] = KR4 + j + 1 < nsteps ? a1 : SPAN_ERR_PAD << 3;
                }
#pragma unroll
                for (uint32_t j = 0; j < 4; j++) tb[j] = Lmain + j < U ? ((tw >> (8 * j)) & 0xFFu) << 3 : SPAN_ERR_PAD << 3;
                double le[8], te[4];
#pragma unroll
                for (uint32_t j = 0; j < 8; j++) le[j] = lds_f64(lb[j]);
#pragma unroll
                for (uint32_t j = 0; j < 4; j++) te[j] = lds_f64(tb[j]);
#pragma unroll
                for (uint32_t j = 0; j < 8; j++) acc += le[j];
                tail0 = te[0]; tail1 = te[1]; tail2 = te[2]; tail3 = te[3];
            }
        }
        SPAN_PHASE(2);   /* matches; the chain steps and qualities behind the rounds */
        double total = 0.0;
        if constexpr (DQ && !LONG) {
            total = ((acc + quad_bcast_f64<0x55>(acc)) + quad_bcast_f64<0xAA>(acc)) + quad_bcast_f64<0xFF>(acc); /* :2098-2099 (lane c = 0) */
            total += tail0; total += tail1; total += tail2; total += tail3; /* :2100-2112 */
        }
        uint32_t gn = 0;   /* the read's G/C bases | its bases that are none of A C G T << 16 */
        if constexpr (DS && !LONG) {
            gn = gcn_sum(facc);
            gn += quad_bcast<0xB1>(gn); /* quad_perm [1,0,3,2] */
            gn += quad_bcast<0x4E>(gn); /* quad_perm [2,3,0,1] */
        }
        if (!LONG && c == 0 && q < nv) {
            if constexpr (DS) {
                const uint32_t gc_cnt = gn & 0xFFFFu, acgt_cnt = U - (gn >> 16);
                if (acgt_cnt > 0) atomicAdd(&l_gc[gc_percent(gc_cnt, acgt_cnt)], 1u);
            }
            if constexpr (DQ) {
                {   /* :2126; a store hipcc does not count either */
                    double *dst = &P.metas[r].accumulated_error_rate;
                    asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(dst), "v"(total) : "memory");
                }
                if (total != total) atomicMin(PAIR == 2 ? (unsigned long long *)par(PAR_QC_BAD) : P.qc_first_bad, (unsigned long long)(P.first_read_index + r));
                /* :2127-2137: the largest i with avg <= thresholds[i] (the table falls with i), avg = total / U.
                   l_thr holds the thresholds of the SUM for this U (total <= l_thr[i] exactly when total / U
                   <= thresholds[i]: no f64 division).  A hardware log2 names a candidate, the three thresholds
                   around it (one round trip to LDS) decide; the bisection of the other kernels only when
                   they do not (NaN: bin 0) */
                uint32_t Ul = U;
                if constexpr (PT) asm volatile("" : "+s"(Ul));   /* made here, not kept across the spans (the builds that carry PerTileQuality have no register to spare) */
                const float lg = __builtin_amdgcn_logf((float)total) - __builtin_amdgcn_logf((float)Ul);   /* log2 of the average */
                const int guess = (int)floorf(-3.0103f * lg);
                const uint32_t b0 = (uint32_t)min(max(guess, 1), 92);
                const double t_lo = l_thr[b0 - 1], t_mid = l_thr[b0], t_hi = l_thr[b0 + 1];
                uint32_t lo;
                if (total <= t_lo && !(total <= t_mid)) lo = b0 - 1;
                else if (total <= t_mid && !(total <= t_hi)) lo = b0;
                else if (total <= t_hi && (b0 + 1 == 93 || !(total <= l_thr[b0 + 2]))) lo = b0 + 1;
                else {
                    uint32_t hi = 93;
                    lo = 0;
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi + 1) >> 1;
                        if (total <= l_thr[mid]) lo = mid; else hi = mid - 1;
                    }
                }
                atomicAdd(&l_ps[lo], 1u);
            }
        }
        SPAN_PHASE(3);   /* per read: bins, the error rate */
        if (ADr && __builtin_amdgcn_ballot_w64(any_hit)) { /* update_adapter_count_array, :2643-2672 */
            if constexpr (LONG) {   /* the rows' records, where issue() keeps the row offsets between two calls */
                if (c == 0) l_rows[q] = rec_cur;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            for (uint32_t i = lane; i < SPAN_R * n_ad; i += 64) {
                const uint32_t v = l_first[i];
                if (v == 0xFFFFFFFFu) continue;
                l_first[i] = 0xFFFFFFFFu;
                const uint32_t a = i % n_ad, start = v - l_adlen[a] + 1;
#ifdef SQ_SPAN_DEBUG
                if (start >= U) { printf("k_span: span %llu row %u adapter %u pos %u start %u\n", (unsigned long long)s, i / n_ad, a, v, start); continue; }
#endif
                if constexpr (LONG) {   /* a candidate for the read's first occurrence of the adapter (k_adapter_first) */
                    const uint32_t record = l_rows[i / n_ad];   /* put there below; `start` may lie in front of the segment (mod 2^32) */
                    atomicMin(&P.long_first[(uint64_t)record * n_ad + a], pos_base + start);
                } else if (P.ad_lds) {
                    atomicAdd(&l_adf[a * hs + start], 1u);
                } else {
                    atomicAdd(&P.ad_fwd[a * P.ad_cap + start], 1ULL);
                    atomicAdd(&P.ad_rev[a * P.ad_cap + (U - 1 - start)], 1ULL);
                }
            }
        }
        if constexpr (DS) {
            if (++since_flush == 7) { flush_counts(); since_flush = 0; }
        }
        SPAN_PHASE(4);   /* first hits to the tables, base counts to LDS */
    };

    uint32_t spans_done = 0;   /* spans this wave has started, over all stretches */
    /* scalars: nothing of the tie between the waves of a pair may cost a vector register (the builds of 128 have none to spare) */
    const uint32_t prog_mine = __builtin_amdgcn_readfirstlane(lds_addr(l_prog) + 4 * (uint32_t)wave);
    const uint32_t prog_partner = __builtin_amdgcn_readfirstlane(lds_addr(l_prog) + 4 * ((uint32_t)wave ^ 1));
    /* SEG: the workgroup's stretch of the launch's spans, one length after the other */
    uint64_t c_lo = 0, c_hi = 0;
    uint32_t seg_i = 0;
    if constexpr (SEG) {
        const uint64_t chunk = ((uint64_t)P.span_total + gridDim.x - 1) / gridDim.x;
        c_lo = min((uint64_t)P.span_total, blockIdx.x * chunk);
        c_hi = min((uint64_t)P.span_total, c_lo + chunk);
        if (P.span_bounds) { c_lo = P.span_bounds[blockIdx.x]; c_hi = P.span_bounds[blockIdx.x + 1]; }   /* shares of equal cost (sq_span_launch_long) */
        while (seg_i < P.span_nsegs && (uint64_t)P.span_segs[seg_i].span0 + P.span_segs[seg_i].nspans <= c_lo) seg_i++;
    }
    for (;;) {
    uint32_t fill = 0;   /* SEG: filler rows of this length that this workgroup counts */
    if constexpr (SEG) {
        if (seg_i >= P.span_nsegs) break;
        const SpanSeg g = P.span_segs[seg_i];
        if (g.span0 >= c_hi) break;
        U = g.U;
        Lmain = 4 * ((U - 1) / 4);
        nsteps = Lmain / 4;
        s = max((uint64_t)g.span0, c_lo) + my_seq;
        s_end = min((uint64_t)g.span0 + g.nspans, c_hi);
        s_last = (uint64_t)g.span0 + g.nspans - 1;
        last_rows = g.last_rows;
        seg_first = g.first;
        seg_span0 = g.span0;
        if constexpr (LONG) pos_base = g.pos_base;
        if constexpr (!LONG) {   /* the bins of the per-read average: thresholds of this length (everybody is behind the barrier of the stretch before) */
            for (int i = tid; i < 96; i += T) l_thr[i] = P.thr_sum[U * 96 + i];
            __syncthreads();
        }
        if (s_last >= c_lo && s_last < s_end) fill = SPAN_R - last_rows;
        if constexpr (SPLIT) role = (uint32_t)wave & 1;
    }
    cur = 0;
    rec_cur = 0;
    if (s < s_end) {
        issue_meta(s, meta_base);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        issue(slot_base, meta_base, role, s);
        rec_cur = rec_next;
        urow_cur = urow_next;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   /* the metas have been read: the next ones may land on them */
        if (s + stride < s_end) issue_meta(s + stride, meta_base);
    }
    while (s < s_end) {
        /* the span in slot `cur` has landed, and so have the metas of the one after it */
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if constexpr (PT) {   /* the header bytes of this span have arrived too */
            asm volatile("" : "+v"(name_next), "+v"(nlen_next));
            name_cur = name_next;
            nlen_cur = nlen_next;
        }
        if constexpr (SPLIT) {
            /* The two waves of a pair fetch the two streams of the same records: nothing else ties them, and
               once they have drifted apart the line a record's sequence ends and its qualities begin in (and the
               span's metas) comes from memory twice -- 1.69 x the algorithmic bytes instead of 1.2 x.  A wave
               does not start span number k of its sequence before its partner has started number k - 1. */
            spans_done++;
            *(volatile SQ_LDS uint32_t *)(uintptr_t)prog_mine = spans_done;   /* every lane the same word */
            while (*(volatile SQ_LDS uint32_t *)(uintptr_t)prog_partner + 1 < spans_done) __builtin_amdgcn_s_sleep(4);
        }
        if constexpr (PAIR == 2) {
            __builtin_amdgcn_sched_barrier(0);
            /* calculate_insert_size (:5667-5707) for the span's 16 pairs, as k_isz_span (sq_span.hip) does it: lane c of a
               quad makes one of the four needle halves (the reverse complements of read 2's first and last 16 bases, 8
               bases each) and scans the windows [8 NW c, 8 NW (c + 1)) of read 1 -- raw bytes, the class pass has not
               touched the slot yet -- for places where a needle half's low dword matches; the rare candidates are looked
               at in window order, the quad's first match is the pair's.  Before the next span is requested: its ends land
               where these lie. */
            constexpr uint32_t WQ = 8 * NW;
            const uint32_t UP4 = 0xDFDFDFDFu, last = U - 16, L2 = P.pair_L2;
            const uint32_t ea = ends_addr + 32 * q + ((c & 1) ? 0 : 8) + ((c & 2) ? 16 : 0);
            const unsigned long long mine = pair_revcomp8(*(SQ_LDS const unsigned long long *)(uintptr_t)ea);
            const uint32_t m_lo = (uint32_t)mine, m_hi = (uint32_t)(mine >> 32);
            const uint32_t hl = quad_bcast<0x00>(m_lo), hl2 = quad_bcast<0x00>(m_hi), hh = quad_bcast<0x55>(m_lo), hh2 = quad_bcast<0x55>(m_hi);
            const uint32_t tl = quad_bcast<0xAA>(m_lo), tl2 = quad_bcast<0xAA>(m_hi), th = quad_bcast<0xFF>(m_lo), th2 = quad_bcast<0xFF>(m_hi);
            const uint32_t ra = slot_base + cur * SLOT + q * ROWB + PRE + WQ * c;
            unsigned long long cand = pair_scan_candidates<NW>(ra, hl, hl2, hh, hh2, tl, tl2, th, th2);   /* bit 63 - k: position WQ c + k */
            uint32_t result = 0;
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
                    const uint32_t d = pair_nonzero_bytes_of(b[0] ^ hl) + pair_nonzero_bytes_of(b[1] ^ hl2) +
                                       pair_nonzero_bytes_of(b[2] ^ hh) + pair_nonzero_bytes_of(b[3] ^ hh2);
                    if (d <= 1) { result = i + 16; break; }
                }
                if ((u0 == tl && u1 == tl2) || (u2 == th && u3 == th2)) {
                    const uint32_t d = pair_nonzero_bytes_of(b[0] ^ tl) + pair_nonzero_bytes_of(b[1] ^ tl2) +
                                       pair_nonzero_bytes_of(b[2] ^ th) + pair_nonzero_bytes_of(b[3] ^ th2);
                    if (d <= 1) { result = i + L2; break; }
                }
            }
            /* the first match in window order: the lowest quarter that has one */
            const uint32_t r0 = quad_bcast<0x00>(result), r1 = quad_bcast<0x55>(result), r2 = quad_bcast<0xAA>(result), r3 = quad_bcast<0xFF>(result);
            result = r0 ? r0 : r1 ? r1 : r2 ? r2 : r3;
            if (c == 0) {
                uint32_t *dst = (uint32_t *)par(PAR_RESULTS) + s * SPAN_R + q;
                asm volatile("global_store_dword %0, %1, off" :: "v"(dst), "v"(result) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   /* the ends have been read: the next span's may land on them */
            __builtin_amdgcn_sched_barrier(0);
        }
        if (s + stride < s_end) {
            issue(slot_base + (cur ^ 1) * SLOT, meta_base, role ^ (SPLIT ? 1u : 0u), s + stride);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   /* the metas have been read: the next ones may land on them */
            if (s + 2 * stride < s_end) issue_meta(s + 2 * stride, meta_base);
        }
        if constexpr (!SPLIT) body(std::true_type{}, std::true_type{});
        else if (role == 0) body(std::true_type{}, std::false_type{});
        else body(std::false_type{}, std::true_type{});
        cur ^= 1;
        s += stride;
        rec_cur = rec_next;
        urow_cur = urow_next;
        if constexpr (SPLIT) role ^= 1;
    }
    if constexpr (!SEG) break;
    /* the length changes: what the workgroup counted goes to the device tables */
    flush_counts();
    since_flush = 0;
    __syncthreads();
    merge_hist(true, fill);
    __syncthreads();
    seg_i++;
    }
    if constexpr (PT) {
        if (pt_reads) pt_flush(pt_lo, pt_hi, pt_reads);
    }
    if constexpr (!SEG) {
        flush_counts();
        __syncthreads();
        merge_hist(false, 0);
    }
    int tid_end = tid;
    if constexpr (PT) { tid_end = (int)threadIdx.x; asm volatile("" : "+v"(tid_end)); }
    for (uint32_t i = tid_end; i < 101; i += T)
        if (l_gc[i]) atomicAdd(&(PAIR == 2 ? (unsigned long long *)par(PAR_QC_GC) : P.qc_gc)[i], (unsigned long long)l_gc[i]);
    for (uint32_t i = tid_end; i < 94; i += T)
        if (l_ps[i]) atomicAdd(&(PAIR == 2 ? (unsigned long long *)par(PAR_QC_PS) : P.qc_ps)[i], (unsigned long long)l_ps[i]);
}

} // namespace

#endif
