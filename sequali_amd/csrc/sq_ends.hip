/*
 * sq_ends.hip -- the modules that only look at the ends of a read:
 * OverrepresentedSequences, DedupEstimator, InsertSizeMetrics.
 *
 * Shape shared by the three: a data-parallel kernel over the records does
 * everything that is independent per read (2-bit canonical k-mers + Wang
 * hash + per-read de-dup; fingerprint + MurmurHash3; the mismatch-tolerant
 * overlap scan), and whatever the reference makes order dependent (first-come
 * caps, the estimator's rebuild quirk) is resolved exactly afterwards on the
 * small ordered remainder.
 */
#include <hipcub/hipcub.hpp>
#include "sq_span.h"

#include <algorithm>
#include <thread>
#include <cmath>

#include "sq_common.h"

namespace {

/* ---- hashes -------------------------------------------------------------------- */
/* wanghash.h:14-26 */
__host__ __device__ inline uint64_t wanghash64(uint64_t k)
{
    k = (~k) + (k << 21);
    k ^= k >> 24;
    k = k + (k << 3) + (k << 8);
    k ^= k >> 14;
    k = k + (k << 2) + (k << 4);
    k ^= k >> 28;
    k += k << 31;
    return k;
}

/* wanghash.h:28-63 */
inline uint64_t wanghash64_inverse(uint64_t k)
{
    uint64_t t;
    t = k - (k << 31); k = k - (t << 31);
    t = k ^ (k >> 28); k = k ^ (t >> 28);
    k *= 14933078535860113213ULL;
    t = k ^ (k >> 14); t = k ^ (t >> 14); t = k ^ (t >> 14); k = k ^ (t >> 14);
    k *= 15244667743933553977ULL;
    t = k ^ (k >> 24); k = k ^ (t >> 24);
    t = ~k; t = ~(k - (t << 21)); t = ~(k - (t << 21)); k = ~(k - (t << 21));
    return k;
}

__host__ __device__ inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
__host__ __device__ inline uint64_t fmix64(uint64_t k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}

/* MurmurHash3_x64_64, murmur3.h:47-158, over bytes produced by `get(i)` */
template <typename Get>
__host__ __device__ inline uint64_t murmur3_x64_64(Get get, uint64_t len, uint64_t seed)
{
    const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
    uint64_t h1 = seed, h2 = seed;
    const uint64_t nblocks = len / 16;
    for (uint64_t b = 0; b < nblocks; b++) {
        uint64_t k1 = 0, k2 = 0;
        for (int j = 0; j < 8; j++) {
            k1 |= (uint64_t)get(b * 16 + j) << (8 * j);
            k2 |= (uint64_t)get(b * 16 + 8 + j) << (8 * j);
        }
        k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
        h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
        k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
        h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
    }
    const uint64_t t0 = nblocks * 16, rem = len & 15;
    uint64_t k1 = 0, k2 = 0;
    for (uint64_t j = 8; j < rem; j++) k2 ^= (uint64_t)get(t0 + j) << (8 * (j - 8));
    if (rem > 8) { k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2; }
    for (uint64_t j = 0; j < (rem < 8 ? rem : 8); j++) k1 ^= (uint64_t)get(t0 + j) << (8 * j);
    if (rem > 0) { k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1; }
    h1 ^= len; h2 ^= len;
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    h1 += h2; h2 += h1;
    return h2;
}

/* the same hash over at most 32 bytes held in four words (little endian, zero padded) */
__host__ __device__ inline uint64_t murmur3_x64_64_w4(uint64_t w0, uint64_t w1, uint64_t w2, uint64_t w3,
                                                      uint64_t len, uint64_t seed)
{
    const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
    uint64_t h1 = seed, h2 = seed;
    uint64_t t1 = w0, t2 = w1; /* the words the tail starts at */
    if (len >= 16) {
        uint64_t k1 = w0, k2 = w1;
        k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
        h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
        k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
        h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
        t1 = w2; t2 = w3;
    }
    if (len >= 32) { /* two whole blocks: no tail */
        uint64_t k1 = w2, k2 = w3;
        k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
        h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
        k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
        h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
    }
    const uint64_t rem = len & 15;
    if (rem > 8) {
        uint64_t k2 = t2 & (rem >= 16 ? ~0ULL : ((1ULL << (8 * (rem - 8))) - 1));
        k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
    }
    if (rem > 0) {
        uint64_t k1 = rem >= 8 ? t1 : (t1 & ((1ULL << (8 * rem)) - 1));
        k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
    }
    h1 ^= len; h2 ^= len;
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    h1 += h2; h2 += h1;
    return h2;
}

/* ================================ OverrepresentedSequences ======================== */

/* sequence_to_canonical_kmer, _qcmodule.c:3657-3694.  >= 0: k-mer; -1: a byte
 * outside ACGTN; -2: N present */
__device__ long long canonical_kmer(const uint8_t *s, uint32_t k, const uint8_t *buf_end)
{
    uint64_t kmer = 0;
    bool has_n = false, has_other = false;
    for (uint32_t i0 = 0; i0 < k; i0 += 8) { /* eight bases per load */
        uint64_t w;
        if (s + i0 + 8 <= buf_end) {
            w = sq_load_u64_unaligned(s + i0);
        } else {
            w = 0;
            for (uint32_t b = 0; b < 8 && s + i0 + b < buf_end; b++) w |= (uint64_t)s[i0 + b] << (8 * b);
        }
        const uint32_t nb = min(8u, k - i0);
        for (uint32_t b = 0; b < nb; b++) {
            const unsigned c = (unsigned)(w >> (8 * b)) & 0xFFu, cls = sq_base_class(c);
            if (cls == 4) {
                if ((c | 0x20u) == 'n') has_n = true; else has_other = true;
            }
            kmer = (kmer << 2) | (cls & 3);
        }
    }
    if (has_other) return -1;
    if (has_n) return -2;
    /* reverse_complement_kmer :3634-3655 */
    uint64_t x = ~kmer;
    x = (x << 32) | (x >> 32);
    x = ((x & 0xFFFF0000FFFF0000ULL) >> 16) | ((x & 0x0000FFFF0000FFFFULL) << 16);
    x = ((x & 0xFF00FF00FF00FF00ULL) >> 8) | ((x & 0x00FF00FF00FF00FFULL) << 8);
    x = ((x & 0xF0F0F0F0F0F0F0F0ULL) >> 4) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
    x = ((x & 0xCCCCCCCCCCCCCCCCULL) >> 2) | ((x & 0x3333333333333333ULL) << 2);
    const uint64_t rc = x >> (64 - 2 * k);
    return (long long)(rc > kmer ? kmer : rc);
}

enum { OVR_NORMAL = 0, OVR_FULL = 1, OVR_CROSSING = 2, OVR_RANKED = 3 };
constexpr unsigned long long RANK_NONE = ~0ULL;

struct OvrParams {
    const uint8_t *buf;
    uint64_t buf_len;
    const sq_meta *metas;
    uint64_t n;               /* records in the batch */
    uint64_t first_sample;    /* record index of the first sampled record */
    uint64_t n_samples;       /* sampled records in this launch */
    uint64_t sample_base;     /* samples of this batch in front of this launch */
    uint64_t rank_base;       /* sample number of the launch's first sample in the rank order */
    uint64_t record_base;     /* records seen before this batch */
    uint32_t k, sample_every;
    long long frags_start, frags_end;
    int mode;
    unsigned long long *hashes; /* open addressing, 0 = empty */
    unsigned int *counts;
    unsigned long long *ranks;  /* crossing mode only */
    uint64_t table_mask;
    unsigned long long *n_unique, *total_fragments, *warn_count;
    long long *warn_last;
    unsigned long long *big_staging; /* [n_samples][big_size] for reads with > 21 fragments */
    uint64_t big_size;
    const uint32_t *occupied;        /* closed table: bit s = slot s holds a key (k_ovr_occupied); NULL: ask the table */
};

/* Sequence_duplication_insert_hash, _qcmodule.c:3542-3568, concurrent form */
__device__ void ovr_insert(const OvrParams &P, unsigned long long h, unsigned long long rank,
                           unsigned long long &new_keys, unsigned int count = 1)
{
    uint64_t i = h & P.table_mask;
    if (P.mode == OVR_FULL) {   /* the table is closed (:3553): plain loads, see k_overrep_par */
        const unsigned long long *keys = P.hashes;
        for (;;) {
            const unsigned long long cur = keys[i];
            if (cur == 0) return;   /* new keys are dropped */
            if (cur == h) {
                if (((const unsigned int *)P.counts)[i] != 0) atomicAdd(&P.counts[i], count);   /* entries removed by the cap keep their key with a zero count */
                return;
            }
            i = (i + 1) & P.table_mask;
        }
    }
    for (;;) {
        unsigned long long cur = __hip_atomic_load(&P.hashes[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == 0) {
            if (P.mode == OVR_FULL) return; /* table is closed: new keys are dropped (:3553) */
            cur = atomicCAS(&P.hashes[i], 0ULL, h);
            if (cur == 0) {
                new_keys++; /* added to P.n_unique once per wave */
                cur = h;
            }
        }
        if (cur == h) {
            if (P.mode == OVR_FULL) {
                /* entries removed by the cap keep their key with a zero count */
                if (__hip_atomic_load(&P.counts[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
                    atomicAdd(&P.counts[i], count);
            } else {
                atomicAdd(&P.counts[i], count);
                /* keys from earlier launches carry rank 0 and stay there */
                if (P.mode >= OVR_CROSSING) atomicMin(&P.ranks[i], rank + 1);
            }
            return;
        }
        i = (i + 1) & P.table_mask;
    }
}

/* OverrepresentedSequences_add_meta, _qcmodule.c:3829-3942: one lane per sampled record */
/* Overrepresented fragments are, by definition, the same hash over and over (poly-G tails,
 * adapters): a workgroup counts fragments in a small LDS table first (write-once entries:
 * hash, count, earliest rank) and brings each entry to the device table once. */
constexpr uint32_t OVR_CACHE = 128;

__global__ void k_overrep(OvrParams P)
{
    __shared__ unsigned long long c_hash[OVR_CACHE], c_rank[OVR_CACHE];
    __shared__ unsigned int c_count[OVR_CACHE];
    for (uint32_t i = threadIdx.x; i < OVR_CACHE; i += blockDim.x) { c_hash[i] = 0; c_rank[i] = ~0ULL; c_count[i] = 0; }
    __syncthreads();
    unsigned long long local_frags = 0, new_keys = 0;
    for (uint64_t s = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; s < P.n_samples;
         s += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = P.first_sample + (P.sample_base + s) * P.sample_every;
        const sq_meta m = P.metas[r];
        const long long L = m.sequence_length, k = P.k;
        if (L < k) continue; /* still counted as sampled (:3837-3844) */
        const uint8_t *seq = P.buf + m.record_start + m.sequence_offset;
        const long long max_frag = (L + k - 1) / k;
        const long long from_mid = max_frag / 2;
        long long n_start = max_frag - from_mid, n_end = from_mid;
        if (P.frags_start < n_start) n_start = P.frags_start;
        if (P.frags_end < n_end) n_end = P.frags_end;
        const long long total = n_start + n_end;
        if (total == 0) continue;
        /* staging table of 2^ceil(log2(1.5 total)) slots (:3884): smallest power of
           two >= 1.5 * total, i.e. >= ceil(3 total / 2) */
        uint64_t size = 1;
        while (2 * size < 3 * (uint64_t)total) size <<= 1;
        unsigned long long small[32];
        unsigned long long *stage = small;
        if (size > 32) stage = P.big_staging + s * P.big_size;
        for (uint64_t i = 0; i < size; i++) stage[i] = 0;
        bool warn = false;
        unsigned long long valid = 0;
        for (long long f = 0; f < total; f++) {
            const long long at = f < n_start ? f * k : L - n_end * k + (f - n_start) * k;
            const long long km = canonical_kmer(seq + at, (uint32_t)k, P.buf + P.buf_len);
            if (km < 0) {
                if (km == -1) warn = true;
                continue;
            }
            valid++;
            const unsigned long long h = wanghash64((uint64_t)km);
            if (h == 0) continue; /* indistinguishable from an empty slot (:3597) */
            uint64_t i = h & (size - 1); /* add_to_staging :3588-3608 */
            for (;;) {
                if (stage[i] == 0) { stage[i] = h; break; }
                if (stage[i] == h) break;
                i = (i + 1) & (size - 1);
            }
        }
        for (uint64_t i = 0; i < size; i++) { /* flushed in slot order (:3925-3930) */
            const unsigned long long h = stage[i];
            if (!h) continue;
            const unsigned long long rank = ((P.rank_base + s) << 24) | i;
            const uint32_t e = (uint32_t)(h >> 24) & (OVR_CACHE - 1);
            const unsigned long long cur = atomicCAS(&c_hash[e], 0ULL, h);
            if (cur == 0 || cur == h) {
                atomicAdd(&c_count[e], 1u);
                atomicMin(&c_rank[e], rank);
            } else {
                ovr_insert(P, h, rank, new_keys);
            }
        }
        local_frags += valid;
        if (warn) {
            atomicAdd(P.warn_count, 1ULL);
            atomicMax(P.warn_last, (long long)(P.record_base + r));
        }
    }
    __syncthreads();
    if (threadIdx.x < OVR_CACHE && c_hash[threadIdx.x])
        ovr_insert(P, c_hash[threadIdx.x], c_rank[threadIdx.x], new_keys, c_count[threadIdx.x]);
    /* one atomic per wave, not per lane: they all go to the same address */
    for (int off = 32; off > 0; off >>= 1) {
        local_frags += __shfl_xor(local_frags, off);
        new_keys += __shfl_xor(new_keys, off);
    }
    if ((threadIdx.x & 63) == 0) {
        if (local_frags) atomicAdd(P.total_fragments, local_frags);
        if (new_keys) atomicAdd(P.n_unique, new_keys);
    }
}

/* The pass over a CLOSED table -- the state a run is in from its first few million reads on (:3553: new keys are dropped,
 * what is found is counted) -- with a lane's memory requests IN FLIGHT TOGETHER (round 6).  k_overrep above is a chain of
 * round trips per lane: a fragment's three loads, its walk through a staging table that lives in scratch memory, its probe of
 * the device table, then the next fragment (profiles/r6/pmc_k_overrep.txt: the waves wait three quarters of their time with
 * 1.3 vector-memory instructions in flight per SIMD).  Here a lane asks for the bytes of ALL its fragments at once (up to
 * OVR_PAR_F of up to 24 bases: the defaults are 10 of 21), makes their hashes in registers, drops a fragment that a fragment
 * in front of it repeats (add_to_staging counts it once, :3588-3608; the staging table's slot ORDER only decides ranks, and a
 * closed table has none to give), asks a bitmap of the table's occupied slots (2 MB, L2-resident: seven probes in ten end
 * there) and walks the probe runs of what is left together, a slot of every fragment per round.  What is found -- the few
 * fragments that come again and again: a poly-G tail, an adapter, a million arrivals of ONE table slot per 100 M reads -- is
 * counted per workgroup in LDS, by table slot, and reaches the table once per workgroup.
 * Kept SMALL on purpose: the first version also carried the open table's modes and was 96 KB of code, more than the
 * instruction cache two CUs share holds -- every change to it measured the same 2.8 ms per 25 M reads
 * (profiles/r6/exp_overrep.txt). */
constexpr int OVR_PAR_F = 10;
/* 8 bases -> 16 bits (A 0 C 1 G 2 T 3, either case; the first base most significant): (c >> 1) & 3 gives A 0 C 1 G 3 T 2, the
   exclusive or with its own upper bit puts G and T right; three folds bring the eight fields together */
__device__ __forceinline__ uint32_t ovr_pack8(unsigned long long w)
{
    unsigned long long t = (w >> 1) & 0x0303030303030303ULL;
    t = (t ^ (t >> 1)) & 0x0303030303030303ULL;
    t = ((t << 2) | (t >> 8)) & 0x000F000F000F000FULL;
    t = ((t << 4) | (t >> 16)) & 0x000000FF000000FFULL;
    return (uint32_t)(((t << 8) | (t >> 32)) & 0xFFFFULL);
}
/* bit 7 of every byte of x that is not zero */
__device__ __forceinline__ unsigned long long ovr_nonzero_bytes(unsigned long long x)
{
    return (((x & 0x7F7F7F7F7F7F7F7FULL) + 0x7F7F7F7F7F7F7F7FULL) | x) & 0x8080808080808080ULL;
}
/* sequence_to_canonical_kmer (:3657-3694) for k <= 24 bases lying in three words: >= 0 the k-mer, -1 a byte outside ACGTN, -2 an N */
__device__ __forceinline__ long long ovr_kmer_of_words(unsigned long long w0, unsigned long long w1, unsigned long long w2, uint32_t k)
{
    const unsigned long long w[3] = {w0, w1, w2};
    unsigned long long not_acgt = 0, is_n = 0;   /* bit 7 of byte j, word i: base 8 i + j of the fragment is none of A C G T / is an N */
    uint64_t kmer = 0;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const unsigned long long u = w[i] & 0xDFDFDFDFDFDFDFDFULL;
        /* the letter a base's two bits stand for, beside the base itself: equal for A C G T in either case, else not */
        unsigned long long t = (w[i] >> 1) & 0x0303030303030303ULL;
        t = (t ^ (t >> 1)) & 0x0303030303030303ULL;
        const uint32_t lo = __builtin_amdgcn_perm(0u, 0x54474341u, (uint32_t)t), hi = __builtin_amdgcn_perm(0u, 0x54474341u, (uint32_t)(t >> 32));
        const unsigned long long letters = ((unsigned long long)hi << 32) | lo;
        const int nb = (int)k - 8 * i;   /* bases of this word that belong to the fragment: its first min(8, nb) bytes */
        const unsigned long long in = nb >= 8 ? ~0ULL : nb <= 0 ? 0ULL : ((1ULL << (8 * nb)) - 1);
        not_acgt |= ovr_nonzero_bytes(u ^ letters) & in;
        is_n |= ~ovr_nonzero_bytes(u ^ 0x4E4E4E4E4E4E4E4EULL) & 0x8080808080808080ULL & in;
        kmer = (kmer << 16) | ovr_pack8(w[i]);
    }
    if (not_acgt & ~is_n) return -1;
    if (is_n) return -2;
    kmer >>= 2 * (24 - k);
    uint64_t x = ~kmer;   /* reverse_complement_kmer :3634-3655 */
    x = (x << 32) | (x >> 32);
    x = ((x & 0xFFFF0000FFFF0000ULL) >> 16) | ((x & 0x0000FFFF0000FFFFULL) << 16);
    x = ((x & 0xFF00FF00FF00FF00ULL) >> 8) | ((x & 0x00FF00FF00FF00FFULL) << 8);
    x = ((x & 0xF0F0F0F0F0F0F0F0ULL) >> 4) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
    x = ((x & 0xCCCCCCCCCCCCCCCCULL) >> 2) | ((x & 0x3333333333333333ULL) << 2);
    const uint64_t rc = x >> (64 - 2 * k);
    return (long long)(rc > kmer ? kmer : rc);
}

__global__ void __launch_bounds__(256) k_overrep_par(OvrParams P)
{
    __shared__ unsigned long long c_slot[OVR_CACHE];   /* table slot + 1 of what the entry counts (0: free) */
    __shared__ unsigned int c_count[OVR_CACHE];
    for (uint32_t i = threadIdx.x; i < OVR_CACHE; i += blockDim.x) { c_slot[i] = 0; c_count[i] = 0; }
    __syncthreads();
    unsigned long long local_frags = 0;
    const unsigned long long *keys = P.hashes;
    const unsigned int *cnts = P.counts;
    for (uint64_t s = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; s < P.n_samples;
         s += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = P.first_sample + (P.sample_base + s) * P.sample_every;
        const sq_meta m = P.metas[r];
        const long long L = m.sequence_length, k = P.k;
        if (L < k) continue; /* still counted as sampled (:3837-3844) */
        const uint8_t *seq = P.buf + m.record_start + m.sequence_offset, *last8 = P.buf + P.buf_len - 8;
        const long long max_frag = (L + k - 1) / k, from_mid = max_frag / 2;
        long long n_start = max_frag - from_mid, n_end = from_mid;
        if (P.frags_start < n_start) n_start = P.frags_start;
        if (P.frags_end < n_end) n_end = P.frags_end;
        const int total = (int)(n_start + n_end);   /* <= OVR_PAR_F: the host sends other batches to k_overrep */
        if (total == 0) continue;
        /* (1) every fragment's bytes: three words each (a word that would reach behind the buffer is taken from its last
           eight bytes and shifted: what lies behind a fragment's k bases is never looked at) */
        unsigned long long w[OVR_PAR_F][3];
#pragma unroll
        for (int f = 0; f < OVR_PAR_F; f++) {
            w[f][0] = w[f][1] = w[f][2] = 0;
            if (f < total) {
                const long long at = f < n_start ? f * k : L - n_end * k + (f - n_start) * k;
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    const uint8_t *p = seq + at + 8 * j;
                    if (8 * j >= k) break;
                    if (p <= last8) w[f][j] = sq_load_u64_unaligned(p);
                    else if (p < last8 + 8) w[f][j] = sq_load_u64_unaligned(last8) >> (8 * (p - last8));
                }
            }
        }
        /* (2) their hashes; a fragment repeated inside the read is counted once */
        unsigned long long h[OVR_PAR_F];
        uint32_t dev = 0;   /* bit f: fragment f asks the table */
        bool warn = false;
        unsigned long long valid = 0;
#pragma unroll
        for (int f = 0; f < OVR_PAR_F; f++) {
            h[f] = 0;
            if (f < total) {
                const long long km = ovr_kmer_of_words(w[f][0], w[f][1], w[f][2], (uint32_t)k);
                if (km == -1) warn = true;
                if (km >= 0) {
                    valid++;
                    h[f] = wanghash64((uint64_t)km);   /* 0: indistinguishable from an empty slot (:3597) */
                    bool dup = h[f] == 0;
#pragma unroll
                    for (int g = 0; g < f; g++) dup |= h[g] == h[f];
                    if (!dup) dev |= 1u << f;
                }
            }
        }
        local_frags += valid;
        if (warn) {
            atomicAdd(P.warn_count, 1ULL);
            atomicMax(P.warn_last, (long long)(P.record_base + r));
        }
        /* (3) the bitmap of occupied slots: PLAIN loads here and below -- nobody writes a key any more, and a count is zero
           (an entry the cap removed) or not for the whole launch, so the XCD's L2 may answer */
        {
            uint32_t occ[OVR_PAR_F];
#pragma unroll
            for (int f = 0; f < OVR_PAR_F; f++) {
                occ[f] = 0;
                if ((dev >> f) & 1u) occ[f] = P.occupied[(h[f] & P.table_mask) >> 5];
            }
#pragma unroll
            for (int f = 0; f < OVR_PAR_F; f++)
                if (((dev >> f) & 1u) && !((occ[f] >> (h[f] & 31u)) & 1u)) dev &= ~(1u << f);
        }
        /* (4) the probe runs, together */
        uint64_t at[OVR_PAR_F];
#pragma unroll
        for (int f = 0; f < OVR_PAR_F; f++) at[f] = h[f] & P.table_mask;
        while (dev) {
            unsigned long long cur[OVR_PAR_F];
#pragma unroll
            for (int f = 0; f < OVR_PAR_F; f++) {
                cur[f] = 0;
                if ((dev >> f) & 1u) cur[f] = keys[at[f]];
            }
#pragma unroll
            for (int f = 0; f < OVR_PAR_F; f++) {
                if (!((dev >> f) & 1u)) continue;
                if (cur[f] != 0 && cur[f] != h[f]) { at[f] = (at[f] + 1) & P.table_mask; continue; }   /* somebody else's: on to the next slot */
                dev &= ~(1u << f);
                const uint64_t i = at[f];
                if (cur[f] == h[f] && cnts[i] != 0) {   /* (entries removed by the cap keep their key with a zero count) */
                    const uint32_t e = (uint32_t)(i * 0x9E3779B1u >> 20) & (OVR_CACHE - 1);
                    const unsigned long long key = i + 1, was = atomicCAS(&c_slot[e], 0ULL, key);
                    if (was == 0 || was == key) atomicAdd(&c_count[e], 1u);
                    else atomicAdd(&P.counts[i], 1u);
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < OVR_CACHE && c_slot[threadIdx.x]) atomicAdd(&P.counts[c_slot[threadIdx.x] - 1], c_count[threadIdx.x]);
    for (int off = 32; off > 0; off >>= 1) local_frags += __shfl_xor(local_frags, off);
    if ((threadIdx.x & 63) == 0 && local_frags) atomicAdd(P.total_fragments, local_frags);
}

/* the closed table's occupied slots as bits (2 MB for the default table's 128 MB of keys: it stays in an XCD's L2) */
__global__ void k_ovr_occupied(const unsigned long long *hashes, uint64_t table_size, uint32_t *bits)
{
    for (uint64_t w = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; w < table_size / 32; w += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t m = 0;
#pragma unroll 8
        for (uint32_t b = 0; b < 32; b++) m |= (hashes[32 * w + b] != 0 ? 1u : 0u) << b;
        bits[w] = m;
    }
}

/* crossing batch, before the launch: keys already in the table rank in front of everything */
__global__ void k_ovr_mark_old(const unsigned long long *hashes, unsigned long long *ranks, uint64_t table_size)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < table_size;
         i += (uint64_t)gridDim.x * blockDim.x)
        ranks[i] = hashes[i] ? 0ULL : RANK_NONE;
}

/* crossing batch: list (rank, slot) of the keys that were new in this batch */
__global__ void k_ovr_collect_new(const unsigned long long *ranks, uint64_t table_size,
                                  unsigned long long *out_rank, unsigned long long *out_slot,
                                  unsigned long long *n_out)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < table_size;
         i += (uint64_t)gridDim.x * blockDim.x) {
        if (ranks[i] != RANK_NONE && ranks[i] != 0) {
            const unsigned long long o = atomicAdd(n_out, 1ULL);
            out_rank[o] = ranks[i];
            out_slot[o] = i;
        }
    }
}

__global__ void k_ovr_kill(unsigned int *counts, const unsigned long long *slots, uint64_t n)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x)
        counts[slots[i]] = 0;
}

/* ---- shard mode (one rank of a multi-GPU job) ------------------------------
 * The table is uncapped and every key keeps the rank of its first occurrence over the
 * whole job, (sampled read number << 24 | staging slot).  The reference's table holds
 * the first max_unique distinct hashes of the job in that order with all their
 * occurrences counted, so the merge is: candidates = every shard's first max_unique
 * keys -> first max_unique distinct of the union -> per-shard counts of those -> sum. */
/* The table closes: the keys the cap removed (count 0) leave it.  They kept their slots while the batch that crossed the cap
 * was being sorted out (a key with a zero count is "seen, not counted"), and the table had been filled up to four fifths for
 * that batch: every probe of the closed table then walked runs of a dozen dead keys (profiles/r6/pmc_k_overrep_compact.txt:
 * 492 vector-memory reads per wave and iteration where ten fragments need ten).  A fragment that meets a dead key and one
 * that meets an empty slot are treated alike (:3553: not in the table, dropped), so the live keys alone, hashed into a fresh
 * table -- three tenths full with the default cap -- give the same counts. */
__global__ void k_ovr_compact(const unsigned long long *oh, const unsigned int *oc, uint64_t size, unsigned long long *nh, unsigned int *nc)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < size; i += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long h = oh[i];
        const unsigned int c = oc[i];
        if (!h || !c) continue;
        uint64_t j = h & (size - 1);
        while (atomicCAS(&nh[j], 0ULL, h) != 0ULL) j = (j + 1) & (size - 1); /* keys are distinct */
        nc[j] = c;
    }
}

__global__ void k_ovr_rehash(const unsigned long long *oh, const unsigned int *oc, const unsigned long long *orank,
                             uint64_t old_size, unsigned long long *nh, unsigned int *nc,
                             unsigned long long *nrank, uint64_t new_mask)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < old_size;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long h = oh[i];
        if (!h) continue;
        uint64_t j = h & new_mask;
        while (atomicCAS(&nh[j], 0ULL, h) != 0ULL) j = (j + 1) & new_mask; /* keys are distinct */
        nc[j] = oc[i];
        if (nrank) nrank[j] = orank[i];
    }
}

__global__ void k_ovr_collect_all(const unsigned long long *hashes, const unsigned long long *ranks,
                                  uint64_t table_size, unsigned long long *out_rank,
                                  unsigned long long *out_hash, unsigned long long *n_out)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < table_size;
         i += (uint64_t)gridDim.x * blockDim.x) {
        if (hashes[i]) {
            const unsigned long long o = atomicAdd(n_out, 1ULL);
            out_rank[o] = ranks[i];
            out_hash[o] = hashes[i];
        }
    }
}

/* position of the first occurrence of every hash of a rank-sorted list */
__global__ void k_ovr_first_pos(const unsigned long long *hashes, uint64_t n, unsigned long long *keys,
                                unsigned long long *pos, uint64_t mask)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long h = hashes[i];
        uint64_t j = h & mask;
        for (;;) {
            unsigned long long cur = atomicCAS(&keys[j], 0ULL, h);
            if (cur == 0 || cur == h) { atomicMin(&pos[j], (unsigned long long)i); break; }
            j = (j + 1) & mask;
        }
    }
}

__global__ void k_ovr_flag_first(const unsigned long long *hashes, uint64_t n, const unsigned long long *keys,
                                 const unsigned long long *pos, uint64_t mask, unsigned char *flags)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long h = hashes[i];
        uint64_t j = h & mask;
        while (keys[j] != h) j = (j + 1) & mask;
        flags[i] = pos[j] == i;
    }
}

__global__ void k_ovr_lookup(const unsigned long long *table, const unsigned int *counts, uint64_t mask,
                             const unsigned long long *hashes, uint64_t n, unsigned long long *out)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long h = hashes[i];
        uint64_t j = h & mask;
        unsigned long long c = 0;
        for (;;) {
            const unsigned long long cur = table[j];
            if (cur == h) { c = counts[j]; break; }
            if (cur == 0) break;
            j = (j + 1) & mask;
        }
        out[i] = c;
    }
}

__global__ void k_ovr_install(unsigned long long *table, unsigned int *counts, uint64_t mask,
                              const unsigned long long *hashes, const unsigned long long *in_counts, uint64_t n)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long h = hashes[i];
        uint64_t j = h & mask;
        while (atomicCAS(&table[j], 0ULL, h) != 0ULL) j = (j + 1) & mask;
        counts[j] = (unsigned int)in_counts[i]; /* the reference's counts are u32 as well (:3452) */
    }
}

/* ================================ DedupEstimator ================================ */

struct DedupParams {
    const uint8_t *buf1, *buf2;
    const sq_meta *metas1, *metas2; /* metas2 == nullptr: single end */
    uint64_t n;
    uint64_t front_len, back_len, front_off, back_off;
    unsigned long long *hashes;     /* [n] */
    unsigned char *special;         /* [n] 1: the host must hash this pair (stale bytes) */
};

/* DedupEstimator_add_sequence_ptr :4462-4485 / _add_sequence_pair_ptr :4487-4517 */
__global__ void k_dedup_hash(DedupParams P)
{
    for (uint64_t r = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; r < P.n;
         r += (uint64_t)gridDim.x * blockDim.x) {
        const sq_meta m1 = P.metas1[r];
        const uint8_t *s1 = P.buf1 + m1.record_start + m1.sequence_offset;
        const uint64_t L1 = m1.sequence_length;
        const uint64_t fp_len = P.front_len + P.back_len;
        unsigned long long h;
        unsigned char special = 0;
        if (!P.metas2) {
            if (L1 <= fp_len) {
                h = murmur3_x64_64([&](uint64_t i) { return s1[i]; }, L1, 0);
            } else {
                const uint64_t rem = L1 - fp_len;
                const uint64_t fo = rem / 2 < P.front_off ? rem / 2 : P.front_off;
                const uint64_t bo = rem / 2 < P.back_off ? rem / 2 : P.back_off;
                const uint8_t *front = s1 + fo, *back = s1 + L1 - (bo + P.back_len);
                const uint64_t fl = P.front_len;
                if (fl == 8 && P.back_len == 8) /* the default fingerprint: two 8-byte loads instead of 16 byte loads */
                    h = murmur3_x64_64_w4(sq_load_u64_unaligned(front), sq_load_u64_unaligned(back), 0, 0, 16, L1 >> 6);
                else
                    h = murmur3_x64_64([&](uint64_t i) { return i < fl ? front[i] : back[i - fl]; },
                                       fp_len, L1 >> 6);
            }
        } else {
            const sq_meta m2 = P.metas2[r];
            const uint8_t *s2 = P.buf2 + m2.record_start + m2.sequence_offset;
            const uint64_t L2 = m2.sequence_length;
            const uint64_t fl = P.front_len < L1 ? P.front_len : L1;
            const uint64_t fo = P.front_off < L1 - fl ? P.front_off : L1 - fl;
            const uint64_t bl = P.back_len < L2 ? P.back_len : L2;
            const uint64_t bo = P.back_off < L2 - bl ? P.back_off : L2 - bl;
            if (fl + bl < fp_len) {
                special = 1; /* bytes of the previous fingerprint shine through (:4512-4516) */
                h = 0;
            } else {
                const uint8_t *front = s1 + fo, *back = s2 + bo;
                if (fl == 8 && bl == 8)
                    h = murmur3_x64_64_w4(sq_load_u64_unaligned(front), sq_load_u64_unaligned(back), 0, 0, 16,
                                          (L1 + L2) >> 6);
                else
                    h = murmur3_x64_64([&](uint64_t i) { return i < fl ? front[i] : back[i - fl]; },
                                       fp_len, (L1 + L2) >> 6);
            }
        }
        P.hashes[r] = h;
        P.special[r] = special;
    }
}

struct DedupKeep {
    unsigned long long ignore_mask;
    const unsigned long long *hashes;
    const unsigned char *special;
    __device__ bool operator()(const unsigned long long &idx) const
    {
        return (special && special[idx]) || (hashes[idx] & ignore_mask) == 0;
    }
};

struct DedupSpecialOnly {
    const unsigned char *special;
    __device__ bool operator()(const unsigned long long &idx) const { return special[idx] != 0; }
};

/* ================================ InsertSizeMetrics ============================== */

/* NUCLEOTIDE_COMPLEMENT, _qcmodule.c:5613-5631 */
__device__ __forceinline__ uint8_t complement_or_zero(uint8_t c)
{
    const unsigned l = c | 0x20u;
    return l == 'a' ? 'T' : l == 'c' ? 'G' : l == 'g' ? 'C' : l == 't' ? 'A' : 0;
}

/* Adapter remainders (InsertSizeMetrics_add_adapter, _qcmodule.c:5570-5611) are
 * counted in a device hash table keyed by the zero-padded 32-byte record
 * {length, bytes[31]}.  Besides the count every key keeps the rank of its first
 * occurrence (pair index over the whole run, read 1 before read 2), which is all
 * the reference's first-come, linear-probing table depends on: at read-out the
 * keys are replayed in rank order into a table of the reference's geometry. */
constexpr uint32_t ISZ_TABLE_BITS = 18;

struct IszTable {
    unsigned long long *hash;   /* 0 = free */
    unsigned long long *count;
    unsigned long long *rank;
    unsigned int *ready;        /* key bytes published */
    unsigned long long *key;    /* [size][4] */
    unsigned long long *n_distinct, *n_events;
    int *overflow;
    uint64_t mask;              /* slots - 1 */
};

struct IszParams {
    const uint8_t *buf1, *buf2;
    uint64_t len1, len2; /* bytes of buf1 / buf2 */
    const sq_meta *metas1, *metas2;
    uint64_t n;
    unsigned long long *insert_sizes; /* [cap] */
    uint32_t lds_sizes;               /* histogram entries privatised in LDS (0: straight to HBM) */
    unsigned long long *max_insert;
    IszTable tab[2];
    uint64_t rank_base;               /* pairs seen before this batch */
    int closed;                       /* the first-come cap was reached in an earlier batch */
};

/* the remainder a[0..len) (<= 31 bytes) as the 32-byte key {length, bytes} and its hash */
__device__ unsigned long long isz_key_of(const uint8_t *a, uint32_t len, const uint8_t *buf_end,
                                         unsigned long long key[4])
{
    unsigned long long r[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint8_t *src = a + 8 * k;
        unsigned long long w = 0;
        if (8u * k < len) {
            if (src + 8 <= buf_end) w = sq_load_u64_unaligned(src);
            else
                for (int b = 0; b < 8 && src + b < buf_end; b++) w |= (unsigned long long)src[b] << (8 * b);
            const uint32_t left = len - 8u * k;
            if (left < 8) w &= (1ULL << (8 * left)) - 1;
        }
        r[k] = w;
    }
    key[0] = len | (r[0] << 8);
    key[1] = (r[0] >> 56) | (r[1] << 8);
    key[2] = (r[1] >> 56) | (r[2] << 8);
    key[3] = (r[2] >> 56) | (r[3] << 8);
    const unsigned long long h = murmur3_x64_64_w4(r[0], r[1], r[2], r[3], len, 0);
    return h ? h : 1; /* 0 marks a free slot */
}

/* `count` occurrences of a key, the earliest of them at `rank`, into the device table */
__device__ void isz_table_add(const IszTable &T, const unsigned long long key[4], unsigned long long h,
                              unsigned long long count, unsigned long long rank, int closed)
{
    uint64_t idx = h & T.mask;
    bool mine = false;
    for (uint64_t spins = 0; spins < (1ull << 22); spins++) {
        unsigned long long cur = __hip_atomic_load(&T.hash[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == 0) {
            if (closed) return; /* a new key cannot be among the first max_adapters any more */
            cur = atomicCAS(&T.hash[idx], 0ULL, h);
            if (cur == 0) {
                for (int k = 0; k < 4; k++) T.key[idx * 4 + k] = key[k];
                __threadfence();
                __hip_atomic_store(&T.ready[idx], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                if (atomicAdd(T.n_distinct, 1ULL) > ((T.mask + 1) / 4) * 3) *T.overflow = 1;
                cur = h;
                mine = true;
            }
        }
        if (cur == h) {
            if (!mine && __hip_atomic_load(&T.ready[idx], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0)
                continue; /* the inserting lane publishes the key in this very iteration */
            bool same = true;
            if (!mine)
                for (int k = 0; k < 4; k++) same &= T.key[idx * 4 + k] == key[k];
            if (same) {
                atomicAdd(&T.count[idx], count);
                atomicMin(&T.rank[idx], rank);
                return;
            }
        }
        idx = (idx + 1) & T.mask;
    }
    *T.overflow = 1;
}

/* Adapter remainders repeat: most reads that run into the adapter leave the same 31 bytes.
 * A workgroup therefore counts them in a small LDS table first (write-once entries: hash,
 * key, count, earliest rank) and brings each entry to the device table once, instead of
 * hammering one device address with two atomics per read. */
constexpr uint32_t ISZ_CACHE = 256;   /* 32 held the full-length remainder and little else: the shorter ones (one key per length) went to the device table, all workgroups on the same few slots */
struct IszCache {
    unsigned long long hash[2][ISZ_CACHE], rank[2][ISZ_CACHE], key[2][ISZ_CACHE][4];
    unsigned int count[2][ISZ_CACHE], ready[2][ISZ_CACHE];
};

/* `count` occurrences of the key, the earliest at `rank`: into the workgroup's cache, or past it into the device table */
__device__ void isz_cache_add(const IszTable &T, IszCache &C, int which, const unsigned long long key[4], unsigned long long h,
                              unsigned int count, unsigned long long rank, int closed)
{
    const uint32_t e = (uint32_t)(h >> 20) & (ISZ_CACHE - 1);
    unsigned long long cur = atomicCAS(&C.hash[which][e], 0ULL, h);
    if (cur == 0) { /* this lane opens the entry */
        for (int k = 0; k < 4; k++) C.key[which][e][k] = key[k];
        __hip_atomic_store(&C.ready[which][e], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        cur = h;
    }
    if (cur == h && __hip_atomic_load(&C.ready[which][e], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) {
        bool same = true;
        for (int k = 0; k < 4; k++) same &= C.key[which][e][k] == key[k];
        if (same) {
            atomicAdd(&C.count[which][e], count);
            atomicMin(&C.rank[which][e], rank);
            return;
        }
    }
    isz_table_add(T, key, h, count, rank, closed); /* entry taken by another key (or not published yet) */
}

/* one remainder of the calling lane (any set of lanes may call) */
__device__ void isz_count_adapter(const IszTable &T, IszCache &C, int which, const uint8_t *a, uint32_t len,
                                  unsigned long long rank, int closed, const uint8_t *buf_end)
{
    unsigned long long key[4];
    const unsigned long long h = isz_key_of(a, len, buf_end, key);
    isz_cache_add(T, C, which, key, h, 1u, rank, closed);
}

/* bytes of v that are not zero */
__device__ __forceinline__ uint32_t isz_nonzero_bytes(uint32_t v)
{
    return (uint32_t)__popc((((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v) & 0x80808080u);
}

__device__ __forceinline__ uint32_t isz_wave_max(uint32_t v)
{
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = __shfl_xor(v, off);
        v = o > v ? o : v;
    }
    return v;
}

/* calculate_insert_size, _qcmodule.c:5667-5707, and the bookkeeping of
 * InsertSizeMetrics_add_sequence_pair_ptr :5709-5744 */
__global__ void k_insert_size(IszParams P)
{
    /* most pairs report the same few sizes (0 = no overlap found): count per workgroup first */
    extern __shared__ unsigned int l_sizes[];
    __shared__ unsigned int l_events[2]; /* number_of_adapters_read1/2 of this workgroup */
    __shared__ unsigned int l_max;       /* its largest insert size */
    __shared__ IszCache cache;
    for (uint32_t i = threadIdx.x; i < P.lds_sizes; i += blockDim.x) l_sizes[i] = 0;
    if (threadIdx.x < 2) l_events[threadIdx.x] = 0;
    if (threadIdx.x == 2) l_max = 0;
    for (uint32_t i = threadIdx.x; i < 2 * ISZ_CACHE; i += blockDim.x) {
        cache.hash[i / ISZ_CACHE][i % ISZ_CACHE] = 0;
        cache.rank[i / ISZ_CACHE][i % ISZ_CACHE] = ~0ULL;
        cache.count[i / ISZ_CACHE][i % ISZ_CACHE] = 0;
        cache.ready[i / ISZ_CACHE][i % ISZ_CACHE] = 0;
    }
    __syncthreads();
    unsigned long long local_max = 0;
    /* whole waves stay in the loop (the scan below runs in lock step): a lane behind the last
       pair has two empty reads */
    for (uint64_t r0 = blockIdx.x * (uint64_t)blockDim.x + (threadIdx.x & ~63u); r0 < P.n;
         r0 += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = r0 + (threadIdx.x & 63u);
        const bool valid = r < P.n;
        sq_meta m1 = {}, m2 = {};
        if (valid) { m1 = P.metas1[r]; m2 = P.metas2[r]; }
        const uint8_t *s1 = P.buf1 + m1.record_start + m1.sequence_offset;
        const uint8_t *s2 = P.buf2 + m2.record_start + m2.sequence_offset;
        const uint32_t L1 = m1.sequence_length, L2 = m2.sequence_length;
        uint32_t result = 0;
        uint64_t h_lo = 0, h_hi = 0, t_lo = 0, t_hi = 0;
        if (L1 >= 16 && L2 >= 16) {
            /* the two needles: reverse complements of the first and of the last 16 bases of
               read 2, from four 8-byte loads */
            {
                const uint64_t a0 = sq_load_u64_unaligned(s2), a1 = sq_load_u64_unaligned(s2 + 8);
                const uint64_t z0 = sq_load_u64_unaligned(s2 + L2 - 16), z1 = sq_load_u64_unaligned(s2 + L2 - 8);
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    /* needle byte (15 - i) = complement of R2 byte i */
                    h_hi |= (uint64_t)complement_or_zero((uint8_t)(a0 >> (8 * i))) << (8 * (7 - i));
                    h_lo |= (uint64_t)complement_or_zero((uint8_t)(a1 >> (8 * i))) << (8 * (7 - i));
                    t_hi |= (uint64_t)complement_or_zero((uint8_t)(z0 >> (8 * i))) << (8 * (7 - i));
                    t_lo |= (uint64_t)complement_or_zero((uint8_t)(z1 >> (8 * i))) << (8 * (7 - i));
                }
            }
        }
        /* read 1 streams through a 16-byte window that moves one base at a time and is refilled
           eight bytes at a time.  The loop runs in lock step for the wave (its trip count and the
           refill steps are scalar), a lane only branches when one of its four 8-byte compares
           hits (:5695), which is rare */
        {
            const bool scan = L1 >= 16 && L2 >= 16;
            const uint32_t last = scan ? L1 - 16 : 0;            /* last window start of this lane */
            /* windows the wave looks at; in a scalar register, so that the loop, its refill rounds
               and its early exit are scalar control flow */
            const uint32_t wave_last = (uint32_t)__builtin_amdgcn_readfirstlane((int)isz_wave_max(scan ? last + 1 : 0));
            /* the window as four dwords and the refill as two: a slide is six v_alignbyte_b32
               (64-bit shifts and compares are quarter-rate instructions) */
            uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0, n0 = 0, n1 = 0;
            if (scan) {
                const uint64_t lo = sq_load_u64_unaligned(s1), hi = sq_load_u64_unaligned(s1 + 8);
                w0 = (uint32_t)lo; w1 = (uint32_t)(lo >> 32); w2 = (uint32_t)hi; w3 = (uint32_t)(hi >> 32);
            }
            const uint8_t *end1 = P.buf1 + P.len1;
            const uint32_t UP4 = 0xDFDFDFDFu;
            const uint32_t hl = (uint32_t)h_lo, hh = (uint32_t)h_hi, tl = (uint32_t)t_lo, th = (uint32_t)t_hi;
            bool done = !scan;
            /* read 1 comes in 32 bytes at a time (a lane streaming its own read is the least
               efficient way to ask memory, the fewer and larger the requests the better: 8-byte
               refills 2.65 ms per 10 M pairs, 16-byte 2.44, 32-byte 2.34), fetched a round (32
               bases) before the window reaches them; the window takes them in quarters of eight */
            struct B32 { uint4 a, b; };
            auto refill = [&](uint32_t at) -> B32 { /* bytes [at, at + 32) of read 1 */
                B32 nx{make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
                if (scan && at < L1) {
                    const uint8_t *src = s1 + at;
                    if (src + 32 <= end1) { __builtin_memcpy(&nx.a, src, 16); __builtin_memcpy(&nx.b, src + 16, 16); }
                    else {
                        uint32_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                        for (int b = 0; b < 32 && src + b < end1; b++) t[b >> 2] |= (uint32_t)src[b] << (8 * (b & 3));
                        nx.a = make_uint4(t[0], t[1], t[2], t[3]);
                        nx.b = make_uint4(t[4], t[5], t[6], t[7]);
                    }
                }
                return nx;
            };
            B32 cur{make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)}, ahead = refill(16);
            for (uint32_t i = 0; i < wave_last; i++) {
                if ((i & 7u) == 0) { /* bytes [i + 16, i + 24) of read 1 */
                    if ((i & 63u) == 0 && i && __all(done || i > last)) break;
                    const uint32_t q = (i >> 3) & 3u;
                    if (q == 0) {
                        cur = ahead;
                        ahead = refill(i + 48);
                    }
                    n0 = q == 0 ? cur.a.x : q == 1 ? cur.a.z : q == 2 ? cur.b.x : cur.b.z;
                    n1 = q == 0 ? cur.a.y : q == 1 ? cur.a.w : q == 2 ? cur.b.y : cur.b.w;
                }
                /* a window can only match a needle half (:5695) if the half's low dword matches */
                const uint32_t u0 = w0 & UP4, u2 = w2 & UP4;
                const bool maybe = (u0 == hl) | (u2 == hh) | (u0 == tl) | (u2 == th);
                if (maybe && !done && i <= last) {
                    const uint32_t u1 = w1 & UP4, u3 = w3 & UP4;
                    /* :5695-5704: a half matches case-insensitively, then at most one raw byte
                       of the 16 may differ */
                    if ((u0 == hl && u1 == (uint32_t)(h_lo >> 32)) || (u2 == hh && u3 == (uint32_t)(h_hi >> 32))) {
                        const uint32_t d = isz_nonzero_bytes(w0 ^ hl) + isz_nonzero_bytes(w1 ^ (uint32_t)(h_lo >> 32)) +
                                           isz_nonzero_bytes(w2 ^ hh) + isz_nonzero_bytes(w3 ^ (uint32_t)(h_hi >> 32));
                        if (d <= 1) { result = i + 16; done = true; }
                    }
                    if (!done && ((u0 == tl && u1 == (uint32_t)(t_lo >> 32)) || (u2 == th && u3 == (uint32_t)(t_hi >> 32)))) {
                        const uint32_t d = isz_nonzero_bytes(w0 ^ tl) + isz_nonzero_bytes(w1 ^ (uint32_t)(t_lo >> 32)) +
                                           isz_nonzero_bytes(w2 ^ th) + isz_nonzero_bytes(w3 ^ (uint32_t)(t_hi >> 32));
                        if (d <= 1) { result = i + L2; done = true; }
                    }
                }
                w0 = __builtin_amdgcn_alignbyte(w1, w0, 1); /* slide the window by one base */
                w1 = __builtin_amdgcn_alignbyte(w2, w1, 1);
                w2 = __builtin_amdgcn_alignbyte(w3, w2, 1);
                w3 = __builtin_amdgcn_alignbyte(n0, w3, 1);
                n0 = __builtin_amdgcn_alignbyte(n1, n0, 1);
                n1 >>= 8;
            }
        }
        if (!valid) continue;
        if (result < P.lds_sizes) atomicAdd(&l_sizes[result], 1u);
        else atomicAdd(&P.insert_sizes[result], 1ULL);
        if (result) {
            if (result > local_max) local_max = result;
            const unsigned long long rank = 2 * (P.rank_base + r);
            if (L1 > result) { /* :5729-5735 */
                atomicAdd(&l_events[0], 1u);
                isz_count_adapter(P.tab[0], cache, 0, s1 + result, min(L1 - result, (uint32_t)SQ_ADAPTER_STORE_SIZE),
                                  rank, P.closed, P.buf1 + P.len1);
            }
            if (L2 > result) { /* :5736-5742 */
                atomicAdd(&l_events[1], 1u);
                isz_count_adapter(P.tab[1], cache, 1, s2 + result, min(L2 - result, (uint32_t)SQ_ADAPTER_STORE_SIZE),
                                  rank + 1, P.closed, P.buf2 + P.len2);
            }
        }
    }
    /* one global atomic per workgroup: half a million threads on one address take milliseconds */
    if (local_max) atomicMax(&l_max, (unsigned int)local_max);
    __syncthreads();
    if (threadIdx.x == 0 && l_max) atomicMax(P.max_insert, (unsigned long long)l_max);
    for (uint32_t t = threadIdx.x; t < 2 * ISZ_CACHE; t += blockDim.x) { /* the workgroup's remainders, each once */
        const uint32_t w = t / ISZ_CACHE, e = t % ISZ_CACHE;
        if (cache.hash[w][e] && cache.count[w][e])
            isz_table_add(P.tab[w], cache.key[w][e], cache.hash[w][e], cache.count[w][e], cache.rank[w][e], P.closed);
    }
    for (uint32_t i = threadIdx.x; i < P.lds_sizes; i += blockDim.x)
        if (l_sizes[i]) atomicAdd(&P.insert_sizes[i], (unsigned long long)l_sizes[i]);
    if (threadIdx.x < 2 && l_events[threadIdx.x])
        atomicAdd(P.tab[threadIdx.x].n_events, (unsigned long long)l_events[threadIdx.x]);
}

/* The adapter remainders of the pairs k_isz_span found one for (results[r] = their insert size):
 * InsertSizeMetrics_add_sequence_pair_ptr :5729-5742, as at the end of k_insert_size */
/* HIST: results[r] is the insert size of EVERY pair as the scan inside read 1's pass left it (k_span<PAIR = 2>,
 * sq_span_kernel.h): the histogram and the maximum (:5722-5727) are counted here, and only the pairs whose insert size is
 * shorter than one of the two (uniform) read lengths have their metas looked at */
template <bool HIST>
__global__ void __launch_bounds__(256) k_isz_adapters(IszParams P, const uint32_t *results, uint32_t len1, uint32_t len2)
{
    __shared__ unsigned int l_events[2];
    __shared__ IszCache cache;
    __shared__ unsigned int l_hist[HIST ? 1024 : 1], l_hmax;
    if (HIST) {
        for (uint32_t i = threadIdx.x; i < 1024; i += blockDim.x) l_hist[i] = 0;
        if (threadIdx.x == 0) l_hmax = 0;
    }
    uint32_t local_max = 0;
    if (threadIdx.x < 2) l_events[threadIdx.x] = 0;
    for (uint32_t i = threadIdx.x; i < 2 * ISZ_CACHE; i += blockDim.x) {
        cache.hash[i / ISZ_CACHE][i % ISZ_CACHE] = 0;
        cache.rank[i / ISZ_CACHE][i % ISZ_CACHE] = ~0ULL;
        cache.count[i / ISZ_CACHE][i % ISZ_CACHE] = 0;
        cache.ready[i / ISZ_CACHE][i % ISZ_CACHE] = 0;
    }
    __syncthreads();
    /* One pair in ten of the synthetic ones leaves a remainder, so every wave holds some: handled where it is met, the long path
       (two metas, the remainder's bytes, a hash, the workgroup's cache: chains of dependent loads) ran for a few lanes of
       every wave -- 0.85 ms per 25 M pairs for 0.1 GB of results.  Now a wave queues the pairs that need it (index, insert
       size: a queue of its own in LDS, slots handed out by a ballot, no barrier) and works 64 of them off with all its lanes
       whenever it holds that many: 0.85 -> 0.53 ms.  (Tried on top and dropped: the wave sorting its lanes by remainder first, one
       lane adding a group's count and earliest rank to the cache -- 0.74 ms: the remainders of a wave are of many lengths, and a
       round of ballots and a 64-bit reduction per distinct one costs more than the queueing at the cache's entries.)  Four pairs
       per thread and turn: one 16-byte load of results. */
    constexpr uint32_t WQ = 128;                       /* a wave adds at most 64 to fewer than 64 */
    __shared__ uint32_t l_queue[4][3 * WQ];            /* per wave (256 threads): the pair's index (64 bits), its insert size */
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t *wq = l_queue[wave];
    uint32_t wq_n = 0;                                  /* wave-uniform */
    auto remainder = [&](uint64_t r, uint32_t result) {   /* :5729-5742 */
        const sq_meta m1 = P.metas1[r], m2 = P.metas2[r];
        const uint8_t *s1 = P.buf1 + m1.record_start + m1.sequence_offset;
        const uint8_t *s2 = P.buf2 + m2.record_start + m2.sequence_offset;
        const uint32_t L1 = m1.sequence_length, L2 = m2.sequence_length;
        const unsigned long long rank = 2 * (P.rank_base + r);
        if (L1 > result) {
            atomicAdd(&l_events[0], 1u);
            isz_count_adapter(P.tab[0], cache, 0, s1 + result, min(L1 - result, (uint32_t)SQ_ADAPTER_STORE_SIZE),
                              rank, P.closed, P.buf1 + P.len1);
        }
        if (L2 > result) {
            atomicAdd(&l_events[1], 1u);
            isz_count_adapter(P.tab[1], cache, 1, s2 + result, min(L2 - result, (uint32_t)SQ_ADAPTER_STORE_SIZE),
                              rank + 1, P.closed, P.buf2 + P.len2);
        }
    };
    auto work_off = [&](uint32_t count) {   /* the wave's first `count` (<= 64) entries, then the rest moves down */
        if (lane < count) remainder(((uint64_t)wq[3 * lane + 1] << 32) | wq[3 * lane], wq[3 * lane + 2]);
        const uint32_t left = wq_n - count;           /* < 64 */
        uint32_t e0 = 0, e1 = 0, e2 = 0;
        if (lane < left) { e0 = wq[3 * (count + lane)]; e1 = wq[3 * (count + lane) + 1]; e2 = wq[3 * (count + lane) + 2]; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane < left) { wq[3 * lane] = e0; wq[3 * lane + 1] = e1; wq[3 * lane + 2] = e2; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        wq_n = left;
    };
    const bool aligned16 = ((uintptr_t)results & 15u) == 0;
    const uint64_t n4 = (P.n + 3) & ~3ULL;
    /* whole waves stay in the loop (ballots): a lane behind the last pair has four invalid ones */
    for (uint64_t w0 = ((uint64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u)) * 4; w0 < n4; w0 += (uint64_t)gridDim.x * blockDim.x * 4) {
        const uint64_t r0 = w0 + (uint64_t)lane * 4;
        uint32_t res[4] = {0, 0, 0, 0};
        if (aligned16 && r0 + 4 <= P.n) {
            const uint4 v = *(const uint4 *)(results + r0);
            res[0] = v.x; res[1] = v.y; res[2] = v.z; res[3] = v.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) if (r0 + k < P.n) res[k] = results[r0 + k];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint64_t r = r0 + k;
            const uint32_t result = res[k];
            const bool valid = r < P.n;
            if (HIST) {
                /* most pairs have no overlap (result 0): 64 lanes adding to ONE LDS word take 64 turns; the wave counts its
                   zeroes with a ballot and one lane adds them */
                const unsigned long long zeroes = __builtin_amdgcn_ballot_w64(valid && result == 0);
                if (valid) {
                    if (result == 0) {
                        if (lane == (uint32_t)__ffsll((long long)zeroes) - 1u) atomicAdd(&l_hist[0], (unsigned int)__popcll(zeroes));
                    } else if (result < 1024) atomicAdd(&l_hist[result], 1u);
                    else atomicAdd(&P.insert_sizes[result], 1ULL);
                    local_max = max(local_max, result);
                }
            }
            const bool need = valid && result && !(HIST && result >= len1 && result >= len2);   /* a remainder on one side at least */
            const unsigned long long needs = __builtin_amdgcn_ballot_w64(need);
            if (need) {
                const uint32_t at = wq_n + (uint32_t)__popcll(needs & ((1ULL << lane) - 1));
                wq[3 * at] = (uint32_t)r; wq[3 * at + 1] = (uint32_t)(r >> 32); wq[3 * at + 2] = result;
            }
            wq_n += (uint32_t)__popcll(needs);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (wq_n >= 64) work_off(64);
        }
    }
    if (wq_n) work_off(wq_n);
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < 2 * ISZ_CACHE; t += blockDim.x) { /* the workgroup's remainders, each once */
        const uint32_t w = t / ISZ_CACHE, e = t % ISZ_CACHE;
        if (cache.hash[w][e] && cache.count[w][e])
            isz_table_add(P.tab[w], cache.key[w][e], cache.hash[w][e], cache.count[w][e], cache.rank[w][e], P.closed);
    }
    if (threadIdx.x < 2 && l_events[threadIdx.x])
        atomicAdd(P.tab[threadIdx.x].n_events, (unsigned long long)l_events[threadIdx.x]);
    if (HIST) {
        if (local_max) atomicMax(&l_hmax, local_max);
        __syncthreads();
        if (threadIdx.x == 0 && l_hmax) atomicMax(P.max_insert, (unsigned long long)l_hmax);
        for (uint32_t i = threadIdx.x; i < 1024; i += blockDim.x)
            if (l_hist[i]) atomicAdd(&P.insert_sizes[i], (unsigned long long)l_hist[i]);
    }
}

/* used slots of an adapter table, unordered */
__global__ void k_isz_collect(IszTable T, unsigned long long *out_slot, unsigned long long *n_out)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i <= T.mask;
         i += (uint64_t)gridDim.x * blockDim.x)
        if (T.hash[i] && T.count[i]) out_slot[atomicAdd(n_out, 1ULL)] = i;
}

__global__ void k_isz_gather(IszTable T, const unsigned long long *slots, uint64_t n, unsigned long long *key,
                             unsigned long long *count, unsigned long long *rank)
{
    for (uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; e < n;
         e += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = slots[e];
        for (int k = 0; k < 4; k++) key[e * 4 + k] = T.key[i * 4 + k];
        count[e] = T.count[i];
        rank[e] = T.rank[i];
    }
}

/* counts of given 32-byte keys {length, bytes[31]} in a table (0: absent) */
__global__ void k_isz_lookup(IszTable T, const unsigned long long *keys, uint64_t n, unsigned long long *out)
{
    for (uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; e < n;
         e += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long *key = keys + e * 4;
        const uint8_t *rec = (const uint8_t *)key;
        const uint32_t len = rec[0];
        unsigned long long h = murmur3_x64_64([&](uint64_t i) { return rec[1 + i]; }, len, 0);
        if (h == 0) h = 1;
        uint64_t idx = h & T.mask;
        unsigned long long c = 0;
        for (;;) {
            const unsigned long long cur = T.hash[idx];
            if (cur == 0) break;
            if (cur == h) {
                bool same = true;
                for (int k = 0; k < 4; k++) same &= T.key[idx * 4 + k] == key[k];
                if (same) { c = T.count[idx]; break; }
            }
            idx = (idx + 1) & T.mask;
        }
        out[e] = c;
    }
}

/* survivors only travel to the host: hash and the "needs the host" flag of every kept index */
__global__ void k_dedup_gather(const unsigned long long *idx, uint64_t n_keep,
                               const unsigned long long *hashes, const unsigned char *special,
                               unsigned long long *out_hash, unsigned char *out_special)
{
    for (uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; e < n_keep;
         e += (uint64_t)gridDim.x * blockDim.x) {
        out_hash[e] = hashes[idx[e]];
        if (out_special) out_special[e] = special ? special[idx[e]] : 0;
    }
}

/* sorted hashes -> how many DISTINCT ones have exactly c trailing zero bits (c = 64: the hash 0), hist[65] */
__global__ void k_dedup_ctz_hist(const unsigned long long *sorted, uint64_t n, unsigned long long *hist)
{
    __shared__ unsigned int l_hist[65];
    for (int i = threadIdx.x; i < 65; i += blockDim.x) l_hist[i] = 0;
    __syncthreads();
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long h = sorted[i];
        if (i != 0 && sorted[i - 1] == h) continue;
        atomicAdd(&l_hist[h ? __builtin_ctzll(h) : 64], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 65; i += blockDim.x)
        if (l_hist[i]) atomicAdd(&hist[i], (unsigned long long)l_hist[i]);
}

/* ---- DedupEstimator's table in HBM (round 6) ---------------------------------------------------------------
 *
 * The reference's estimator (:4426-4460) is sequential: a hash that passes the mask is looked up by linear probing,
 * counted if found, inserted into the first empty slot if not; an arrival that finds the table full rebuilds it
 * first.  What of that is observable -- which hashes are stored, their counts AND their slots (duplication_counts()
 * walks the table in slot order, :4736-4744) -- is reproduced by parallel steps, piece by piece of the stream:
 *
 *   k_dd_classify   every survivor of the piece (the hashes that pass the mask, in read order) is looked up in the
 *                   table AS IT STOOD WHEN THE PIECE BEGAN: entries never move or leave between rebuilds, so what is
 *                   found then is found when its turn comes, in the same slot
 *   (sort)          the survivors not found, stably by hash: the first of several equal ones will insert, the
 *                   others find it (they become its count)
 *   k_dd_heads ...  which arrival fills the table (the stored count reaches max_stored at the need-th FIRST
 *                   occurrence of a new hash); the survivor behind it is the one whose arrival rebuilds, and the
 *                   piece ends in front of it
 *   k_dd_insert     linear probing by PRIORITY (Shun & Blelloch, "Phase-concurrent hash tables for determinism",
 *                   SPAA 2014: an element that meets one of lower priority takes its slot and the displaced one
 *                   moves on): with the order of arrival as the priority and resident entries above all, the layout
 *                   is the one sequential insertion in order of arrival leaves, whatever order the lanes run in.
 *                   The slots hold TOKENS (0: a resident entry, 1 + rank of arrival, ~0: empty) while the lanes
 *                   compete; k_dd_finalize writes hash and count where the tokens came to rest
 *   k_dd_rebuild_*  DedupEstimator_increment_modulo (:4383-4423) the same way: the entries that pass the stricter
 *                   mask into an empty table, priority = slot in the old one (the order the reference walks it)
 *   k_dd_trigger    the arrival that caused the rebuild, placed with the OLD bit count -- one lane
 *
 * One thing a lookup at the piece's start cannot see: an entry placed by k_dd_trigger does not sit in the probe run
 * of its own hash, so a later arrival of that hash finds it only if entries inserted meanwhile have closed the gap.
 * Such an entry is tracked (`odd`); a piece in which its hash arrives without being found takes the host's
 * sequential loop (dedup_host_piece) -- about one piece in twenty rebuilds on the synthetic reads.
 */
constexpr unsigned int DD_EMPTY = 0xFFFFFFFFu, DD_MISS = 0xFFFFFFFFu;

/* survivors: sel[e] = index (relative to the piece) of the e-th hash that passes the mask */
__global__ void k_dd_classify(const unsigned long long *hashes, const unsigned long long *sel, uint64_t n_sel,
                              const unsigned long long *thash, const unsigned int *tcount, uint64_t bits,
                              uint64_t mask, unsigned long long odd_hash, int odd_valid, unsigned int *slot_out,
                              unsigned int *flags /* [0]: the odd entry's hash arrived and was not found */)
{
    for (uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; e < n_sel; e += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long h = hashes[sel[e]];
        unsigned int found = DD_MISS;
        for (uint64_t i = (h >> bits) & mask;; i = (i + 1) & mask) {
            if (tcount[i] == 0) break;
            if (thash[i] == h) { found = (unsigned int)i; break; }
        }
        slot_out[e] = found;
        if (found == DD_MISS && odd_valid && h == odd_hash) flags[0] = 1;
    }
}

struct DedupIsMiss {
    const unsigned int *slot;
    __device__ bool operator()(const unsigned long long &e) const { return slot[e] == DD_MISS; }
};

/* miss[r] = survivor index e of the r-th survivor not found -> its hash (the sort's key) and r (the value) */
__global__ void k_dd_miss_keys(const unsigned long long *hashes, const unsigned long long *sel, const unsigned long long *miss,
                               uint64_t n_miss, unsigned long long *key, unsigned long long *val)
{
    for (uint64_t r = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; r < n_miss; r += (uint64_t)gridDim.x * blockDim.x) {
        key[r] = hashes[sel[miss[r]]];
        val[r] = r;
    }
}

/* sorted by hash (stable: equal hashes in order of arrival): first[r] = 1 for the first arrival of every new hash */
__global__ void k_dd_heads(const unsigned long long *skey, const unsigned long long *sval, uint64_t n_miss, unsigned int *first)
{
    for (uint64_t j = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; j < n_miss; j += (uint64_t)gridDim.x * blockDim.x)
        first[sval[j]] = (j == 0 || skey[j - 1] != skey[j]) ? 1u : 0u;
}

/* pos = inclusive sum of first: out[0] = the r whose first occurrence is the need-th new hash */
__global__ void k_dd_find_nth(const unsigned int *first, const unsigned int *pos, uint64_t n_miss, unsigned int need,
                              unsigned long long *out)
{
    for (uint64_t r = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; r < n_miss; r += (uint64_t)gridDim.x * blockDim.x)
        if (first[r] && pos[r] == need) out[0] = r;
}

__global__ void k_dd_tok_init(const unsigned int *tcount, unsigned int *tok, uint64_t size)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < size; i += (uint64_t)gridDim.x * blockDim.x)
        tok[i] = tcount[i] ? 0u : DD_EMPTY;
}

/* the token v looks for its slot from slot i on: smaller tokens stay, a larger one (or none) gives way and moves on */
__device__ __forceinline__ void dd_priority_insert(unsigned int *tok, uint64_t mask, uint64_t i, unsigned int v)
{
    for (;;) {
        const unsigned int c = __hip_atomic_load(&tok[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (c < v) { i = (i + 1) & mask; continue; }
        if (atomicCAS(&tok[i], c, v) != c) continue;   /* somebody else was faster: look at the slot again */
        if (c == DD_EMPTY) return;
        v = c;                                           /* the displaced token goes on from the next slot */
        i = (i + 1) & mask;
    }
}

/* found survivors in front of the piece's end: their entry's count; heads of the sorted misses: the count the new
 * entry will have (its arrivals in front of the end) and the token's walk into the table */
__global__ void k_dd_apply_found(const unsigned int *slot, uint64_t e_end, unsigned int *tcount)
{
    for (uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; e < e_end; e += (uint64_t)gridDim.x * blockDim.x)
        if (slot[e] != DD_MISS) atomicAdd(&tcount[slot[e]], 1u);
}

__global__ void __launch_bounds__(256) k_dd_insert(const unsigned long long *skey, const unsigned long long *sval, uint64_t n_miss,
                            const unsigned long long *miss, uint64_t e_end, uint64_t bits, uint64_t mask,
                            unsigned int *tok, unsigned int *new_count /* [n_miss], by rank */,
                            unsigned long long *n_inserted)
{
    for (uint64_t base = blockIdx.x * 256ull; base < n_miss; base += gridDim.x * 256ull) {   /* every lane of a wave the same trips */
        const uint64_t j = base + threadIdx.x;
        bool mine = j < n_miss && (j == 0 || skey[j - 1] != skey[j]);
        unsigned int r = 0;
        if (mine) {
            r = (unsigned int)sval[j];
            mine = miss[r] < e_end;                       /* the first arrival lies behind the end: so do all */
        }
        const unsigned long long b = __ballot(mine);
        if (b && (threadIdx.x & 63) == 0) atomicAdd(n_inserted, (unsigned long long)__popcll(b));
        if (!mine) continue;
        unsigned int c = 0;
        for (uint64_t k = j; k < n_miss && skey[k] == skey[j]; k++) c += miss[sval[k]] < e_end;
        new_count[r] = c;
        dd_priority_insert(tok, mask, (skey[j] >> bits) & mask, r + 1);
    }
}

__global__ void k_dd_finalize(const unsigned int *tok, uint64_t size, const unsigned long long *key /* by rank */,
                              const unsigned int *new_count, unsigned long long *thash, unsigned int *tcount)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < size; i += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned int t = tok[i];
        if (t == 0 || t == DD_EMPTY) continue;
        thash[i] = key[t - 1];
        tcount[i] = new_count[t - 1];
    }
}

/* DedupEstimator_increment_modulo :4383-4423: the old table's entries that pass the mask of new_bits bits */
__global__ void __launch_bounds__(256) k_dd_rebuild_insert(const unsigned long long *thash, const unsigned int *tcount, uint64_t size,
                                    uint64_t new_bits, unsigned int *tok, unsigned long long *kept)
{
    const unsigned long long ignore = (1ULL << new_bits) - 1;
    for (uint64_t base = blockIdx.x * 256ull; base < size; base += gridDim.x * 256ull) {
        const uint64_t i = base + threadIdx.x;
        const bool mine = i < size && tcount[i] != 0 && (thash[i] & ignore) == 0;
        const unsigned long long b = __ballot(mine);
        if (b && (threadIdx.x & 63) == 0) atomicAdd(kept, (unsigned long long)__popcll(b));
        if (mine) dd_priority_insert(tok, size - 1, (thash[i] >> new_bits) & (size - 1), (unsigned int)i + 1);
    }
}

__global__ void k_dd_rebuild_finalize(const unsigned int *tok, uint64_t size, const unsigned long long *ohash,
                                      const unsigned int *ocount, unsigned long long *nhash, unsigned int *ncount)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < size; i += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned int t = tok[i];
        nhash[i] = t == DD_EMPTY ? 0ull : ohash[t - 1];
        ncount[i] = t == DD_EMPTY ? 0u : ocount[t - 1];
    }
}

/* the tail of DedupEstimator_add_fingerprint :4453-4459 for the arrival that rebuilt: the slot by the OLD bit count.
 * out[0] = the hash, out[1] = 1 if it was inserted (0: found and counted) */
__global__ void k_dd_trigger(const unsigned long long *hashes, const unsigned long long *sel, uint64_t e,
                             uint64_t old_bits, uint64_t mask, unsigned long long *thash, unsigned int *tcount,
                             unsigned long long *out)
{
    const unsigned long long h = hashes[sel[e]];
    out[0] = h;
    for (uint64_t i = (h >> old_bits) & mask;; i = (i + 1) & mask) {
        if (tcount[i] == 0) { thash[i] = h; tcount[i] = 1; out[1] = 1; return; }
        if (thash[i] == h) { tcount[i]++; out[1] = 0; return; }
    }
}

/* hashes the host made for pairs shorter than the fingerprint, into the batch's hashes */
__global__ void k_dd_patch(unsigned long long *hashes, const unsigned long long *pos, const unsigned long long *val, uint64_t n)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        hashes[pos[i]] = val[i];
}

int blocks_for(uint64_t n, int cap = 16384)
{
    uint64_t b = (n + 255) / 256;
    if (b > (uint64_t)cap) b = cap;
    return (int)(b ? b : 1);
}

} // namespace

/* ================================== module objects ================================= */

struct sq_overrep {
    sq_ctx *ctx;
    uint64_t max_unique, k, sample_every;
    long long frags_start, frags_end;
    uint64_t number_of_sequences = 0, sampled_sequences = 0;
    uint64_t table_size = 0;
    bool full = false;
    uint32_t *d_occupied = nullptr;   /* the closed table's occupied slots as bits; made at the first launch on the closed table */
    bool occupied_valid = false;
    unsigned long long *d_hashes = nullptr, *d_ranks = nullptr;
    unsigned int *d_counts = nullptr;
    unsigned long long *d_scalars = nullptr; /* [0] n_unique [1] total_fragments [2] warn_count [3] warn_last */
    uint64_t n_unique_host = 0;              /* value after the last synchronised batch */
    /* shard mode: uncapped, ranked table over records [first_record, ...) of the job */
    bool shard = false;
    uint64_t first_record = 0;
};

SQ_EXPORT sq_overrep *sq_overrep_new(sq_ctx *ctx, int64_t max_unique_fragments, int64_t fragment_length,
                                     int64_t sample_every, int64_t bases_from_start,
                                     int64_t bases_from_end)
{
    /* OverrepresentedSequences__new__, _qcmodule.c:3464-3540 */
    if (max_unique_fragments < 1) {
        sq_set_error("max_unique_fragments should be at least 1, got: %lld", (long long)max_unique_fragments);
        return nullptr;
    }
    if ((fragment_length & 1) == 0 || fragment_length > 31 || fragment_length < 3) {
        sq_set_error("fragment_length must be between 3 and 31 and be an uneven number, got: %lld",
                     (long long)fragment_length);
        return nullptr;
    }
    if (sample_every < 1) {
        sq_set_error("sample_every must be 1 or greater. Got %lld", (long long)sample_every);
        return nullptr;
    }
    if (bases_from_start < 0) bases_from_start = UINT32_MAX;
    if (bases_from_end < 0) bases_from_end = UINT32_MAX;
    sq_overrep *o = new sq_overrep();
    o->ctx = ctx;
    o->max_unique = max_unique_fragments;
    o->k = fragment_length;
    o->sample_every = sample_every;
    o->frags_start = (bases_from_start + fragment_length - 1) / fragment_length;
    o->frags_end = (bases_from_end + fragment_length - 1) / fragment_length;
    /* sized for the cap plus one launch worth of not-yet-capped keys; the
       reported contents do not depend on the table geometry */
    uint64_t want = 2 * o->max_unique + (1u << 16);
    o->table_size = 1;
    while (o->table_size < want) o->table_size <<= 1;
    SQ_HIP_NULL(hipMalloc((void **)&o->d_hashes, o->table_size * 8));
    SQ_HIP_NULL(hipMalloc((void **)&o->d_counts, o->table_size * 4));
    SQ_HIP_NULL(hipMalloc((void **)&o->d_scalars, 4 * 8));
    SQ_HIP_NULL(hipMemsetAsync(o->d_hashes, 0, o->table_size * 8, ctx->stream));
    SQ_HIP_NULL(hipMemsetAsync(o->d_counts, 0, o->table_size * 4, ctx->stream));
    SQ_HIP_NULL(hipMemsetAsync(o->d_scalars, 0, 3 * 8, ctx->stream));
    SQ_HIP_NULL(hipMemsetAsync(o->d_scalars + 3, 0xFF, 8, ctx->stream)); /* -1 */
    SQ_HIP_NULL(hipStreamSynchronize(ctx->stream));
    return o;
}

SQ_EXPORT void sq_overrep_free(sq_overrep *o)
{
    if (!o) return;
    (void)hipStreamSynchronize(o->ctx->stream);
    for (void *p : {(void *)o->d_hashes, (void *)o->d_counts, (void *)o->d_ranks, (void *)o->d_scalars, (void *)o->d_occupied})
        if (p) (void)hipFree(p);
    delete o;
}

/* shard mode: room for `keys` distinct hashes at <= 70 % load (the table doubles) */
static int ovr_reserve(sq_overrep *o, uint64_t keys)
{
    sq_ctx *ctx = o->ctx;
    uint64_t size = o->table_size;
    while (keys > size / 10 * 7) size <<= 1;
    if (size == o->table_size) return SQ_OK;
    if (size > (1ULL << 32)) { sq_set_error("OverrepresentedSequences: shard table beyond 2^32 slots"); return SQ_ERR_MEMORY; }
    unsigned long long *nh = nullptr, *nr = nullptr;
    unsigned int *nc = nullptr;
    SQ_HIP(hipMalloc((void **)&nh, size * 8));
    SQ_HIP(hipMalloc((void **)&nc, size * 4));
    SQ_HIP(hipMalloc((void **)&nr, size * 8));
    SQ_HIP(hipMemsetAsync(nh, 0, size * 8, ctx->stream));
    SQ_HIP(hipMemsetAsync(nc, 0, size * 4, ctx->stream));
    SQ_HIP(hipMemsetAsync(nr, 0xFF, size * 8, ctx->stream));
    hipLaunchKernelGGL(k_ovr_rehash, dim3(blocks_for(o->table_size)), dim3(256), 0, ctx->stream, o->d_hashes,
                       o->d_counts, o->d_ranks, o->table_size, nh, nc, nr, size - 1);
    SQ_HIP(hipStreamSynchronize(ctx->stream));
    (void)hipFree(o->d_hashes); (void)hipFree(o->d_counts); (void)hipFree(o->d_ranks);
    o->d_hashes = nh; o->d_counts = nc; o->d_ranks = nr;
    o->table_size = size;
    return SQ_OK;
}

SQ_EXPORT int sq_overrep_add_batch(sq_overrep *o, sq_batch *b)
{
    sq_ctx *ctx = o->ctx;
    const uint64_t n = b->n;
    /* records with (number_of_sequences + r) % sample_every == 0 are sampled (:3833) */
    const uint64_t record_base = o->first_record + o->number_of_sequences;
    const uint64_t phase = record_base % o->sample_every;
    const uint64_t first = phase == 0 ? 0 : o->sample_every - phase;
    const uint64_t n_samples = first < n ? (n - first + o->sample_every - 1) / o->sample_every : 0;
    /* sampled records of the job in front of this batch */
    const uint64_t samples_before = (record_base + o->sample_every - 1) / o->sample_every;
    o->number_of_sequences += n;
    o->sampled_sequences += n_samples;
    if (n_samples == 0) return SQ_OK;

    /* an upper bound of the fragments one read can stage */
    const uint64_t k = o->k, maxL = b->max_length;
    if (maxL < k) return SQ_OK;
    const uint64_t max_frag = (maxL + k - 1) / k;
    uint64_t per_read = std::min<uint64_t>(o->frags_start, max_frag - max_frag / 2) +
                        std::min<uint64_t>(o->frags_end, max_frag / 2);
    if (per_read == 0) return SQ_OK;
    uint64_t big_size = 1;
    while (2 * big_size < 3 * per_read) big_size <<= 1;
    const bool need_big = big_size > 32;

    OvrParams P{};
    P.buf = b->d_buf; P.buf_len = b->buf_len; P.metas = b->d_metas; P.n = n;
    P.first_sample = first; P.record_base = record_base;
    P.k = (uint32_t)k; P.sample_every = (uint32_t)o->sample_every;
    P.frags_start = o->frags_start; P.frags_end = o->frags_end;
    P.hashes = o->d_hashes; P.counts = o->d_counts; P.table_mask = o->table_size - 1;
    P.n_unique = o->d_scalars; P.total_fragments = o->d_scalars + 1;
    P.warn_count = o->d_scalars + 2; P.warn_last = (long long *)(o->d_scalars + 3);
    P.big_size = big_size;

    uint64_t done = 0;
    while (done < n_samples) {
        uint64_t chunk = n_samples - done;
        int mode;
        if (o->shard) {
            mode = OVR_RANKED;
            int rc = ovr_reserve(o, o->n_unique_host + chunk * per_read);
            if (rc) return rc;
            P.hashes = o->d_hashes; P.counts = o->d_counts; P.table_mask = o->table_size - 1;
        } else if (o->full) {
            mode = OVR_FULL;
            if (!o->occupied_valid && o->table_size >= 32) {   /* the keys do not change any more */
                if (!o->d_occupied) SQ_HIP(hipMalloc((void **)&o->d_occupied, o->table_size / 8));
                hipLaunchKernelGGL(k_ovr_occupied, dim3(blocks_for(o->table_size / 32)), dim3(256), 0, ctx->stream, o->d_hashes, o->table_size, o->d_occupied);
                o->occupied_valid = true;
            }
            P.occupied = o->occupied_valid ? o->d_occupied : nullptr;
        } else {
            /* keys this launch can add at most; keep the open-addressing table under ~80 % */
            const uint64_t room_cap = o->max_unique - o->n_unique_host;
            const uint64_t room_tab = (o->table_size / 5) * 4 - o->n_unique_host;
            if (chunk * per_read <= room_cap) {
                mode = OVR_NORMAL;
            } else {
                mode = OVR_CROSSING;
                chunk = std::max<uint64_t>(1, std::min<uint64_t>(chunk, room_tab / per_read));
                if (!o->d_ranks) SQ_HIP(hipMalloc((void **)&o->d_ranks, o->table_size * 8));
                hipLaunchKernelGGL(k_ovr_mark_old, dim3(blocks_for(o->table_size)), dim3(256), 0,
                                   ctx->stream, o->d_hashes, o->d_ranks, o->table_size);
            }
        }
        if (need_big) { /* bound the workspace */
            const uint64_t max_chunk = std::max<uint64_t>(1, (1ull << 30) / (big_size * 8));
            chunk = std::min(chunk, max_chunk);
            SQ_HIP(hipMalloc((void **)&P.big_staging, chunk * big_size * 8));
        }
        P.mode = mode;
        P.ranks = o->d_ranks;
        P.sample_base = done;
        P.rank_base = o->shard ? samples_before + done : done;
        P.n_samples = chunk;
        /* a workgroup counts hot fragments in LDS first: fewer, longer-lived ones.  Twice the workgroups that fit at a time:
           the pass is a chain of memory round trips per lane, and the second half fills the tail the first leaves
           (20.2 -> 17.7 ms per 100 M reads; profiles/r6/exp_overrep.txt) */
        if (mode == OVR_FULL && P.occupied && per_read <= (uint64_t)OVR_PAR_F && k <= 24 && !sq_knobs().overrep_chain)   /* the closed table: every load of a lane in flight at once */
            hipLaunchKernelGGL(k_overrep_par, dim3(blocks_for(chunk, ctx->num_cus * 8)), dim3(256), 0, ctx->stream, P);
        else
            hipLaunchKernelGGL(k_overrep, dim3(blocks_for(chunk, ctx->num_cus * 16)), dim3(256), 0, ctx->stream, P);
        SQ_HIP(hipGetLastError());
        if (need_big) {
            SQ_HIP(hipStreamSynchronize(ctx->stream));
            SQ_HIP(hipFree(P.big_staging));
            P.big_staging = nullptr;
        }
        done += chunk;
        if (mode == OVR_FULL) continue;
        SQ_HIP(hipMemcpyAsync(&ctx->pinned[24], o->d_scalars, 8, hipMemcpyDeviceToHost, ctx->stream));
        SQ_HIP(hipStreamSynchronize(ctx->stream));
        const uint64_t n_after = ctx->pinned[24];
        if (mode == OVR_CROSSING && n_after > o->max_unique) {
            /* H2: the table keeps the first max_unique distinct hashes in (sampled read,
               staging slot) order; drop the new keys that rank behind the cut */
            const uint64_t n_new = n_after - o->n_unique_host, keep = o->max_unique - o->n_unique_host;
            unsigned long long *d_buf = nullptr, *d_n = nullptr;
            SQ_HIP(hipMalloc((void **)&d_buf, 4 * n_new * 8));
            SQ_HIP(hipMalloc((void **)&d_n, 8));
            SQ_HIP(hipMemsetAsync(d_n, 0, 8, ctx->stream));
            unsigned long long *d_rank = d_buf, *d_slot = d_buf + n_new, *d_rank2 = d_buf + 2 * n_new,
                               *d_slot2 = d_buf + 3 * n_new;
            hipLaunchKernelGGL(k_ovr_collect_new, dim3(blocks_for(o->table_size)), dim3(256), 0,
                               ctx->stream, o->d_ranks, o->table_size, d_rank, d_slot, d_n);
            /* by rank, on the device: the keys behind the first `keep` lose their counts */
            size_t temp_bytes = 0;
            void *d_temp = nullptr;
            SQ_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, d_rank, d_rank2, d_slot, d_slot2, (int)n_new, 0,
                                                      64, ctx->stream));
            SQ_HIP(hipMalloc(&d_temp, temp_bytes ? temp_bytes : 8));
            SQ_HIP(hipcub::DeviceRadixSort::SortPairs(d_temp, temp_bytes, d_rank, d_rank2, d_slot, d_slot2, (int)n_new, 0,
                                                      64, ctx->stream));
            if (n_new > keep)
                hipLaunchKernelGGL(k_ovr_kill, dim3(blocks_for(n_new - keep)), dim3(256), 0, ctx->stream,
                                   o->d_counts, d_slot2 + keep, (uint64_t)(n_new - keep));
            unsigned long long capped = o->max_unique;
            SQ_HIP(hipMemcpyAsync(o->d_scalars, &capped, 8, hipMemcpyHostToDevice, ctx->stream));
            SQ_HIP(hipStreamSynchronize(ctx->stream));
            (void)hipFree(d_temp); (void)hipFree(d_buf); (void)hipFree(d_n);
            o->n_unique_host = o->max_unique;
        } else {
            o->n_unique_host = n_after;
        }
        if (!o->shard && o->n_unique_host >= o->max_unique && !o->full) {
            o->full = true;
            if (mode == OVR_CROSSING) {   /* dead keys in the table: out with them (k_ovr_compact) */
                unsigned long long *nh = nullptr;
                unsigned int *nc = nullptr;
                SQ_HIP(hipMalloc((void **)&nh, o->table_size * 8));
                SQ_HIP(hipMalloc((void **)&nc, o->table_size * 4));
                SQ_HIP(hipMemsetAsync(nh, 0, o->table_size * 8, ctx->stream));
                SQ_HIP(hipMemsetAsync(nc, 0, o->table_size * 4, ctx->stream));
                hipLaunchKernelGGL(k_ovr_compact, dim3(blocks_for(o->table_size)), dim3(256), 0, ctx->stream, o->d_hashes, o->d_counts,
                                   o->table_size, nh, nc);
                SQ_HIP(hipStreamSynchronize(ctx->stream));
                (void)hipFree(o->d_hashes); (void)hipFree(o->d_counts);
                o->d_hashes = nh; o->d_counts = nc;
                P.hashes = nh; P.counts = nc;
                if (o->d_ranks) { (void)hipFree(o->d_ranks); o->d_ranks = nullptr; }
                o->occupied_valid = false;
            }
        }
    }
    return SQ_OK;
}

SQ_EXPORT int sq_overrep_set_shard(sq_overrep *o, uint64_t first_record_index)
{
    if (o->number_of_sequences) {
        sq_set_error("sq_overrep_set_shard: call it before the first record array");
        return SQ_ERR_VALUE;
    }
    o->shard = true;
    o->first_record = first_record_index;
    if (!o->d_ranks) SQ_HIP(hipMalloc((void **)&o->d_ranks, o->table_size * 8));
    SQ_HIP(hipMemsetAsync(o->d_ranks, 0xFF, o->table_size * 8, o->ctx->stream));
    return SQ_OK;
}

/* first min(unique keys, max_unique) keys of this shard in rank order, device arrays */
SQ_EXPORT int64_t sq_overrep_shard_candidates(sq_overrep *o, uint64_t *d_hashes, uint64_t *d_ranks, size_t cap)
{
    sq_ctx *ctx = o->ctx;
    if (!o->shard) { sq_set_error("sq_overrep_shard_candidates: not in shard mode"); return SQ_ERR_VALUE; }
    const uint64_t n = o->n_unique_host, m = std::min<uint64_t>(n, o->max_unique);
    if (!d_hashes || cap < m || m == 0) return (int64_t)m;
    unsigned long long *d_buf = nullptr, *d_n = nullptr;
    SQ_HIP(hipMalloc((void **)&d_buf, 4 * n * 8));
    SQ_HIP(hipMalloc((void **)&d_n, 8));
    SQ_HIP(hipMemsetAsync(d_n, 0, 8, ctx->stream));
    unsigned long long *rank_in = d_buf, *hash_in = d_buf + n, *rank_out = d_buf + 2 * n, *hash_out = d_buf + 3 * n;
    hipLaunchKernelGGL(k_ovr_collect_all, dim3(blocks_for(o->table_size)), dim3(256), 0, ctx->stream,
                       o->d_hashes, o->d_ranks, o->table_size, rank_in, hash_in, d_n);
    size_t temp_bytes = 0;
    void *d_temp = nullptr;
    SQ_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, rank_in, rank_out, hash_in, hash_out, (int)n, 0,
                                              64, ctx->stream));
    SQ_HIP(hipMalloc(&d_temp, temp_bytes ? temp_bytes : 8));
    SQ_HIP(hipcub::DeviceRadixSort::SortPairs(d_temp, temp_bytes, rank_in, rank_out, hash_in, hash_out, (int)n, 0,
                                              64, ctx->stream));
    SQ_HIP(hipMemcpyAsync(d_hashes, hash_out, m * 8, hipMemcpyDeviceToDevice, ctx->stream));
    SQ_HIP(hipMemcpyAsync(d_ranks, rank_out, m * 8, hipMemcpyDeviceToDevice, ctx->stream));
    SQ_HIP(hipStreamSynchronize(ctx->stream));
    (void)hipFree(d_temp); (void)hipFree(d_buf); (void)hipFree(d_n);
    return (int64_t)m;
}

/* the job's table keys: the first max_unique distinct hashes of the shards' candidates
 * (concatenated in any order) by rank; device arrays in, device array out */
SQ_EXPORT int64_t sq_overrep_shard_select(sq_overrep *o, const uint64_t *d_hashes, const uint64_t *d_ranks,
                                          size_t n, uint64_t *d_selected, size_t cap)
{
    sq_ctx *ctx = o->ctx;
    if (n == 0) return 0;
    uint64_t tsize = 1;
    while (tsize < 2 * n) tsize <<= 1;
    unsigned long long *d_buf = nullptr, *d_keys = nullptr, *d_pos = nullptr, *d_num = nullptr;
    unsigned char *d_flags = nullptr;
    SQ_HIP(hipMalloc((void **)&d_buf, 3 * n * 8));
    SQ_HIP(hipMalloc((void **)&d_keys, tsize * 8));
    SQ_HIP(hipMalloc((void **)&d_pos, tsize * 8));
    SQ_HIP(hipMalloc((void **)&d_flags, n));
    SQ_HIP(hipMalloc((void **)&d_num, 8));
    unsigned long long *rank_out = d_buf, *hash_out = d_buf + n, *sel = d_buf + 2 * n;
    size_t temp_bytes = 0, temp2 = 0;
    void *d_temp = nullptr;
    SQ_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, (const unsigned long long *)d_ranks, rank_out,
                                              (const unsigned long long *)d_hashes, hash_out, (int)n, 0, 64,
                                              ctx->stream));
    SQ_HIP(hipcub::DeviceSelect::Flagged(nullptr, temp2, hash_out, d_flags, sel, d_num, (int)n, ctx->stream));
    temp_bytes = std::max(temp_bytes, temp2);
    SQ_HIP(hipMalloc(&d_temp, temp_bytes ? temp_bytes : 8));
    SQ_HIP(hipcub::DeviceRadixSort::SortPairs(d_temp, temp_bytes, (const unsigned long long *)d_ranks, rank_out,
                                              (const unsigned long long *)d_hashes, hash_out, (int)n, 0, 64,
                                              ctx->stream));
    SQ_HIP(hipMemsetAsync(d_keys, 0, tsize * 8, ctx->stream));
    SQ_HIP(hipMemsetAsync(d_pos, 0xFF, tsize * 8, ctx->stream));
    hipLaunchKernelGGL(k_ovr_first_pos, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, hash_out, (uint64_t)n,
                       d_keys, d_pos, tsize - 1);
    hipLaunchKernelGGL(k_ovr_flag_first, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, hash_out, (uint64_t)n,
                       d_keys, d_pos, tsize - 1, d_flags);
    SQ_HIP(hipcub::DeviceSelect::Flagged(d_temp, temp_bytes, hash_out, d_flags, sel, d_num, (int)n, ctx->stream));
    SQ_HIP(hipMemcpyAsync(&ctx->pinned[25], d_num, 8, hipMemcpyDeviceToHost, ctx->stream));
    SQ_HIP(hipStreamSynchronize(ctx->stream));
    const uint64_t m = std::min<uint64_t>(ctx->pinned[25], o->max_unique);
    int64_t result = (int64_t)m;
    if (d_selected && cap >= m) {
        if (m) SQ_HIP(hipMemcpy(d_selected, sel, m * 8, hipMemcpyDeviceToDevice));
    } else if (d_selected) {
        sq_set_error("sq_overrep_shard_select: destination too small");
        result = SQ_ERR_VALUE;
    }
    (void)hipFree(d_temp); (void)hipFree(d_buf); (void)hipFree(d_keys); (void)hipFree(d_pos);
    (void)hipFree(d_flags); (void)hipFree(d_num);
    return result;
}

/* this shard's count of every selected hash (0 when the shard never saw it) */
SQ_EXPORT int sq_overrep_shard_lookup(sq_overrep *o, const uint64_t *d_hashes, size_t n, uint64_t *d_counts)
{
    if (n == 0) return SQ_OK;
    hipLaunchKernelGGL(k_ovr_lookup, dim3(blocks_for(n)), dim3(256), 0, o->ctx->stream, o->d_hashes, o->d_counts,
                       o->table_size - 1, (const unsigned long long *)d_hashes, (uint64_t)n,
                       (unsigned long long *)d_counts);
    SQ_HIP(hipGetLastError());
    SQ_HIP(hipStreamSynchronize(o->ctx->stream));
    return SQ_OK;
}

/* replaces the shard's state by the job's: the selected keys with their summed counts and
 * totals = {number_of_sequences, sampled_sequences, total_fragments, warning count,
 * last warning record (-1: none)}; the object leaves shard mode */
SQ_EXPORT int sq_overrep_shard_install(sq_overrep *o, const uint64_t *d_hashes, const uint64_t *d_counts, size_t n,
                                       const uint64_t *totals)
{
    sq_ctx *ctx = o->ctx;
    if (n > o->max_unique) { sq_set_error("sq_overrep_shard_install: more keys than max_unique_fragments"); return SQ_ERR_VALUE; }
    SQ_HIP(hipStreamSynchronize(ctx->stream));
    uint64_t want = 2 * o->max_unique + (1u << 16), size = 1;
    while (size < want) size <<= 1;
    if (size != o->table_size) {
        (void)hipFree(o->d_hashes); (void)hipFree(o->d_counts);
        o->d_hashes = nullptr; o->d_counts = nullptr;
        o->table_size = size;
        SQ_HIP(hipMalloc((void **)&o->d_hashes, size * 8));
        SQ_HIP(hipMalloc((void **)&o->d_counts, size * 4));
    }
    if (o->d_ranks) { (void)hipFree(o->d_ranks); o->d_ranks = nullptr; }
    SQ_HIP(hipMemsetAsync(o->d_hashes, 0, size * 8, ctx->stream));
    SQ_HIP(hipMemsetAsync(o->d_counts, 0, size * 4, ctx->stream));
    if (n)
        hipLaunchKernelGGL(k_ovr_install, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, o->d_hashes, o->d_counts,
                           size - 1, (const unsigned long long *)d_hashes, (const unsigned long long *)d_counts,
                           (uint64_t)n);
    const unsigned long long scalars[4] = {n, totals[2], totals[3], totals[4]};
    SQ_HIP(hipMemcpyAsync(o->d_scalars, scalars, 32, hipMemcpyHostToDevice, ctx->stream));
    SQ_HIP(hipStreamSynchronize(ctx->stream));
    o->number_of_sequences = totals[0];
    o->sampled_sequences = totals[1];
    o->n_unique_host = n;
    o->full = n >= o->max_unique;
    o->occupied_valid = false;
    if (o->d_occupied) { (void)hipFree(o->d_occupied); o->d_occupied = nullptr; }   /* (the table may have another size now) */
    o->shard = false;
    o->first_record = 0;
    return SQ_OK;
}

SQ_EXPORT int sq_overrep_add(sq_overrep *o, const uint8_t *buf, size_t buf_len, const sq_meta *metas, size_t n)
{
    sq_batch *b = sq_batch_upload(o->ctx, buf, buf_len, metas, n);
    if (!b) return SQ_ERR_MEMORY;
    int rc = sq_overrep_add_batch(o, b);
    sq_batch_free(b);
    return rc;
}

SQ_EXPORT int sq_overrep_flush(sq_overrep *o) { return sq_synchronize(o->ctx); }
SQ_EXPORT uint64_t sq_overrep_number_of_sequences(sq_overrep *o) { return o->number_of_sequences; }
SQ_EXPORT uint64_t sq_overrep_sampled_sequences(sq_overrep *o) { return o->sampled_sequences; }

static uint64_t ovr_scalar(sq_overrep *o, int i)
{
    unsigned long long v = 0;
    (void)hipStreamSynchronize(o->ctx->stream);
    (void)hipMemcpy(&v, o->d_scalars + i, 8, hipMemcpyDeviceToHost);
    return v;
}

SQ_EXPORT uint64_t sq_overrep_collected_unique_fragments(sq_overrep *o) { return ovr_scalar(o, 0); }
SQ_EXPORT uint64_t sq_overrep_total_fragments(sq_overrep *o) { return ovr_scalar(o, 1); }
SQ_EXPORT uint64_t sq_overrep_warning_count(sq_overrep *o) { return ovr_scalar(o, 2); }
SQ_EXPORT int64_t sq_overrep_last_warning_record(sq_overrep *o) { return (int64_t)ovr_scalar(o, 3); }

SQ_EXPORT int64_t sq_overrep_get_counts(sq_overrep *o, uint64_t *kmers, uint64_t *counts, size_t cap)
{
    SQ_HIP(hipStreamSynchronize(o->ctx->stream));
    std::vector<unsigned long long> h(o->table_size);
    std::vector<unsigned int> c(o->table_size);
    SQ_HIP(hipMemcpy(h.data(), o->d_hashes, o->table_size * 8, hipMemcpyDeviceToHost));
    SQ_HIP(hipMemcpy(c.data(), o->d_counts, o->table_size * 4, hipMemcpyDeviceToHost));
    size_t n = 0;
    for (uint64_t i = 0; i < o->table_size; i++) {
        if (h[i] == 0 || c[i] == 0) continue;
        if (kmers && n < cap) {
            kmers[n] = wanghash64_inverse(h[i]); /* :4042 */
            counts[n] = c[i];
        }
        n++;
    }
    return (int64_t)n;
}

/* ---- DedupEstimator ------------------------------------------------------------------ */

struct sq_dedup {
    sq_ctx *ctx;
    uint64_t modulo_bits = 0, table_size, max_stored, stored = 0;
    uint64_t front_len, back_len, front_off, back_off;
    /* The table lives in HBM (d_hash / d_count; round 6) and is copied to the host when somebody asks for it (getters,
       export) or when a piece has to take the sequential loop; an estimator without a context (the head of a gather
       merge in tests/test_dedup_gather_cpu.py) has the host's copy only.  host_valid / dev_valid: which copy is current */
    std::vector<uint64_t> hash;
    std::vector<uint32_t> count;
    bool host_valid = true, dev_valid = false;
    unsigned long long *d_hash = nullptr, *d_hash2 = nullptr;
    unsigned int *d_count = nullptr, *d_count2 = nullptr, *d_tok = nullptr;
    /* the one entry that may sit outside the probe run of its hash: the arrival that caused the last rebuild, if it was
       inserted (placed with the old bit count, :4453) and still passes the mask */
    uint64_t odd_hash = 0;
    bool odd_valid = false;
    uint64_t host_pieces = 0, device_pieces = 0;   /* pieces that took the sequential loop / the parallel steps */
    std::vector<uint8_t> store; /* the fingerprint buffer the reference reuses */
    std::vector<unsigned long long> h_hashes;   /* host side of a piece of survivors (dedup_host_piece): grow-only */
    /* deferred mode (a shard of a multi-GPU job): add_* only hashes; the hashes
       stay in HBM until sq_dedup_resolve() runs the insertion tail over them, after the
       state of the shard in front has been imported */
    bool deferred = false;
    unsigned long long *d_stream = nullptr;
    size_t stream_cap = 0;
    uint64_t stream_n = 0;
    std::vector<uint8_t> store_known; /* store bytes written since the shard began */
    std::vector<uint8_t> store_in;    /* the store the shard starts from */
    struct Unresolved {               /* a short pair whose fingerprint shows bytes of store_in */
        uint64_t pos, seed;
        std::vector<uint8_t> bytes, known;
    };
    std::vector<Unresolved> unresolved;
    /* gather merge: the resident hashes that pass a mask of pass_bits bits, in read order */
    std::vector<unsigned long long> pass;
    uint64_t pass_bits = 0;
    bool pass_valid = false;
};

SQ_EXPORT sq_dedup *sq_dedup_new(sq_ctx *ctx, int64_t max_stored_fingerprints, int64_t front_sequence_length,
                                 int64_t back_sequence_length, int64_t front_sequence_offset,
                                 int64_t back_sequence_offset)
{
    /* DedupEstimator__new__, _qcmodule.c:4301-4380 */
    if (max_stored_fingerprints < 100) {
        sq_set_error("max_stored_fingerprints must be at least 100, not %lld", (long long)max_stored_fingerprints);
        return nullptr;
    }
    const char *names[4] = {"front_sequence_length", "back_sequence_length", "front_sequence_offset",
                            "back_sequence_offset"};
    const int64_t vals[4] = {front_sequence_length, back_sequence_length, front_sequence_offset,
                             back_sequence_offset};
    for (int i = 0; i < 4; i++) {
        if (vals[i] < 0) {
            sq_set_error("%s must be at least 0, got %lld.", names[i], (long long)vals[i]);
            return nullptr;
        }
    }
    if (front_sequence_length + back_sequence_length == 0) {
        sq_set_error("The sum of front_sequence_length and back_sequence_length must be at least 0");
        return nullptr;
    }
    sq_dedup *d = new sq_dedup();
    d->ctx = ctx;
    const uint64_t bits = (uint64_t)(log2(max_stored_fingerprints * 1.5) + 1);
    d->table_size = 1ULL << bits;
    d->max_stored = max_stored_fingerprints;
    d->front_len = front_sequence_length; d->back_len = back_sequence_length;
    d->front_off = front_sequence_offset; d->back_off = back_sequence_offset;
    d->hash.assign(d->table_size, 0);
    d->count.assign(d->table_size, 0);
    d->store.assign(d->front_len + d->back_len, 0);
    d->store_known.assign(d->front_len + d->back_len, 1);
    d->store_in = d->store;
    return d;
}

SQ_EXPORT void sq_dedup_free(sq_dedup *d)
{
    if (!d) return;
    if (d->ctx && (d->d_stream || d->d_hash)) (void)hipStreamSynchronize(d->ctx->stream);
    if (d->d_stream) (void)hipFree(d->d_stream);
    for (void *p : {(void *)d->d_hash, (void *)d->d_hash2, (void *)d->d_count, (void *)d->d_count2, (void *)d->d_tok})
        if (p) (void)hipFree(p);
    delete d;
}

namespace {

/* the current table on the host (getters, export, the sequential loop) */
int dedup_to_host(sq_dedup *d)
{
    if (d->host_valid) return SQ_OK;
    SQ_HIP(hipMemcpyAsync(d->hash.data(), d->d_hash, d->table_size * 8, hipMemcpyDeviceToHost, d->ctx->stream));
    SQ_HIP(hipMemcpyAsync(d->count.data(), d->d_count, d->table_size * 4, hipMemcpyDeviceToHost, d->ctx->stream));
    SQ_HIP(hipStreamSynchronize(d->ctx->stream));
    d->host_valid = true;
    return SQ_OK;
}

/* the current table in HBM (the parallel steps) */
int dedup_to_device(sq_dedup *d)
{
    if (d->dev_valid) return SQ_OK;
    if (!d->d_hash) {
        SQ_HIP(hipMalloc((void **)&d->d_hash, d->table_size * 8));
        SQ_HIP(hipMalloc((void **)&d->d_hash2, d->table_size * 8));
        SQ_HIP(hipMalloc((void **)&d->d_count, d->table_size * 4));
        SQ_HIP(hipMalloc((void **)&d->d_count2, d->table_size * 4));
        SQ_HIP(hipMalloc((void **)&d->d_tok, d->table_size * 4));
    }
    SQ_HIP(hipMemcpyAsync(d->d_hash, d->hash.data(), d->table_size * 8, hipMemcpyHostToDevice, d->ctx->stream));
    SQ_HIP(hipMemcpyAsync(d->d_count, d->count.data(), d->table_size * 4, hipMemcpyHostToDevice, d->ctx->stream));
    SQ_HIP(hipStreamSynchronize(d->ctx->stream));   /* the vectors may change behind this call */
    d->dev_valid = true;
    return SQ_OK;
}

/* which entry of the host's table cannot be reached from the slot its hash starts at (an imported state does not
 * say): walking the run in front of every entry back to an empty slot is the table's size times a short run */
void dedup_find_odd(sq_dedup *d)
{
    const uint64_t mask = d->table_size - 1, bits = d->modulo_bits;
    d->odd_valid = false;
    for (uint64_t i = 0; i < d->table_size; i++) {
        if (!d->count[i]) continue;
        const uint64_t h = d->hash[i];
        if (bits && (h & ((1ULL << bits) - 1))) continue;   /* does not pass the mask: never arrives again */
        bool reachable = true;
        for (uint64_t j = (h >> bits) & mask; j != i; j = (j + 1) & mask)
            if (!d->count[j]) { reachable = false; break; }
        if (!reachable) { d->odd_hash = h; d->odd_valid = true; }
    }
}

/* DedupEstimator_increment_modulo, _qcmodule.c:4382-4423 (host copy) */
void dedup_rebuild(sq_dedup *d)
{
    const uint64_t bits = d->modulo_bits + 1, ignore = (1ULL << bits) - 1, mask = d->table_size - 1;
    std::vector<uint64_t> nh(d->table_size, 0);
    std::vector<uint32_t> nc(d->table_size, 0);
    uint64_t kept = 0;
    for (uint64_t i = 0; i < d->table_size; i++) {
        if (d->count[i] == 0 || (d->hash[i] & ignore)) continue;
        uint64_t j = (d->hash[i] >> bits) & mask;
        while (nc[j] != 0) j = (j + 1) & mask;
        nh[j] = d->hash[i];
        nc[j] = d->count[i];
        kept++;
    }
    d->hash.swap(nh);
    d->count.swap(nc);
    d->modulo_bits = bits;
    d->stored = kept;
}

/* the tail of DedupEstimator_add_fingerprint, _qcmodule.c:4430-4459, quirks included
 * (the pre-rebuild bit count indexes the triggering hash), on the host copy */
inline void dedup_insert(sq_dedup *d, uint64_t h)
{
    const uint64_t bits = d->modulo_bits;
    if (h & ((1ULL << bits) - 1)) return;
    bool rebuilt = false;
    if (d->stored >= d->max_stored) { dedup_rebuild(d); rebuilt = true; d->odd_valid = false; }
    const uint64_t mask = d->table_size - 1;
    uint64_t i = (h >> bits) & mask;
    for (;;) {
        if (d->count[i] == 0) {
            d->hash[i] = h; d->count[i] = 1; d->stored++;
            if (rebuilt && (h & ((1ULL << d->modulo_bits) - 1)) == 0) { d->odd_hash = h; d->odd_valid = true; }
            return;
        }
        if (d->hash[i] == h) { d->count[i]++; return; }
        i = (i + 1) & mask;
    }
}

int fetch_sequence(sq_batch *b, uint64_t r, std::vector<uint8_t> &out)
{
    sq_meta m;
    if (!b->h_metas.empty()) m = b->h_metas[r];
    else SQ_HIP(hipMemcpy(&m, b->d_metas + r, sizeof(sq_meta), hipMemcpyDeviceToHost));
    out.resize(m.sequence_length);
    if (!m.sequence_length) return SQ_OK;
    const uint64_t off = m.record_start + m.sequence_offset;
    if (!b->h_buf.empty()) memcpy(out.data(), b->h_buf.data() + off, m.sequence_length);
    else SQ_HIP(hipMemcpy(out.data(), b->d_buf + off, m.sequence_length, hipMemcpyDeviceToHost));
    return SQ_OK;
}

/* what pair r writes into the fingerprint store (:4503-4514); returns bytes written */
int pair_store_bytes(sq_dedup *d, sq_batch *b1, sq_batch *b2, uint64_t r, std::vector<uint8_t> &bytes,
                     uint64_t *total_length = nullptr)
{
    std::vector<uint8_t> s1, s2;
    int rc = fetch_sequence(b1, r, s1);
    if (rc) return rc;
    rc = fetch_sequence(b2, r, s2);
    if (rc) return rc;
    const uint64_t L1 = s1.size(), L2 = s2.size();
    if (total_length) *total_length = L1 + L2;
    const uint64_t fl = std::min<uint64_t>(d->front_len, L1), fo = std::min<uint64_t>(d->front_off, L1 - fl);
    const uint64_t bl = std::min<uint64_t>(d->back_len, L2), bo = std::min<uint64_t>(d->back_off, L2 - bl);
    bytes.assign(s1.begin() + fo, s1.begin() + fo + fl);
    bytes.insert(bytes.end(), s2.begin() + bo, s2.begin() + bo + bl);
    return SQ_OK;
}

/* One piece through the reference's loop on the host's copy of the table: survivors sel[0, n_sel) of `hashes` (both
 * on the device), in order.  For estimators without a device table (deferred shards use the parallel steps too) and
 * for the pieces the parallel steps must not take (the odd entry's hash arrived; SQ_DEDUP_SEQUENTIAL=1). */
int dedup_host_piece(sq_dedup *d, const unsigned long long *d_hashes, const unsigned long long *d_sel, uint64_t n_sel)
{
    sq_ctx *ctx = d->ctx;
    int rc = dedup_to_host(d);
    if (rc) return rc;
    if (d->h_hashes.size() < n_sel) d->h_hashes.resize(n_sel);
    unsigned long long *d_kh = (unsigned long long *)sq_scratch(ctx, 10, n_sel * 8);
    if (!d_kh) { sq_set_error("out of device memory"); return SQ_ERR_MEMORY; }
    hipLaunchKernelGGL(k_dedup_gather, dim3(blocks_for(n_sel)), dim3(256), 0, ctx->stream, d_sel, n_sel, d_hashes,
                       (const unsigned char *)nullptr, d_kh, (unsigned char *)nullptr);
    SQ_HIP(hipMemcpyAsync(d->h_hashes.data(), d_kh, n_sel * 8, hipMemcpyDeviceToHost, ctx->stream));
    SQ_HIP(hipStreamSynchronize(ctx->stream));
    const unsigned long long *hashes = d->h_hashes.data();
    for (uint64_t e = 0; e < n_sel; e++) {
        if (e + 12 < n_sel) { /* the slot a hash lands in is known ahead: hide the table's cache misses */
            const uint64_t slot = (hashes[e + 12] >> d->modulo_bits) & (d->table_size - 1);
            __builtin_prefetch(&d->count[slot]);
            __builtin_prefetch(&d->hash[slot]);
        }
        dedup_insert(d, hashes[e]);
    }
    d->dev_valid = false;
    d->host_pieces++;
    return SQ_OK;
}

/* the device's copy of a small result, behind everything queued on the stream */
int dedup_read_back(sq_ctx *ctx, const void *d_src, void *dst, size_t bytes)
{
    SQ_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    SQ_HIP(hipStreamSynchronize(ctx->stream));
    return SQ_OK;
}

/* indices [0,n) for which pred holds, in order, on the device (a counting iterator: 0 .. n-1 is never written out);
 * the count on the host.  The result lives in the context's scratch slot `slot0` (valid until the next call that
 * names it): a hipMalloc / hipFree pair per array and batch costs more than the selection */
template <typename Pred>
int ordered_select(sq_ctx *ctx, uint64_t n, Pred pred, int slot0, unsigned long long **d_out, uint64_t *count)
{
    size_t temp_bytes = 0;
    *d_out = nullptr;
    *count = 0;
    if (n == 0) return SQ_OK;
    unsigned long long *d_sel = (unsigned long long *)sq_scratch(ctx, slot0, n * 8);
    unsigned long long *d_num = (unsigned long long *)sq_scratch(ctx, 8, 8);
    if (!d_sel || !d_num) { sq_set_error("out of device memory"); return SQ_ERR_MEMORY; }
    hipcub::CountingInputIterator<unsigned long long> it(0ull);
    SQ_HIP(hipcub::DeviceSelect::If(nullptr, temp_bytes, it, d_sel, d_num, (int)n, pred, ctx->stream));
    void *d_temp = sq_scratch(ctx, 9, temp_bytes ? temp_bytes : 8);
    if (!d_temp) { sq_set_error("out of device memory"); return SQ_ERR_MEMORY; }
    SQ_HIP(hipcub::DeviceSelect::If(d_temp, temp_bytes, it, d_sel, d_num, (int)n, pred, ctx->stream));
    SQ_HIP(hipMemcpyAsync(&ctx->pinned[16], d_num, 8, hipMemcpyDeviceToHost, ctx->stream));
    SQ_HIP(hipStreamSynchronize(ctx->stream));
    *count = ctx->pinned[16];
    *d_out = d_sel;
    return SQ_OK;
}

/* The stream of hashes [0,n) on the device, in read order, through the estimator (the header above k_dd_classify has
 * the plan).  Only hashes that pass the mask in force can matter, and the mask only ever gets stricter (H3): a piece is
 * filtered with the mask in force when it starts and ends in front of the arrival that rebuilds. */
int dedup_process(sq_dedup *d, const unsigned long long *d_hashes, uint64_t n)
{
    sq_ctx *ctx = d->ctx;
    const uint64_t T = d->table_size, tmask = T - 1;
    const bool force_host = sq_knobs().dedup_sequential;
    unsigned quick = 0;   /* pieces in a row that ended at their first survivors: hashes whose low bits are not spread */
    int rc;
    for (uint64_t off = 0; off < n;) {
        const int64_t need = (int64_t)d->max_stored - (int64_t)d->stored;   /* new hashes until the table is full */
        const uint64_t bits = d->modulo_bits;
        /* about one read in 2^bits survives; behind `need` new ones the piece ends anyway */
        const uint64_t want = (uint64_t)std::max<int64_t>(need, 0) * 5 / 4 + 65536;
        const uint64_t piece = std::min<uint64_t>({n - off, want << std::min<uint64_t>(bits, 20), (uint64_t)1 << 30});
        const unsigned long long *ph = d_hashes + off;
        unsigned long long *d_sel = nullptr;
        uint64_t n_sel = 0;
        rc = ordered_select(ctx, piece, DedupKeep{bits ? (1ULL << bits) - 1 : 0ull, ph, nullptr}, 7, &d_sel, &n_sel);
        if (rc) return rc;
        if (n_sel == 0) { off += piece; continue; }
        if (force_host || quick > 16) {
            rc = dedup_host_piece(d, ph, d_sel, n_sel);
            if (rc) return rc;
            off += piece;
            continue;
        }
        rc = dedup_to_device(d);
        if (rc) return rc;
        unsigned long long *d_res = (unsigned long long *)sq_scratch(ctx, 32, 64);   /* 0: n-th first arrival, 1: inserted, 2: kept, 3-4: trigger, 5 (as u32): flags */
        unsigned int *d_slot = (unsigned int *)sq_scratch(ctx, 33, n_sel * 4);
        if (!d_res || !d_slot) { sq_set_error("out of device memory"); return SQ_ERR_MEMORY; }
        SQ_HIP(hipMemsetAsync(d_res, 0, 64, ctx->stream));
        uint64_t e_end = n_sel;       /* survivors [0, e_end) arrive while the table has room */
        bool trigger = false;
        unsigned long long *d_miss = nullptr, *d_key = nullptr, *d_skey = nullptr;
        unsigned long long *d_val = nullptr, *d_sval = nullptr;
        unsigned int *d_newc = nullptr;
        uint64_t n_miss = 0;
        if (need <= 0) {              /* full already: the first survivor rebuilds */
            e_end = 0;
            trigger = true;
        } else {
            hipLaunchKernelGGL(k_dd_classify, dim3(blocks_for(n_sel)), dim3(256), 0, ctx->stream, ph, d_sel, n_sel, d->d_hash,
                               d->d_count, bits, tmask, (unsigned long long)d->odd_hash, d->odd_valid ? 1 : 0, d_slot,
                               (unsigned int *)(d_res + 5));
            rc = ordered_select(ctx, n_sel, DedupIsMiss{d_slot}, 34, &d_miss, &n_miss);
            if (rc) return rc;
            uint64_t res[8];
            rc = dedup_read_back(ctx, d_res, res, sizeof res);
            if (rc) return rc;
            if ((unsigned int)res[5]) {   /* the odd entry's hash arrived and a lookup now cannot say what it will find then */
                rc = dedup_host_piece(d, ph, d_sel, n_sel);
                if (rc) return rc;
                off += piece;
                continue;
            }
            if (n_miss) {
                d_key = (unsigned long long *)sq_scratch(ctx, 35, n_miss * 8);
                d_skey = (unsigned long long *)sq_scratch(ctx, 36, n_miss * 8);
                d_val = (unsigned long long *)sq_scratch(ctx, 37, n_miss * 3 * 8);   /* ranks, sorted ranks; first arrivals / new counts, positions */
                if (!d_key || !d_skey || !d_val) { sq_set_error("out of device memory"); return SQ_ERR_MEMORY; }
                d_sval = d_val + n_miss;
                d_newc = (unsigned int *)(d_val + 2 * n_miss);
                unsigned int *d_pos = d_newc + n_miss;
                hipLaunchKernelGGL(k_dd_miss_keys, dim3(blocks_for(n_miss)), dim3(256), 0, ctx->stream, ph, d_sel, d_miss, n_miss,
                                   d_key, d_val);
                size_t temp_bytes = 0;
                SQ_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, d_key, d_skey, d_val, d_sval, (int)n_miss, 0, 64, ctx->stream));
                void *d_temp = sq_scratch(ctx, 38, temp_bytes ? temp_bytes : 8);
                if (!d_temp) { sq_set_error("out of device memory"); return SQ_ERR_MEMORY; }
                SQ_HIP(hipcub::DeviceRadixSort::SortPairs(d_temp, temp_bytes, d_key, d_skey, d_val, d_sval, (int)n_miss, 0, 64, ctx->stream));
                if (n_miss >= (uint64_t)need) {   /* enough arrivals that were not found to fill the table, if enough of them are new */
                    unsigned int *d_first = d_newc;   /* the counts are written behind this use */
                    hipLaunchKernelGGL(k_dd_heads, dim3(blocks_for(n_miss)), dim3(256), 0, ctx->stream, d_skey, d_sval, n_miss, d_first);
                    size_t scan_bytes = 0;
                    SQ_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, scan_bytes, d_first, d_pos, (int)n_miss, ctx->stream));
                    void *d_scan = sq_scratch(ctx, 38, scan_bytes ? scan_bytes : 8);
                    if (!d_scan) { sq_set_error("out of device memory"); return SQ_ERR_MEMORY; }
                    SQ_HIP(hipcub::DeviceScan::InclusiveSum(d_scan, scan_bytes, d_first, d_pos, (int)n_miss, ctx->stream));
                    SQ_HIP(hipMemsetAsync(d_res, 0xFF, 8, ctx->stream));
                    hipLaunchKernelGGL(k_dd_find_nth, dim3(blocks_for(n_miss)), dim3(256), 0, ctx->stream, d_first, d_pos, n_miss,
                                       (unsigned int)need, d_res);
                    /* d_miss[r]: the survivor that fills the table; the one behind it rebuilds */
                    rc = dedup_read_back(ctx, d_res, res, 8);
                    if (rc) return rc;
                    if (res[0] != ~0ull) {
                        unsigned long long e_fill = 0;
                        rc = dedup_read_back(ctx, d_miss + res[0], &e_fill, 8);
                        if (rc) return rc;
                        if (e_fill + 1 < n_sel) { e_end = e_fill + 1; trigger = true; }
                    }
                    SQ_HIP(hipMemsetAsync(d_res, 0, 8, ctx->stream));
                }
            }
            /* survivors [0, e_end): the found ones count, the first arrivals of new hashes take their slots */
            hipLaunchKernelGGL(k_dd_apply_found, dim3(blocks_for(e_end)), dim3(256), 0, ctx->stream, d_slot, e_end, d->d_count);
            if (n_miss) {
                hipLaunchKernelGGL(k_dd_tok_init, dim3(blocks_for(T)), dim3(256), 0, ctx->stream, d->d_count, d->d_tok, T);
                hipLaunchKernelGGL(k_dd_insert, dim3(blocks_for(n_miss)), dim3(256), 0, ctx->stream, d_skey, d_sval, n_miss, d_miss,
                                   e_end, bits, tmask, d->d_tok, d_newc, d_res + 1);
                hipLaunchKernelGGL(k_dd_finalize, dim3(blocks_for(T)), dim3(256), 0, ctx->stream, d->d_tok, T, d_key, d_newc,
                                   d->d_hash, d->d_count);
            }
            d->host_valid = false;
        }
        uint64_t trig_at = 0;
        if (trigger) {
            /* DedupEstimator_increment_modulo, then the arrival itself by the old bit count */
            SQ_HIP(hipMemsetAsync(d->d_tok, 0xFF, T * 4, ctx->stream));
            hipLaunchKernelGGL(k_dd_rebuild_insert, dim3(blocks_for(T)), dim3(256), 0, ctx->stream, d->d_hash, d->d_count, T,
                               bits + 1, d->d_tok, d_res + 2);
            hipLaunchKernelGGL(k_dd_rebuild_finalize, dim3(blocks_for(T)), dim3(256), 0, ctx->stream, d->d_tok, T, d->d_hash,
                               d->d_count, d->d_hash2, d->d_count2);
            std::swap(d->d_hash, d->d_hash2);
            std::swap(d->d_count, d->d_count2);
            hipLaunchKernelGGL(k_dd_trigger, dim3(1), dim3(1), 0, ctx->stream, ph, d_sel, e_end, bits, tmask, d->d_hash, d->d_count,
                               d_res + 3);
            SQ_HIP(hipMemcpyAsync(&trig_at, d_sel + e_end, 8, hipMemcpyDeviceToHost, ctx->stream));
            d->host_valid = false;
        }
        SQ_HIP(hipGetLastError());
        uint64_t res[8];
        rc = dedup_read_back(ctx, d_res, res, sizeof res);
        if (rc) return rc;
        d->device_pieces++;
        if (!trigger) {
            d->stored += res[1];
            off += piece;
            quick = 0;
            continue;
        }
        d->modulo_bits = bits + 1;
        d->stored = res[2] + res[4];
        d->odd_hash = res[3];
        d->odd_valid = res[4] != 0 && (res[3] & ((1ULL << (bits + 1)) - 1)) == 0;
        quick = e_end < 4 ? quick + 1 : 0;
        off += trig_at + 1;
    }
    return SQ_OK;
}

/* the hashes of pairs shorter than the fingerprint: bytes of the store shine through (:4512-4516), so they are made
 * on the host, pair after pair in read order, the store carried from one to the next (a pair in front that is not a
 * short one rewrote all of it), and written into the batch's hashes before anything looks at them.  Leaves the
 * store as the batch's last pair leaves it. */
int dedup_patch_short_pairs(sq_dedup *d, sq_batch *b1, sq_batch *b2, unsigned long long *d_hashes,
                            const unsigned char *d_special, uint64_t n)
{
    sq_ctx *ctx = d->ctx;
    const uint64_t fp_len = d->front_len + d->back_len;
    unsigned long long *d_idx = nullptr;
    uint64_t n_special = 0;
    int rc = ordered_select(ctx, n, DedupSpecialOnly{d_special}, 7, &d_idx, &n_special);
    if (rc) return rc;
    std::vector<unsigned long long> idx(n_special), val(n_special);
    if (n_special) SQ_HIP(hipMemcpy(idx.data(), d_idx, n_special * 8, hipMemcpyDeviceToHost));
    std::vector<uint8_t> cur = d->store, w;
    uint64_t prev = UINT64_MAX;
    for (uint64_t e = 0; e < n_special; e++) {
        const uint64_t r = idx[e];
        if (r != 0 && prev != r - 1) { /* the pair in front rewrote the whole store */
            rc = pair_store_bytes(d, b1, b2, r - 1, cur);
            if (rc) return rc;
            cur.resize(fp_len);
        }
        uint64_t total = 0;
        rc = pair_store_bytes(d, b1, b2, r, w, &total);
        if (rc) return rc;
        std::copy(w.begin(), w.end(), cur.begin());
        prev = r;
        const uint8_t *sp = cur.data();
        val[e] = murmur3_x64_64([&](uint64_t i) { return sp[i]; }, fp_len, total >> 6);
    }
    if (prev != n - 1) { /* the last pair rewrote the whole store */
        rc = pair_store_bytes(d, b1, b2, n - 1, cur);
        if (rc) return rc;
        cur.resize(fp_len);
    }
    d->store = cur;
    if (n_special) {
        unsigned long long *d_pv = (unsigned long long *)sq_scratch(ctx, 10, 2 * n_special * 8);
        if (!d_pv) { sq_set_error("out of device memory"); return SQ_ERR_MEMORY; }
        SQ_HIP(hipMemcpyAsync(d_pv, idx.data(), n_special * 8, hipMemcpyHostToDevice, ctx->stream));
        SQ_HIP(hipMemcpyAsync(d_pv + n_special, val.data(), n_special * 8, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_dd_patch, dim3(blocks_for(n_special)), dim3(256), 0, ctx->stream, d_hashes, d_pv, d_pv + n_special, n_special);
        SQ_HIP(hipStreamSynchronize(ctx->stream));   /* idx and val leave with this frame */
    }
    return SQ_OK;
}

/* deferred mode: the batch's hashes join the resident stream.  Short pairs (their
 * fingerprint shows bytes of the one before, :4512-4516) are hashed here when every such
 * byte was written inside this shard, else kept until the store of the shard in front
 * is known (sq_dedup_resolve). */
int dedup_defer(sq_dedup *d, sq_batch *b1, sq_batch *b2, const unsigned long long *d_hashes,
                const unsigned char *d_special, uint64_t n)
{
    sq_ctx *ctx = d->ctx;
    const size_t want = d->stream_n + n;
    if (want > d->stream_cap) {
        int rc = sq_grow_device(ctx, &d->d_stream, &d->stream_cap, std::max(want, 2 * d->stream_cap));
        if (rc) return rc;
    }
    SQ_HIP(hipMemcpyAsync(d->d_stream + d->stream_n, d_hashes, n * 8, hipMemcpyDeviceToDevice, ctx->stream));
    if (b2) {
        const uint64_t fp_len = d->front_len + d->back_len;
        unsigned long long *d_idx = nullptr;
        uint64_t n_special = 0;
        int rc = ordered_select(ctx, n, DedupSpecialOnly{d_special}, 7, &d_idx, &n_special);
        if (rc) return rc;
        std::vector<unsigned long long> idx(n_special);
        if (n_special) SQ_HIP(hipMemcpy(idx.data(), d_idx, n_special * 8, hipMemcpyDeviceToHost));
        std::vector<uint8_t> cur = d->store, known = d->store_known, w;
        uint64_t prev = UINT64_MAX;
        for (uint64_t e = 0; e < n_special; e++) {
            const uint64_t r = idx[e];
            if (r != 0 && prev != r - 1) { /* the pair in front rewrote the whole store */
                rc = pair_store_bytes(d, b1, b2, r - 1, cur);
                if (rc) return rc;
                cur.resize(fp_len);
                known.assign(fp_len, 1);
            }
            uint64_t total = 0;
            rc = pair_store_bytes(d, b1, b2, r, w, &total);
            if (rc) return rc;
            for (uint64_t i = 0; i < w.size(); i++) { cur[i] = w[i]; known[i] = 1; }
            prev = r;
            if (std::find(known.begin(), known.end(), 0) == known.end()) {
                const uint8_t *sp = cur.data();
                const unsigned long long h = murmur3_x64_64([&](uint64_t i) { return sp[i]; }, fp_len, total >> 6);
                SQ_HIP(hipMemcpyAsync(d->d_stream + d->stream_n + r, &h, 8, hipMemcpyHostToDevice, ctx->stream));
                SQ_HIP(hipStreamSynchronize(ctx->stream));
            } else {
                d->unresolved.push_back({d->stream_n + r, total >> 6, cur, known});
            }
        }
        if (prev != n - 1) { /* the last pair rewrote the whole store */
            rc = pair_store_bytes(d, b1, b2, n - 1, cur);
            if (rc) return rc;
            cur.resize(fp_len);
            known.assign(fp_len, 1);
        }
        d->store = cur;
        d->store_known = known;
    }
    SQ_HIP(hipStreamSynchronize(ctx->stream));
    d->stream_n += n;
    return SQ_OK;
}

int dedup_run(sq_dedup *d, sq_batch *b1, sq_batch *b2)
{
    sq_ctx *ctx = d->ctx;
    const uint64_t n = b1->n;
    if (n == 0) return SQ_OK;
    unsigned long long *d_hashes = (unsigned long long *)sq_scratch(ctx, 12, n * 8);
    unsigned char *d_special = (unsigned char *)sq_scratch(ctx, 13, n);
    if (!d_hashes || !d_special) { sq_set_error("out of device memory"); return SQ_ERR_MEMORY; }
    DedupParams P{};
    P.buf1 = b1->d_buf; P.metas1 = b1->d_metas;
    P.buf2 = b2 ? b2->d_buf : nullptr; P.metas2 = b2 ? b2->d_metas : nullptr;
    P.n = n;
    P.front_len = d->front_len; P.back_len = d->back_len;
    P.front_off = d->front_off; P.back_off = d->back_off;
    P.hashes = d_hashes; P.special = d_special;
    hipLaunchKernelGGL(k_dedup_hash, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, P);
    SQ_HIP(hipGetLastError());
    int rc;
    if (d->deferred) {
        rc = dedup_defer(d, b1, b2, d_hashes, d_special, n);
    } else {
        rc = b2 ? dedup_patch_short_pairs(d, b1, b2, d_hashes, d_special, n) : SQ_OK;
        if (rc == SQ_OK) rc = dedup_process(d, d_hashes, n);
    }
    (void)hipStreamSynchronize(ctx->stream);
    return rc;
}

} // namespace

SQ_EXPORT int sq_dedup_add_batch(sq_dedup *d, sq_batch *b) { return dedup_run(d, b, nullptr); }

SQ_EXPORT int sq_dedup_add_batch_pair(sq_dedup *d, sq_batch *b1, sq_batch *b2)
{
    if (b1->n != b2->n) { /* :4606-4612 */
        sq_set_error("record_array1 and record_array2 must be of the same size. Got %zu and %zu respectively.",
                     b1->n, b2->n);
        return SQ_ERR_VALUE;
    }
    return dedup_run(d, b1, b2);
}

SQ_EXPORT int sq_dedup_add(sq_dedup *d, const uint8_t *buf, size_t buf_len, const sq_meta *metas, size_t n)
{
    sq_batch *b = sq_batch_upload(d->ctx, buf, buf_len, metas, n);
    if (!b) return SQ_ERR_MEMORY;
    int rc = sq_dedup_add_batch(d, b);
    sq_batch_free(b);
    return rc;
}

SQ_EXPORT int sq_dedup_add_pair(sq_dedup *d, const uint8_t *buf1, size_t len1, const sq_meta *metas1,
                                const uint8_t *buf2, size_t len2, const sq_meta *metas2, size_t n)
{
    sq_batch *b1 = sq_batch_upload(d->ctx, buf1, len1, metas1, n);
    sq_batch *b2 = sq_batch_upload(d->ctx, buf2, len2, metas2, n);
    int rc = (b1 && b2) ? sq_dedup_add_batch_pair(d, b1, b2) : SQ_ERR_MEMORY;
    sq_batch_free(b1);
    sq_batch_free(b2);
    return rc;
}

SQ_EXPORT int sq_dedup_flush(sq_dedup *d) { return sq_synchronize(d->ctx); }
SQ_EXPORT uint64_t sq_dedup_modulo_bits(sq_dedup *d) { return d->modulo_bits; }
SQ_EXPORT uint64_t sq_dedup_hash_table_size(sq_dedup *d) { return d->table_size; }
SQ_EXPORT uint64_t sq_dedup_tracked_sequences(sq_dedup *d) { return d->stored; }
/* pieces of the stream that went through the parallel steps on the device / through the host's sequential loop */
SQ_EXPORT uint64_t sq_dedup_device_pieces(sq_dedup *d) { return d->device_pieces; }
SQ_EXPORT uint64_t sq_dedup_host_pieces(sq_dedup *d) { return d->host_pieces; }

SQ_EXPORT int64_t sq_dedup_duplication_counts(sq_dedup *d, uint64_t *out, size_t cap)
{
    if (dedup_to_host(d)) return SQ_ERR_HIP;
    size_t n = 0;
    for (uint64_t i = 0; i < d->table_size; i++) { /* :4736-4744 slot order */
        if (!d->count[i]) continue;
        if (out && n < cap) out[n] = d->count[i];
        n++;
    }
    return (int64_t)n;
}

/* ---- DedupEstimator across shards: hash in parallel, insert in shard order --- */
SQ_EXPORT int sq_dedup_set_deferred(sq_dedup *d, int on)
{
    if (!on && d->stream_n) { sq_set_error("sq_dedup_set_deferred: unresolved hashes pending"); return SQ_ERR_VALUE; }
    if (on && !d->deferred) {
        d->store_in = d->store;
        d->store_known.assign(d->store.size(), 0);
    }
    d->deferred = on != 0;
    return SQ_OK;
}

SQ_EXPORT uint64_t sq_dedup_pending(sq_dedup *d) { return d->stream_n; }

/* runs the insertion tail over the shard's resident hashes, in read order */
SQ_EXPORT int sq_dedup_resolve(sq_dedup *d)
{
    sq_ctx *ctx = d->ctx;
    const uint64_t fp_len = d->front_len + d->back_len;
    for (auto &u : d->unresolved) {
        for (uint64_t i = 0; i < fp_len; i++)
            if (!u.known[i]) u.bytes[i] = d->store_in[i];
        const uint8_t *sp = u.bytes.data();
        const unsigned long long h = murmur3_x64_64([&](uint64_t i) { return sp[i]; }, fp_len, u.seed);
        SQ_HIP(hipMemcpy(d->d_stream + u.pos, &h, 8, hipMemcpyHostToDevice));
    }
    d->unresolved.clear();
    for (uint64_t i = 0; i < fp_len; i++)
        if (!d->store_known[i]) d->store[i] = d->store_in[i];
    d->store_known.assign(fp_len, d->deferred ? 0 : 1);
    d->store_in = d->store;
    if (d->stream_n) {
        int rc = dedup_process(d, d->d_stream, d->stream_n);
        if (rc) return rc;
    }
    SQ_HIP(hipStreamSynchronize(ctx->stream));
    d->stream_n = 0;
    return SQ_OK;
}

/* the estimator's state as bytes: {magic, modulo_bits, stored, table_size, fp_len} u64,
 * store[fp_len] padded to 8, hash[table_size] u64, count[table_size] u32 */
static const uint64_t DEDUP_MAGIC = 0x3150554445445153ULL; /* "SQDEDUP1" */

SQ_EXPORT uint64_t sq_dedup_state_bytes(sq_dedup *d)
{
    const uint64_t fp_len = d->front_len + d->back_len;
    return 40 + (fp_len + 7) / 8 * 8 + d->table_size * 12;
}

SQ_EXPORT int sq_dedup_export_state(sq_dedup *d, void *out, size_t cap)
{
    if (d->stream_n) { sq_set_error("sq_dedup_export_state: resolve the pending hashes first"); return SQ_ERR_VALUE; }
    if (cap < sq_dedup_state_bytes(d)) { sq_set_error("sq_dedup_export_state: destination too small"); return SQ_ERR_VALUE; }
    if (int rc = dedup_to_host(d)) return rc;
    const uint64_t fp_len = d->front_len + d->back_len, pad = (fp_len + 7) / 8 * 8;
    uint8_t *p = (uint8_t *)out;
    const uint64_t head[5] = {DEDUP_MAGIC, d->modulo_bits, d->stored, d->table_size, fp_len};
    memcpy(p, head, 40); p += 40;
    memset(p, 0, pad);
    memcpy(p, d->store.data(), fp_len); p += pad;
    memcpy(p, d->hash.data(), d->table_size * 8); p += d->table_size * 8;
    memcpy(p, d->count.data(), d->table_size * 4);
    return SQ_OK;
}

/* continues from the state another shard exported: its table, its modulo bits and its
 * fingerprint store; to be called before sq_dedup_resolve() of this shard */
SQ_EXPORT int sq_dedup_import_state(sq_dedup *d, const void *in, size_t len)
{
    const uint64_t fp_len = d->front_len + d->back_len, pad = (fp_len + 7) / 8 * 8;
    const uint8_t *p = (const uint8_t *)in;
    uint64_t head[5];
    if (len < 40) { sq_set_error("sq_dedup_import_state: truncated state"); return SQ_ERR_VALUE; }
    memcpy(head, p, 40); p += 40;
    if (head[0] != DEDUP_MAGIC || head[3] != d->table_size || head[4] != fp_len ||
        len < 40 + pad + d->table_size * 12) {
        sq_set_error("sq_dedup_import_state: state of a differently configured estimator");
        return SQ_ERR_VALUE;
    }
    d->modulo_bits = head[1];
    d->stored = head[2];
    d->store_in.assign(p, p + fp_len); p += pad;
    memcpy(d->hash.data(), p, d->table_size * 8); p += d->table_size * 8;
    memcpy(d->count.data(), p, d->table_size * 4);
    d->host_valid = true;
    d->dev_valid = false;
    dedup_find_odd(d);
    if (!d->deferred || (d->stream_n == 0 && d->unresolved.empty() &&
                         std::find(d->store_known.begin(), d->store_known.end(), 1) == d->store_known.end()))
        d->store = d->store_in; /* nothing of this shard is pending: the store is the imported one */
    return SQ_OK;
}

/* ---- DedupEstimator across shards, by gathering -------------------------------
 *
 * The relay above costs one insertion tail and one broadcast of the table (12 bytes a slot) PER SHARD, one after the
 * other.  This merge lets every shard do its part at the same time and leaves one tail to the head (the first shard):
 *
 *   1. sq_dedup_shard_store      the store bytes the shard wrote, and which of them (a few bytes: gathered)
 *   2. sq_dedup_shard_settle     with the store of the shards in front: the hashes of short pairs at the shard's start
 *                                are finished, and the shard's LOWER BOUND is counted on the device: the least b such
 *                                that at most max_stored DISTINCT hashes of the shard have b trailing zero bits
 *   3. sq_dedup_shard_passing    the shard's hashes that pass a mask of B bits, in read order, where B is the largest
 *                                lower bound of the shards in front (gathered to the head)
 *   4. sq_dedup_feed_hashes      the head, after sq_dedup_resolve() of its own shard: those hashes through the
 *                                insertion tail, shard after shard
 *   5. sq_dedup_export_state / _import_state   the head's table to everyone, once
 *
 * Why that is the sequential estimator, bit for bit.  A hash that arrives when the estimator has b modulo bits and
 * does not pass a mask of b bits changes nothing (:4433); the bits only grow; so leaving out of a shard's stream
 * every hash that fails a mask of B bits changes nothing IF the estimator has at least B bits when the shard's first
 * hash arrives.  After a stream that held the distinct hashes X the table holds every one of them that passes the
 * mask in force (it passed every earlier, weaker mask, was inserted, and survives each rebuild), and the table holds
 * at most max_stored entries -- so the bits in force are at least the shard's lower bound, whatever came before or
 * between.  "At most max_stored" has one way out (:4436-4451 rebuilds once per arriving hash: if a rebuild drops
 * nothing the insert behind it overfills the table), which takes hashes whose low bits are not spread at all; the
 * feed therefore CHECKS the premise (the head's bits against the mask the hashes were filtered with) and answers
 * SQ_DEDUP_FEED_TOO_STRICT without touching the estimator if it does not hold: the caller finishes with the relay
 * from that shard on (dist.merge_dedup does), and the result is exact either way.
 */
/* hist[c] = distinct hashes with exactly c trailing zero bits (c = 64: the hash 0) -> the least b such that at most
 * max_stored of them have b trailing zero bits or more; no device, no object: the host half of the settle step */
SQ_EXPORT uint64_t sq_dedup_lower_bound_of(const uint64_t *hist65, uint64_t max_stored)
{
    uint64_t at_least[66];
    at_least[65] = 0;
    for (int c = 64; c >= 0; c--) at_least[c] = at_least[c + 1] + hist65[c];
    uint64_t b = 0;
    while (b < 63 && at_least[b] > max_stored) b++;
    return b;
}

/* head != 0 (the job's first shard): the bytes the shard did not write are those the estimator held when deferred
 * mode began, so all of them are known */
SQ_EXPORT int64_t sq_dedup_shard_store(sq_dedup *d, int head, uint8_t *bytes, uint8_t *known, size_t cap)
{
    const uint64_t fp_len = d->front_len + d->back_len;
    if (!d->deferred) { sq_set_error("sq_dedup_shard_store: the estimator is not in deferred mode"); return SQ_ERR_VALUE; }
    if (bytes && known) {
        if (cap < fp_len) { sq_set_error("sq_dedup_shard_store: destination too small"); return SQ_ERR_VALUE; }
        for (uint64_t i = 0; i < fp_len; i++) {
            const bool k = d->store_known[i] != 0;
            bytes[i] = k || !head ? d->store[i] : d->store_in[i];
            known[i] = k || head;
        }
    }
    return (int64_t)fp_len;
}

SQ_EXPORT int sq_dedup_shard_settle(sq_dedup *d, const uint8_t *store_in, size_t len, uint64_t *lower_bound)
{
    sq_ctx *ctx = d->ctx;
    const uint64_t fp_len = d->front_len + d->back_len;
    if (!d->deferred) { sq_set_error("sq_dedup_shard_settle: the estimator is not in deferred mode"); return SQ_ERR_VALUE; }
    if (store_in) {
        if (len != fp_len) { sq_set_error("sq_dedup_shard_settle: a store of %zu bytes, the estimator's has %llu", len, (unsigned long long)fp_len); return SQ_ERR_VALUE; }
        d->store_in.assign(store_in, store_in + fp_len);
    }
    for (auto &u : d->unresolved) {
        for (uint64_t i = 0; i < fp_len; i++)
            if (!u.known[i]) u.bytes[i] = d->store_in[i];
        const uint8_t *sp = u.bytes.data();
        const unsigned long long h = murmur3_x64_64([&](uint64_t i) { return sp[i]; }, fp_len, u.seed);
        SQ_HIP(hipMemcpy(d->d_stream + u.pos, &h, 8, hipMemcpyHostToDevice));
    }
    d->unresolved.clear();
    for (uint64_t i = 0; i < fp_len; i++)
        if (!d->store_known[i]) d->store[i] = d->store_in[i];
    d->store_known.assign(fp_len, 1); /* the store is final: a later import or resolve leaves it alone */
    d->pass_valid = false;
    if (!lower_bound) return SQ_OK;
    *lower_bound = 0;
    const uint64_t n = d->stream_n;
    if (n <= d->max_stored) return SQ_OK; /* not even all of them distinct could make a rebuild certain */
    if (n > 0x7fffffffull) { sq_set_error("sq_dedup_shard_settle: %llu resident hashes (the sort takes 2^31-1)", (unsigned long long)n); return SQ_ERR_VALUE; }
    unsigned long long *d_sorted = (unsigned long long *)sq_scratch(ctx, 29, n * 8);
    unsigned long long *d_hist = (unsigned long long *)sq_scratch(ctx, 31, 65 * 8);
    if (!d_sorted || !d_hist) { sq_set_error("out of device memory"); return SQ_ERR_MEMORY; }
    size_t temp_bytes = 0;
    SQ_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, temp_bytes, d->d_stream, d_sorted, (int)n, 0, 64, ctx->stream));
    void *d_temp = sq_scratch(ctx, 30, temp_bytes ? temp_bytes : 8);
    if (!d_temp) { sq_set_error("out of device memory"); return SQ_ERR_MEMORY; }
    SQ_HIP(hipcub::DeviceRadixSort::SortKeys(d_temp, temp_bytes, d->d_stream, d_sorted, (int)n, 0, 64, ctx->stream));
    SQ_HIP(hipMemsetAsync(d_hist, 0, 65 * 8, ctx->stream));
    hipLaunchKernelGGL(k_dedup_ctz_hist, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, d_sorted, n, d_hist);
    SQ_HIP(hipGetLastError());
    unsigned long long hist[65];
    SQ_HIP(hipMemcpyAsync(hist, d_hist, sizeof hist, hipMemcpyDeviceToHost, ctx->stream));
    SQ_HIP(hipStreamSynchronize(ctx->stream));
    *lower_bound = sq_dedup_lower_bound_of((const uint64_t *)hist, d->max_stored);
    return SQ_OK;
}

/* the count, and with `out` the hashes themselves (as many as fit in cap) */
SQ_EXPORT int64_t sq_dedup_shard_passing(sq_dedup *d, uint64_t bits, uint64_t *out, size_t cap)
{
    sq_ctx *ctx = d->ctx;
    if (!d->deferred) { sq_set_error("sq_dedup_shard_passing: the estimator is not in deferred mode"); return SQ_ERR_VALUE; }
    if (!d->unresolved.empty()) { sq_set_error("sq_dedup_shard_passing: sq_dedup_shard_settle first"); return SQ_ERR_VALUE; }
    if (bits > 63) { sq_set_error("sq_dedup_shard_passing: a mask of %llu bits", (unsigned long long)bits); return SQ_ERR_VALUE; }
    if (!d->pass_valid || d->pass_bits != bits) {
        d->pass.clear();
        const uint64_t chunk = 1ull << 23;
        for (uint64_t off = 0; off < d->stream_n; off += chunk) {
            const uint64_t n = std::min(chunk, d->stream_n - off);
            const unsigned long long *d_hashes = d->d_stream + off;
            unsigned long long *d_idx = nullptr;
            uint64_t n_keep = 0;
            int rc = ordered_select(ctx, n, DedupKeep{(1ULL << bits) - 1, d_hashes, nullptr}, 7, &d_idx, &n_keep);
            if (rc) return rc;
            if (!n_keep) continue;
            unsigned long long *d_kh = (unsigned long long *)sq_scratch(ctx, 10, n_keep * 8);
            unsigned char *d_ks = (unsigned char *)sq_scratch(ctx, 11, n_keep);
            if (!d_kh || !d_ks) { sq_set_error("out of device memory"); return SQ_ERR_MEMORY; }
            hipLaunchKernelGGL(k_dedup_gather, dim3(blocks_for(n_keep)), dim3(256), 0, ctx->stream, d_idx, n_keep, d_hashes,
                               (const unsigned char *)nullptr, d_kh, d_ks);
            const size_t at = d->pass.size();
            d->pass.resize(at + n_keep);
            SQ_HIP(hipMemcpyAsync(d->pass.data() + at, d_kh, n_keep * 8, hipMemcpyDeviceToHost, ctx->stream));
            SQ_HIP(hipStreamSynchronize(ctx->stream));
        }
        d->pass_bits = bits;
        d->pass_valid = true;
    }
    if (out) memcpy(out, d->pass.data(), std::min(cap, d->pass.size()) * 8);
    return (int64_t)d->pass.size();
}

/* the shard's hashes went to the head: forget them (the table arrives with sq_dedup_import_state) */
SQ_EXPORT int sq_dedup_shard_drop(sq_dedup *d)
{
    if (!d->unresolved.empty()) { sq_set_error("sq_dedup_shard_drop: sq_dedup_shard_settle first"); return SQ_ERR_VALUE; }
    d->stream_n = 0;
    d->pass.clear();
    d->pass.shrink_to_fit();
    d->pass_valid = false;
    return SQ_OK;
}

/* the head: n hashes of one later shard (HOST array, read order, filtered with a mask of filtered_bits bits) through
 * the insertion tail, then the store that shard leaves.  SQ_DEDUP_FEED_TOO_STRICT (1) and nothing done when the
 * estimator has fewer bits than the filter assumed. */
SQ_EXPORT int sq_dedup_feed_hashes(sq_dedup *d, const uint64_t *hashes, size_t n, uint64_t filtered_bits,
                                   const uint8_t *store_after, size_t store_len)
{
    const uint64_t fp_len = d->front_len + d->back_len;
    if (d->stream_n || !d->unresolved.empty()) { sq_set_error("sq_dedup_feed_hashes: resolve the pending hashes first"); return SQ_ERR_VALUE; }
    if (store_after && store_len != fp_len) { sq_set_error("sq_dedup_feed_hashes: a store of %zu bytes, the estimator's has %llu", store_len, (unsigned long long)fp_len); return SQ_ERR_VALUE; }
    if (d->modulo_bits < filtered_bits) return SQ_DEDUP_FEED_TOO_STRICT;
    if (d->ctx && n) {   /* the stream through the table in HBM */
        unsigned long long *d_in = (unsigned long long *)sq_scratch(d->ctx, 39, n * 8);
        if (!d_in) { sq_set_error("out of device memory"); return SQ_ERR_MEMORY; }
        SQ_HIP(hipMemcpyAsync(d_in, hashes, n * 8, hipMemcpyHostToDevice, d->ctx->stream));
        SQ_HIP(hipStreamSynchronize(d->ctx->stream));
        if (int rc = dedup_process(d, d_in, n)) return rc;
    } else {             /* an estimator without a device: the reference's loop */
        for (size_t e = 0; e < n; e++) {
            if (e + 12 < n) {
                const uint64_t slot = (hashes[e + 12] >> d->modulo_bits) & (d->table_size - 1);
                __builtin_prefetch(&d->count[slot]);
                __builtin_prefetch(&d->hash[slot]);
            }
            dedup_insert(d, hashes[e]);
        }
    }
    if (store_after) {
        d->store.assign(store_after, store_after + fp_len);
        d->store_in = d->store;
    }
    return SQ_OK;
}

/* ---- InsertSizeMetrics ------------------------------------------------------------------ */

struct sq_adapter_entry {
    uint64_t hash = 0, count = 0;
    uint8_t len = 0;
    uint8_t bytes[SQ_ADAPTER_STORE_SIZE] = {0};
};

struct sq_insertsize {
    sq_ctx *ctx;
    uint64_t max_adapters, table_size;
    uint64_t total_reads = 0;
    IszTable tab[2];
    bool closed = false;
    size_t cap = 0; /* device histogram length */
    unsigned long long *d_sizes = nullptr, *d_max = nullptr;
    uint64_t max_insert = 0;
    /* read-out: the reference's tables, rebuilt from the device tables */
    uint64_t entries[2] = {0, 0};
    std::vector<sq_adapter_entry> table[2];
    /* shard mode: ranks count pairs of the whole job, tables never close */
    bool shard = false;
    uint64_t first_pair = 0;
};

static void isz_free_table(IszTable &T)
{
    for (void *p : {(void *)T.hash, (void *)T.count, (void *)T.rank, (void *)T.ready, (void *)T.key,
                    (void *)T.n_distinct, (void *)T.overflow})
        if (p) (void)hipFree(p);
    T = IszTable{};
}

static int isz_alloc_table(sq_ctx *ctx, IszTable &T, uint32_t bits)
{
    const uint64_t size = 1ull << bits;
    T = IszTable{};
    T.mask = size - 1;
    SQ_HIP(hipMalloc((void **)&T.hash, size * 8));
    SQ_HIP(hipMalloc((void **)&T.count, size * 8));
    SQ_HIP(hipMalloc((void **)&T.rank, size * 8));
    SQ_HIP(hipMalloc((void **)&T.ready, size * 4));
    SQ_HIP(hipMalloc((void **)&T.key, size * 32));
    SQ_HIP(hipMalloc((void **)&T.n_distinct, 16));
    SQ_HIP(hipMalloc((void **)&T.overflow, 4));
    T.n_events = T.n_distinct + 1;
    SQ_HIP(hipMemsetAsync(T.hash, 0, size * 8, ctx->stream));
    SQ_HIP(hipMemsetAsync(T.count, 0, size * 8, ctx->stream));
    SQ_HIP(hipMemsetAsync(T.rank, 0xFF, size * 8, ctx->stream));
    SQ_HIP(hipMemsetAsync(T.ready, 0, size * 4, ctx->stream));
    SQ_HIP(hipMemsetAsync(T.n_distinct, 0, 16, ctx->stream));
    SQ_HIP(hipMemsetAsync(T.overflow, 0, 4, ctx->stream));
    /* the module's kernels run on ctx->stream (non-blocking): order the clears on it */
    SQ_HIP(hipStreamSynchronize(ctx->stream));
    return SQ_OK;
}

SQ_EXPORT sq_insertsize *sq_insertsize_new(sq_ctx *ctx, int64_t max_adapters)
{
    if (max_adapters < 1) { /* :5515 */
        sq_set_error("max_adapters must be at least 1, got %lld", (long long)max_adapters);
        return nullptr;
    }
    sq_insertsize *z = new sq_insertsize();
    z->ctx = ctx;
    z->max_adapters = max_adapters;
    z->table_size = 1ULL << (uint64_t)(log2(max_adapters * 1.5) + 1); /* :5525 */
    SQ_HIP_NULL(hipMalloc((void **)&z->d_max, 8));
    SQ_HIP_NULL(hipMemsetAsync(z->d_max, 0, 8, ctx->stream));
    for (int w = 0; w < 2; w++)
        if (isz_alloc_table(ctx, z->tab[w], ISZ_TABLE_BITS) != SQ_OK) return nullptr;
    return z;
}

SQ_EXPORT void sq_insertsize_free(sq_insertsize *z)
{
    if (!z) return;
    (void)hipStreamSynchronize(z->ctx->stream);
    if (z->d_sizes) (void)hipFree(z->d_sizes);
    if (z->d_max) (void)hipFree(z->d_max);
    for (int w = 0; w < 2; w++) isz_free_table(z->tab[w]);
    delete z;
}

namespace {

/* InsertSizeMetrics_add_adapter, _qcmodule.c:5570-5611, for a key that brings its
 * whole count along (keys are replayed in the order of their first occurrence) */
void isz_replay_adapter(sq_insertsize *z, const uint8_t *a, size_t len, uint64_t count, int which)
{
    const uint64_t h = murmur3_x64_64([&](uint64_t i) { return a[i]; }, len, 0);
    const bool full = z->entries[which] == z->max_adapters;
    const uint64_t mask = z->table_size - 1;
    uint64_t i = h & mask;
    for (;;) {
        sq_adapter_entry &e = z->table[which][i];
        if (e.hash == h) {
            if (len == e.len && memcmp(a, e.bytes, len) == 0) { e.count += count; return; }
        } else if (e.count == 0) {
            if (!full) {
                e.hash = h; e.len = (uint8_t)len; e.count = count;
                memcpy(e.bytes, a, len);
                z->entries[which]++;
            }
            return;
        }
        i = (i + 1) & mask;
    }
}

/* (key, count, rank) of every used slot of table w, unordered */
int isz_download_used(sq_insertsize *z, int w, std::vector<unsigned long long> &key,
                      std::vector<unsigned long long> &count, std::vector<unsigned long long> &rank)
{
    sq_ctx *ctx = z->ctx;
    const IszTable &T = z->tab[w];
    unsigned long long nd = 0;
    SQ_HIP(hipStreamSynchronize(ctx->stream));
    SQ_HIP(hipMemcpy(&nd, T.n_distinct, 8, hipMemcpyDeviceToHost));
    key.clear(); count.clear(); rank.clear();
    if (nd == 0) return SQ_OK;
    unsigned long long *d_slots = nullptr, *d_n = nullptr, *d_out = nullptr;
    SQ_HIP(hipMalloc((void **)&d_slots, nd * 8));
    SQ_HIP(hipMalloc((void **)&d_n, 8));
    SQ_HIP(hipMalloc((void **)&d_out, nd * 48));
    SQ_HIP(hipMemsetAsync(d_n, 0, 8, ctx->stream));
    hipLaunchKernelGGL(k_isz_collect, dim3(blocks_for(T.mask + 1)), dim3(256), 0, ctx->stream, T, d_slots, d_n);
    unsigned long long used = 0;
    SQ_HIP(hipMemcpyAsync(&used, d_n, 8, hipMemcpyDeviceToHost, ctx->stream));
    SQ_HIP(hipStreamSynchronize(ctx->stream));
    if (used) {
        hipLaunchKernelGGL(k_isz_gather, dim3(blocks_for(used)), dim3(256), 0, ctx->stream, T, d_slots,
                           (uint64_t)used, d_out, d_out + 4 * used, d_out + 5 * used);
        key.resize(4 * used); count.resize(used); rank.resize(used);
        SQ_HIP(hipMemcpyAsync(key.data(), d_out, used * 32, hipMemcpyDeviceToHost, ctx->stream));
        SQ_HIP(hipMemcpyAsync(count.data(), d_out + 4 * used, used * 8, hipMemcpyDeviceToHost, ctx->stream));
        SQ_HIP(hipMemcpyAsync(rank.data(), d_out + 5 * used, used * 8, hipMemcpyDeviceToHost, ctx->stream));
        SQ_HIP(hipStreamSynchronize(ctx->stream));
    }
    (void)hipFree(d_slots); (void)hipFree(d_n); (void)hipFree(d_out);
    return SQ_OK;
}

int isz_rebuild_tables(sq_insertsize *z)
{
    for (int w = 0; w < 2; w++) {
        std::vector<unsigned long long> key, count, rank;
        int rc = isz_download_used(z, w, key, count, rank);
        if (rc) return rc;
        std::vector<uint64_t> used(count.size());
        for (uint64_t i = 0; i < used.size(); i++) used[i] = i;
        std::sort(used.begin(), used.end(), [&](uint64_t x, uint64_t y) { return rank[x] < rank[y]; });
        z->table[w].assign(z->table_size, sq_adapter_entry());
        z->entries[w] = 0;
        for (uint64_t i : used) {
            const uint8_t *rec = (const uint8_t *)&key[i * 4];
            isz_replay_adapter(z, rec + 1, rec[0], count[i], w);
        }
    }
    return SQ_OK;
}

} // namespace

static int isz_add_batch_pair(sq_insertsize *z, sq_batch *b1, sq_batch *b2, const uint32_t *scanned, uint64_t scanned_pairs);

SQ_EXPORT int sq_insertsize_add_batch_pair(sq_insertsize *z, sq_batch *b1, sq_batch *b2)
{
    return isz_add_batch_pair(z, b1, b2, nullptr, 0);
}

/* the same behind an overlap scan that ran inside read 1's QCMetrics pass (k_span<PAIR = 2>, sq_pair.hip): scanned[r] =
   the insert size of pair r for the first scanned_pairs pairs (a multiple of 16); the rest is scanned here */
int sq_insertsize_add_batch_pair_scanned(sq_insertsize *z, sq_batch *b1, sq_batch *b2, const uint32_t *scanned, uint64_t scanned_pairs)
{
    return isz_add_batch_pair(z, b1, b2, scanned, scanned_pairs);
}

/* where the scan inside read 1's pass leaves its results: [n] */
uint32_t *sq_insertsize_scan_results(sq_insertsize *z, uint64_t n)
{
    return (uint32_t *)sq_scratch(z->ctx, 16, n * 4);
}

/* does the histogram hold the largest value calculate_insert_size can return for these batches?  (grown if not) */
int sq_insertsize_reserve_for(sq_insertsize *z, sq_batch *b1, sq_batch *b2)
{
    return sq_grow_device(z->ctx, &z->d_sizes, &z->cap, (size_t)(b1->max_length + b2->max_length + 17));
}

static int isz_add_batch_pair(sq_insertsize *z, sq_batch *b1, sq_batch *b2, const uint32_t *scanned, uint64_t scanned_pairs)
{
    if (b1->n != b2->n) { /* :5842-5848 */
        sq_set_error("record_array1 and record_array2 must be of the same size. Got %zu and %zu respectively.",
                     b1->n, b2->n);
        return SQ_ERR_VALUE;
    }
    sq_ctx *ctx = z->ctx;
    const uint64_t n = b1->n;
    if (n == 0) return SQ_OK;
    /* the largest value calculate_insert_size can return */
    const size_t need = (size_t)(b1->max_length + b2->max_length + 17);
    int rc = sq_grow_device(ctx, &z->d_sizes, &z->cap, need);
    if (rc) return rc;
    IszParams P{};
    P.buf1 = b1->d_buf; P.buf2 = b2->d_buf; P.metas1 = b1->d_metas; P.metas2 = b2->d_metas;
    P.len1 = b1->buf_len;
    P.len2 = b2->buf_len;
    P.n = n; P.insert_sizes = z->d_sizes; P.max_insert = z->d_max;
    P.tab[0] = z->tab[0]; P.tab[1] = z->tab[1];
    P.rank_base = z->first_pair + z->total_reads;
    P.closed = z->closed ? 1 : 0;
    P.lds_sizes = (uint32_t)std::min<size_t>(z->cap, 8192);
    /* pairs of one read length each: the scan streams read 1 through LDS (k_isz_span, sq_span.hip)
       and leaves the adapter remainders to k_isz_adapters; the last few pairs (and everything
       else) go through k_insert_size.  SQ_SPAN=0: k_insert_size for all. */
    uint64_t covered = 0;
    if (scanned && scanned_pairs) {
        IszParams A = P;
        A.n = covered = scanned_pairs;
        sq_route(ctx, "k_isz_adapters<hist>");
        hipLaunchKernelGGL(k_isz_adapters<true>, dim3(blocks_for(covered, 4 * ctx->num_cus)), dim3(256), 0, ctx->stream, A, scanned,
                           (uint32_t)b1->max_length, (uint32_t)b2->max_length);
        SQ_HIP(hipGetLastError());
    } else if (b1->slack && b2->slack && b1->min_length == b1->max_length && b2->min_length == b2->max_length &&
        sq_knobs().span) {
        uint32_t *d_results = (uint32_t *)sq_scratch(ctx, 16, n * 4);
        if (d_results) {
            IszSpanParams S{};
            S.buf1 = P.buf1; S.buf2 = P.buf2; S.metas1 = P.metas1; S.metas2 = P.metas2; S.n = n;
            S.L1 = (uint32_t)b1->max_length; S.L2 = (uint32_t)b2->max_length;
            S.insert_sizes = P.insert_sizes; S.lds_sizes = (uint32_t)std::min<size_t>(z->cap, 1024);
            S.max_insert = P.max_insert; S.results = d_results;
            rc = sq_isz_span_launch(ctx, S, &covered);
            if (rc) return rc;
            if (covered) {
                IszParams A = P;
                A.n = covered;
                /* a workgroup per CU: every workgroup brings its cache of remainders to the device tables when it
                   is done, all of them at about the same time and most of them the same few keys */
                hipLaunchKernelGGL(k_isz_adapters<false>, dim3(blocks_for(covered, 4 * ctx->num_cus)), dim3(256), 0, ctx->stream, A, (const uint32_t *)d_results, 0u, 0u);
                SQ_HIP(hipGetLastError());
            }
        }
    }
    if (covered < n) {
        IszParams R = P;
        R.metas1 += covered; R.metas2 += covered; R.n = n - covered; R.rank_base += covered;
        hipLaunchKernelGGL(k_insert_size, dim3(blocks_for(R.n, 2048)), dim3(256), R.lds_sizes * 4, ctx->stream, R);
        SQ_HIP(hipGetLastError());
    }
    z->total_reads += n;
    if (!z->closed) {
        /* once max_adapters distinct remainders exist in both tables, later batches can
           only add to keys that are already there (first come, :5583,5599) */
        SQ_HIP(hipMemcpyAsync(&ctx->pinned[32], z->tab[0].n_distinct, 8, hipMemcpyDeviceToHost, ctx->stream));
        SQ_HIP(hipMemcpyAsync(&ctx->pinned[33], z->tab[1].n_distinct, 8, hipMemcpyDeviceToHost, ctx->stream));
        SQ_HIP(hipMemcpyAsync(&ctx->pinned[34], z->tab[0].overflow, 4, hipMemcpyDeviceToHost, ctx->stream));
        SQ_HIP(hipMemcpyAsync(&ctx->pinned[35], z->tab[1].overflow, 4, hipMemcpyDeviceToHost, ctx->stream));
        SQ_HIP(hipStreamSynchronize(ctx->stream));
        if ((uint32_t)ctx->pinned[34] || (uint32_t)ctx->pinned[35]) {
            sq_set_error("InsertSizeMetrics: more than %llu distinct adapter remainders in flight",
                         (unsigned long long)((z->tab[0].mask + 1) / 4 * 3));
            return SQ_ERR_MEMORY;
        }
        if (!z->shard && ctx->pinned[32] >= z->max_adapters && ctx->pinned[33] >= z->max_adapters)
            z->closed = true;
    }
    return SQ_OK;
}

SQ_EXPORT int sq_insertsize_add_pair(sq_insertsize *z, const uint8_t *buf1, size_t len1, const sq_meta *metas1,
                                     const uint8_t *buf2, size_t len2, const sq_meta *metas2, size_t n)
{
    sq_batch *b1 = sq_batch_upload(z->ctx, buf1, len1, metas1, n);
    sq_batch *b2 = sq_batch_upload(z->ctx, buf2, len2, metas2, n);
    int rc = (b1 && b2) ? sq_insertsize_add_batch_pair(z, b1, b2) : SQ_ERR_MEMORY;
    sq_batch_free(b1);
    sq_batch_free(b2);
    return rc;
}

SQ_EXPORT int sq_insertsize_flush(sq_insertsize *z) { return sq_synchronize(z->ctx); }
SQ_EXPORT uint64_t sq_insertsize_total_reads(sq_insertsize *z) { return z->total_reads; }
static uint64_t isz_events(sq_insertsize *z, int w)
{
    unsigned long long v = 0;
    (void)hipStreamSynchronize(z->ctx->stream);
    (void)hipMemcpy(&v, z->tab[w].n_events, 8, hipMemcpyDeviceToHost);
    return v;
}
SQ_EXPORT uint64_t sq_insertsize_number_of_adapters_read1(sq_insertsize *z) { return isz_events(z, 0); }
SQ_EXPORT uint64_t sq_insertsize_number_of_adapters_read2(sq_insertsize *z) { return isz_events(z, 1); }

SQ_EXPORT int64_t sq_insertsize_insert_sizes(sq_insertsize *z, uint64_t *out, size_t cap)
{
    sq_ctx *ctx = z->ctx;
    SQ_HIP(hipStreamSynchronize(ctx->stream));
    unsigned long long mx = 0;
    SQ_HIP(hipMemcpy(&mx, z->d_max, 8, hipMemcpyDeviceToHost));
    z->max_insert = mx;
    const size_t count = (size_t)mx + 1; /* :5884 */
    if (out && cap >= count) {
        if (z->d_sizes) SQ_HIP(hipMemcpy(out, z->d_sizes, count * 8, hipMemcpyDeviceToHost));
        else out[0] = 0;
    }
    return (int64_t)count;
}

SQ_EXPORT int64_t sq_insertsize_adapters(sq_insertsize *z, int read2, uint8_t *bytes, uint8_t *lengths,
                                         uint64_t *counts, size_t cap)
{
    if (isz_rebuild_tables(z) != SQ_OK) return SQ_ERR_HIP;
    size_t n = 0;
    for (const sq_adapter_entry &e : z->table[read2 ? 1 : 0]) { /* :5894-5910 slot order */
        if (!e.count) continue;
        if (bytes && n < cap) {
            memset(bytes + n * SQ_ADAPTER_STORE_SIZE, 0, SQ_ADAPTER_STORE_SIZE);
            memcpy(bytes + n * SQ_ADAPTER_STORE_SIZE, e.bytes, e.len);
            lengths[n] = e.len;
            counts[n] = e.count;
        }
        n++;
    }
    return (int64_t)n;
}

/* ---- InsertSizeMetrics across shards -------------------------------------- */
SQ_EXPORT int sq_insertsize_set_shard(sq_insertsize *z, uint64_t first_pair_index, uint32_t table_bits)
{
    if (z->total_reads) {
        sq_set_error("sq_insertsize_set_shard: call it before the first record arrays");
        return SQ_ERR_VALUE;
    }
    if (table_bits < ISZ_TABLE_BITS) table_bits = ISZ_TABLE_BITS;
    if (table_bits > 28) { sq_set_error("sq_insertsize_set_shard: table_bits above 28"); return SQ_ERR_VALUE; }
    SQ_HIP(hipStreamSynchronize(z->ctx->stream));
    for (int w = 0; w < 2; w++) {
        if (z->tab[w].mask + 1 == (1ull << table_bits)) continue;
        isz_free_table(z->tab[w]);
        int rc = isz_alloc_table(z->ctx, z->tab[w], table_bits);
        if (rc) return rc;
    }
    z->shard = true;
    z->closed = false;
    z->first_pair = first_pair_index;
    return SQ_OK;
}

/* the first min(distinct, max_adapters) remainders of this shard's table by rank:
 * keys [n][32] = {length, bytes[31]}, ranks [n]; host arrays */
SQ_EXPORT int64_t sq_insertsize_shard_candidates(sq_insertsize *z, int read2, uint8_t *keys, uint64_t *ranks,
                                                 size_t cap)
{
    std::vector<unsigned long long> key, count, rank;
    if (isz_download_used(z, read2 ? 1 : 0, key, count, rank) != SQ_OK) return SQ_ERR_HIP;
    std::vector<uint64_t> order(count.size());
    for (uint64_t i = 0; i < order.size(); i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint64_t x, uint64_t y) { return rank[x] < rank[y]; });
    const size_t m = std::min<size_t>(order.size(), z->max_adapters);
    if (!keys || cap < m) return (int64_t)m;
    for (size_t e = 0; e < m; e++) {
        memcpy(keys + e * 32, &key[order[e] * 4], 32);
        ranks[e] = rank[order[e]];
    }
    return (int64_t)m;
}

/* first max_adapters distinct keys of the shards' candidates by rank (host arrays) */
SQ_EXPORT int64_t sq_insertsize_shard_select(sq_insertsize *z, const uint8_t *keys, const uint64_t *ranks, size_t n,
                                             uint8_t *out_keys, uint64_t *out_ranks, size_t cap)
{
    std::vector<uint64_t> order(n);
    for (uint64_t i = 0; i < n; i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint64_t x, uint64_t y) {
        const int c = memcmp(keys + x * 32, keys + y * 32, 32);
        return c != 0 ? c < 0 : ranks[x] < ranks[y];
    });
    std::vector<uint64_t> firsts; /* per distinct key the entry with the smallest rank */
    for (uint64_t e = 0; e < n; e++)
        if (e == 0 || memcmp(keys + order[e] * 32, keys + order[e - 1] * 32, 32) != 0) firsts.push_back(order[e]);
    std::sort(firsts.begin(), firsts.end(), [&](uint64_t x, uint64_t y) { return ranks[x] < ranks[y]; });
    const size_t m = std::min<size_t>(firsts.size(), z->max_adapters);
    if (!out_keys || cap < m) return (int64_t)m;
    for (size_t e = 0; e < m; e++) {
        memcpy(out_keys + e * 32, keys + firsts[e] * 32, 32);
        out_ranks[e] = ranks[firsts[e]];
    }
    return (int64_t)m;
}

/* this shard's count of every selected key (host arrays) */
SQ_EXPORT int sq_insertsize_shard_lookup(sq_insertsize *z, int read2, const uint8_t *keys, size_t n, uint64_t *counts)
{
    if (n == 0) return SQ_OK;
    sq_ctx *ctx = z->ctx;
    unsigned long long *d_keys = nullptr, *d_out = nullptr;
    SQ_HIP(hipMalloc((void **)&d_keys, n * 32));
    SQ_HIP(hipMalloc((void **)&d_out, n * 8));
    SQ_HIP(hipMemcpyAsync(d_keys, keys, n * 32, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_isz_lookup, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, z->tab[read2 ? 1 : 0], d_keys,
                       (uint64_t)n, d_out);
    SQ_HIP(hipMemcpyAsync(counts, d_out, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    SQ_HIP(hipStreamSynchronize(ctx->stream));
    (void)hipFree(d_keys); (void)hipFree(d_out);
    return SQ_OK;
}

/* replaces table `read2` by the job's: selected keys, their ranks and summed counts;
 * n_events = the job's number_of_adapters_read{1,2} */
SQ_EXPORT int sq_insertsize_shard_install(sq_insertsize *z, int read2, const uint8_t *keys, const uint64_t *ranks,
                                          const uint64_t *counts, size_t n, uint64_t n_events)
{
    const int w = read2 ? 1 : 0;
    uint32_t bits = ISZ_TABLE_BITS;
    while ((1ull << bits) < 2 * n) bits++;
    SQ_HIP(hipStreamSynchronize(z->ctx->stream));
    isz_free_table(z->tab[w]);
    int rc = isz_alloc_table(z->ctx, z->tab[w], bits);
    if (rc) return rc;
    IszTable &T = z->tab[w];
    const uint64_t size = T.mask + 1;
    std::vector<unsigned long long> hash(size, 0), count(size, 0), rank(size, ~0ULL), key(size * 4, 0);
    std::vector<unsigned int> ready(size, 0);
    for (size_t e = 0; e < n; e++) {
        const uint8_t *rec = keys + e * 32;
        unsigned long long h = murmur3_x64_64([&](uint64_t i) { return rec[1 + i]; }, rec[0], 0);
        if (h == 0) h = 1;
        uint64_t i = h & T.mask;
        while (hash[i]) i = (i + 1) & T.mask;
        hash[i] = h; count[i] = counts[e]; rank[i] = ranks[e]; ready[i] = 1;
        memcpy(&key[i * 4], rec, 32);
    }
    SQ_HIP(hipMemcpy(T.hash, hash.data(), size * 8, hipMemcpyHostToDevice));
    SQ_HIP(hipMemcpy(T.count, count.data(), size * 8, hipMemcpyHostToDevice));
    SQ_HIP(hipMemcpy(T.rank, rank.data(), size * 8, hipMemcpyHostToDevice));
    SQ_HIP(hipMemcpy(T.ready, ready.data(), size * 4, hipMemcpyHostToDevice));
    SQ_HIP(hipMemcpy(T.key, key.data(), size * 32, hipMemcpyHostToDevice));
    const unsigned long long scalars[2] = {n, n_events};
    SQ_HIP(hipMemcpy(T.n_distinct, scalars, 16, hipMemcpyHostToDevice));
    return SQ_OK;
}

/* the job's total_reads and insert size histogram; ends shard mode */
SQ_EXPORT int sq_insertsize_shard_set_totals(sq_insertsize *z, uint64_t total_reads, const uint64_t *insert_sizes,
                                             size_t len)
{
    sq_ctx *ctx = z->ctx;
    SQ_HIP(hipStreamSynchronize(ctx->stream));
    int rc = sq_grow_device(ctx, &z->d_sizes, &z->cap, len ? len : 1);
    if (rc) return rc;
    SQ_HIP(hipMemsetAsync(z->d_sizes, 0, z->cap * 8, ctx->stream));
    if (len) SQ_HIP(hipMemcpyAsync(z->d_sizes, insert_sizes, len * 8, hipMemcpyHostToDevice, ctx->stream));
    const unsigned long long mx = len ? len - 1 : 0;
    SQ_HIP(hipMemcpyAsync(z->d_max, &mx, 8, hipMemcpyHostToDevice, ctx->stream));
    SQ_HIP(hipStreamSynchronize(ctx->stream));
    z->total_reads = total_reads;
    z->shard = false;
    z->first_pair = 0;
    unsigned long long nd[2] = {0, 0};
    for (int w = 0; w < 2; w++) SQ_HIP(hipMemcpy(&nd[w], z->tab[w].n_distinct, 8, hipMemcpyDeviceToHost));
    z->closed = nd[0] >= z->max_adapters && nd[1] >= z->max_adapters;
    return SQ_OK;
}
