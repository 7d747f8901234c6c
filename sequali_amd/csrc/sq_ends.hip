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

#include <algorithm>
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

/* ================================ OverrepresentedSequences ======================== */

/* sequence_to_canonical_kmer, _qcmodule.c:3657-3694.  >= 0: k-mer; -1: a byte
 * outside ACGTN; -2: N present */
__device__ long long canonical_kmer(const uint8_t *s, uint32_t k)
{
    uint64_t kmer = 0;
    bool has_n = false, has_other = false;
    for (uint32_t i = 0; i < k; i++) {
        const unsigned c = s[i], cls = sq_base_class(c);
        if (cls == 4) {
            if ((c | 0x20u) == 'n') has_n = true; else has_other = true;
        }
        kmer = (kmer << 2) | (cls & 3);
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

enum { OVR_NORMAL = 0, OVR_FULL = 1, OVR_CROSSING = 2 };
constexpr unsigned long long RANK_NONE = ~0ULL;

struct OvrParams {
    const uint8_t *buf;
    const sq_meta *metas;
    uint64_t n;               /* records in the batch */
    uint64_t first_sample;    /* record index of the first sampled record */
    uint64_t n_samples;       /* sampled records in this launch */
    uint64_t sample_base;     /* samples of this batch in front of this launch */
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
};

/* Sequence_duplication_insert_hash, _qcmodule.c:3542-3568, concurrent form */
__device__ void ovr_insert(const OvrParams &P, unsigned long long h, unsigned long long rank)
{
    uint64_t i = h & P.table_mask;
    for (;;) {
        unsigned long long cur = __hip_atomic_load(&P.hashes[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == 0) {
            if (P.mode == OVR_FULL) return; /* table is closed: new keys are dropped (:3553) */
            cur = atomicCAS(&P.hashes[i], 0ULL, h);
            if (cur == 0) {
                atomicAdd(P.n_unique, 1ULL);
                cur = h;
            }
        }
        if (cur == h) {
            if (P.mode == OVR_FULL) {
                /* entries removed by the cap keep their key with a zero count */
                if (__hip_atomic_load(&P.counts[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
                    atomicAdd(&P.counts[i], 1u);
            } else {
                atomicAdd(&P.counts[i], 1u);
                /* keys from earlier launches carry rank 0 and stay there */
                if (P.mode == OVR_CROSSING) atomicMin(&P.ranks[i], rank + 1);
            }
            return;
        }
        i = (i + 1) & P.table_mask;
    }
}

/* OverrepresentedSequences_add_meta, _qcmodule.c:3829-3942: one lane per sampled record */
__global__ void k_overrep(OvrParams P)
{
    unsigned long long local_frags = 0;
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
            const long long km = canonical_kmer(seq + at, (uint32_t)k);
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
        for (uint64_t i = 0; i < size; i++) /* flushed in slot order (:3925-3930) */
            if (stage[i]) ovr_insert(P, stage[i], ((P.sample_base + s) << 24) | i);
        local_frags += valid;
        if (warn) {
            atomicAdd(P.warn_count, 1ULL);
            atomicMax(P.warn_last, (long long)(P.record_base + r));
        }
    }
    if (local_frags) atomicAdd(P.total_fragments, local_frags);
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
        return special[idx] || (hashes[idx] & ignore_mask) == 0;
    }
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
constexpr uint64_t ISZ_TABLE_SIZE = 1ull << ISZ_TABLE_BITS;

struct IszTable {
    unsigned long long *hash;   /* 0 = free */
    unsigned long long *count;
    unsigned long long *rank;
    unsigned int *ready;        /* key bytes published */
    unsigned long long *key;    /* [size][4] */
    unsigned long long *n_distinct, *n_events;
    int *overflow;
};

struct IszParams {
    const uint8_t *buf1, *buf2;
    const sq_meta *metas1, *metas2;
    uint64_t n;
    unsigned long long *insert_sizes; /* [cap] */
    uint32_t lds_sizes;               /* histogram entries privatised in LDS (0: straight to HBM) */
    unsigned long long *max_insert;
    IszTable tab[2];
    uint64_t rank_base;               /* pairs seen before this batch */
    int closed;                       /* the first-come cap was reached in an earlier batch */
};

__device__ void isz_count_adapter(const IszTable &T, const uint8_t *a, uint32_t len,
                                  unsigned long long rank, int closed)
{
    unsigned long long key[4] = {0, 0, 0, 0};
    key[0] = len;
    for (uint32_t i = 0; i < len; i++) key[(i + 1) >> 3] |= (unsigned long long)a[i] << (8 * ((i + 1) & 7));
    unsigned long long h = murmur3_x64_64([&](uint64_t i) { return a[i]; }, len, 0);
    if (h == 0) h = 1; /* 0 marks a free slot */
    atomicAdd(T.n_events, 1ULL);
    uint64_t idx = h & (ISZ_TABLE_SIZE - 1);
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
                if (atomicAdd(T.n_distinct, 1ULL) > (ISZ_TABLE_SIZE / 4) * 3) *T.overflow = 1;
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
                atomicAdd(&T.count[idx], 1ULL);
                atomicMin(&T.rank[idx], rank);
                return;
            }
        }
        idx = (idx + 1) & (ISZ_TABLE_SIZE - 1);
    }
    *T.overflow = 1;
}

/* calculate_insert_size, _qcmodule.c:5667-5707, and the bookkeeping of
 * InsertSizeMetrics_add_sequence_pair_ptr :5709-5744 */
__global__ void k_insert_size(IszParams P)
{
    /* most pairs report the same few sizes (0 = no overlap found): count per workgroup first */
    extern __shared__ unsigned int l_sizes[];
    for (uint32_t i = threadIdx.x; i < P.lds_sizes; i += blockDim.x) l_sizes[i] = 0;
    __syncthreads();
    unsigned long long local_max = 0;
    for (uint64_t r = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; r < P.n;
         r += (uint64_t)gridDim.x * blockDim.x) {
        const sq_meta m1 = P.metas1[r], m2 = P.metas2[r];
        const uint8_t *s1 = P.buf1 + m1.record_start + m1.sequence_offset;
        const uint8_t *s2 = P.buf2 + m2.record_start + m2.sequence_offset;
        const uint32_t L1 = m1.sequence_length, L2 = m2.sequence_length;
        uint32_t result = 0;
        if (L1 >= 16 && L2 >= 16) {
            uint64_t h_lo = 0, h_hi = 0, t_lo = 0, t_hi = 0;
            for (int i = 0; i < 16; i++) {
                /* needle byte (15 - i) = complement of R2 byte i */
                const uint64_t hb = complement_or_zero(s2[i]);
                const uint64_t tb = complement_or_zero(s2[L2 - 16 + i]);
                const int pos = 15 - i;
                if (pos < 8) { h_lo |= hb << (8 * pos); t_lo |= tb << (8 * pos); }
                else { h_hi |= hb << (8 * (pos - 8)); t_hi |= tb << (8 * (pos - 8)); }
            }
            uint64_t lo = sq_load_u64_unaligned(s1), hi = sq_load_u64_unaligned(s1 + 8);
            const uint64_t UP = 0xDFDFDFDFDFDFDFDFULL;
            for (uint32_t i = 0; i + 16 <= L1; i++) {
                const uint64_t ulo = lo & UP, uhi = hi & UP;
                if (ulo == h_lo || uhi == h_hi) { /* :5695 then exact Hamming on raw bytes */
                    const uint64_t x = lo ^ h_lo, y = hi ^ h_hi;
                    int d = 0;
                    for (int b = 0; b < 8; b++) d += ((x >> (8 * b)) & 0xFF) != 0;
                    for (int b = 0; b < 8; b++) d += ((y >> (8 * b)) & 0xFF) != 0;
                    if (d <= 1) { result = i + 16; break; }
                }
                if (ulo == t_lo || uhi == t_hi) {
                    const uint64_t x = lo ^ t_lo, y = hi ^ t_hi;
                    int d = 0;
                    for (int b = 0; b < 8; b++) d += ((x >> (8 * b)) & 0xFF) != 0;
                    for (int b = 0; b < 8; b++) d += ((y >> (8 * b)) & 0xFF) != 0;
                    if (d <= 1) { result = i + L2; break; }
                }
                if (i + 17 <= L1) { /* slide the 16-byte window by one base */
                    lo = (lo >> 8) | (hi << 56);
                    hi = (hi >> 8) | ((uint64_t)s1[i + 16] << 56);
                }
            }
        }
        if (result < P.lds_sizes) atomicAdd(&l_sizes[result], 1u);
        else atomicAdd(&P.insert_sizes[result], 1ULL);
        if (result) {
            if (result > local_max) local_max = result;
            const unsigned long long rank = 2 * (P.rank_base + r);
            if (L1 > result) /* :5729-5735 */
                isz_count_adapter(P.tab[0], s1 + result, min(L1 - result, (uint32_t)SQ_ADAPTER_STORE_SIZE), rank, P.closed);
            if (L2 > result) /* :5736-5742 */
                isz_count_adapter(P.tab[1], s2 + result, min(L2 - result, (uint32_t)SQ_ADAPTER_STORE_SIZE), rank + 1, P.closed);
        }
    }
    if (local_max) atomicMax(P.max_insert, local_max);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < P.lds_sizes; i += blockDim.x)
        if (l_sizes[i]) atomicAdd(&P.insert_sizes[i], (unsigned long long)l_sizes[i]);
}

/* survivors only travel to the host: hash and the "needs the host" flag of every kept index */
__global__ void k_dedup_gather(const unsigned long long *idx, uint64_t n_keep,
                               const unsigned long long *hashes, const unsigned char *special,
                               unsigned long long *out_hash, unsigned char *out_special)
{
    for (uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; e < n_keep;
         e += (uint64_t)gridDim.x * blockDim.x) {
        out_hash[e] = hashes[idx[e]];
        out_special[e] = special[idx[e]];
    }
}

__global__ void k_iota(unsigned long long *p, uint64_t n)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x)
        p[i] = i;
}

int blocks_for(uint64_t n, int cap = 16384)
{
    uint64_t b = (n + 255) / 256;
    if (b > (uint64_t)cap) b = cap;
    return (int)(b ? b : 1);
}

/* indices [0,n) for which pred holds, in order, on the device; count on the host */
template <typename Pred>
int ordered_select(sq_ctx *ctx, uint64_t n, Pred pred, unsigned long long **d_out, uint64_t *count)
{
    unsigned long long *d_in = nullptr, *d_sel = nullptr, *d_num = nullptr;
    void *d_temp = nullptr;
    size_t temp_bytes = 0;
    *d_out = nullptr;
    *count = 0;
    if (n == 0) return SQ_OK;
    SQ_HIP(hipMalloc((void **)&d_in, n * 8));
    SQ_HIP(hipMalloc((void **)&d_sel, n * 8));
    SQ_HIP(hipMalloc((void **)&d_num, 8));
    hipLaunchKernelGGL(k_iota, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, d_in, n);
    SQ_HIP(hipcub::DeviceSelect::If(nullptr, temp_bytes, d_in, d_sel, d_num, (int)n, pred, ctx->stream));
    SQ_HIP(hipMalloc(&d_temp, temp_bytes ? temp_bytes : 8));
    SQ_HIP(hipcub::DeviceSelect::If(d_temp, temp_bytes, d_in, d_sel, d_num, (int)n, pred, ctx->stream));
    SQ_HIP(hipMemcpyAsync(&ctx->pinned[16], d_num, 8, hipMemcpyDeviceToHost, ctx->stream));
    SQ_HIP(hipStreamSynchronize(ctx->stream));
    *count = ctx->pinned[16];
    (void)hipFree(d_temp);
    (void)hipFree(d_in);
    (void)hipFree(d_num);
    *d_out = d_sel;
    return SQ_OK;
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
    unsigned long long *d_hashes = nullptr, *d_ranks = nullptr;
    unsigned int *d_counts = nullptr;
    unsigned long long *d_scalars = nullptr; /* [0] n_unique [1] total_fragments [2] warn_count [3] warn_last */
    uint64_t n_unique_host = 0;              /* value after the last synchronised batch */
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
    for (void *p : {(void *)o->d_hashes, (void *)o->d_counts, (void *)o->d_ranks, (void *)o->d_scalars})
        if (p) (void)hipFree(p);
    delete o;
}

SQ_EXPORT int sq_overrep_add_batch(sq_overrep *o, sq_batch *b)
{
    sq_ctx *ctx = o->ctx;
    const uint64_t n = b->n;
    /* records with (number_of_sequences + r) % sample_every == 0 are sampled (:3833) */
    const uint64_t phase = o->number_of_sequences % o->sample_every;
    const uint64_t first = phase == 0 ? 0 : o->sample_every - phase;
    const uint64_t n_samples = first < n ? (n - first + o->sample_every - 1) / o->sample_every : 0;
    const uint64_t record_base = o->number_of_sequences;
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
    P.buf = b->d_buf; P.metas = b->d_metas; P.n = n;
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
        if (o->full) {
            mode = OVR_FULL;
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
        P.n_samples = chunk;
        hipLaunchKernelGGL(k_overrep, dim3(blocks_for(chunk)), dim3(256), 0, ctx->stream, P);
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
            unsigned long long *d_rank = nullptr, *d_slot = nullptr, *d_n = nullptr;
            SQ_HIP(hipMalloc((void **)&d_rank, n_new * 8));
            SQ_HIP(hipMalloc((void **)&d_slot, n_new * 8));
            SQ_HIP(hipMalloc((void **)&d_n, 8));
            SQ_HIP(hipMemsetAsync(d_n, 0, 8, ctx->stream));
            hipLaunchKernelGGL(k_ovr_collect_new, dim3(blocks_for(o->table_size)), dim3(256), 0,
                               ctx->stream, o->d_ranks, o->table_size, d_rank, d_slot, d_n);
            std::vector<unsigned long long> ranks(n_new), slots(n_new);
            SQ_HIP(hipMemcpyAsync(ranks.data(), d_rank, n_new * 8, hipMemcpyDeviceToHost, ctx->stream));
            SQ_HIP(hipMemcpyAsync(slots.data(), d_slot, n_new * 8, hipMemcpyDeviceToHost, ctx->stream));
            SQ_HIP(hipStreamSynchronize(ctx->stream));
            std::vector<uint64_t> order(n_new);
            for (uint64_t i = 0; i < n_new; i++) order[i] = i;
            std::sort(order.begin(), order.end(), [&](uint64_t x, uint64_t y) { return ranks[x] < ranks[y]; });
            std::vector<unsigned long long> dead;
            for (uint64_t i = keep; i < n_new; i++) dead.push_back(slots[order[i]]);
            SQ_HIP(hipMemcpyAsync(d_slot, dead.data(), dead.size() * 8, hipMemcpyHostToDevice, ctx->stream));
            hipLaunchKernelGGL(k_ovr_kill, dim3(blocks_for(dead.size())), dim3(256), 0, ctx->stream,
                               o->d_counts, d_slot, (uint64_t)dead.size());
            unsigned long long capped = o->max_unique;
            SQ_HIP(hipMemcpyAsync(o->d_scalars, &capped, 8, hipMemcpyHostToDevice, ctx->stream));
            SQ_HIP(hipStreamSynchronize(ctx->stream));
            (void)hipFree(d_rank); (void)hipFree(d_slot); (void)hipFree(d_n);
            o->n_unique_host = o->max_unique;
        } else {
            o->n_unique_host = n_after;
        }
        if (o->n_unique_host >= o->max_unique) o->full = true;
    }
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
    std::vector<uint64_t> hash;
    std::vector<uint32_t> count;
    std::vector<uint8_t> store; /* the fingerprint buffer the reference reuses */
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
    return d;
}

SQ_EXPORT void sq_dedup_free(sq_dedup *d) { delete d; }

namespace {

/* DedupEstimator_increment_modulo, _qcmodule.c:4382-4423 */
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
 * (the pre-rebuild bit count indexes the triggering hash) */
inline void dedup_insert(sq_dedup *d, uint64_t h)
{
    const uint64_t bits = d->modulo_bits;
    if (h & ((1ULL << bits) - 1)) return;
    if (d->stored >= d->max_stored) dedup_rebuild(d);
    const uint64_t mask = d->table_size - 1;
    uint64_t i = (h >> bits) & mask;
    for (;;) {
        if (d->count[i] == 0) { d->hash[i] = h; d->count[i] = 1; d->stored++; return; }
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
int pair_store_bytes(sq_dedup *d, sq_batch *b1, sq_batch *b2, uint64_t r, std::vector<uint8_t> &bytes)
{
    std::vector<uint8_t> s1, s2;
    int rc = fetch_sequence(b1, r, s1);
    if (rc) return rc;
    rc = fetch_sequence(b2, r, s2);
    if (rc) return rc;
    const uint64_t L1 = s1.size(), L2 = s2.size();
    const uint64_t fl = std::min<uint64_t>(d->front_len, L1), fo = std::min<uint64_t>(d->front_off, L1 - fl);
    const uint64_t bl = std::min<uint64_t>(d->back_len, L2), bo = std::min<uint64_t>(d->back_off, L2 - bl);
    bytes.assign(s1.begin() + fo, s1.begin() + fo + fl);
    bytes.insert(bytes.end(), s2.begin() + bo, s2.begin() + bo + bl);
    return SQ_OK;
}

/* contents of the reference's fingerprint store right after pair r of this batch */
int store_after_pair(sq_dedup *d, sq_batch *b1, sq_batch *b2, uint64_t r, std::vector<uint8_t> &store)
{
    const uint64_t fp_len = d->front_len + d->back_len;
    store = d->store; /* state carried in from earlier batches */
    std::vector<bool> known(fp_len, false);
    uint64_t unknown = fp_len;
    for (uint64_t j = r + 1; j-- > 0 && unknown;) {
        std::vector<uint8_t> w;
        int rc = pair_store_bytes(d, b1, b2, j, w);
        if (rc) return rc;
        for (uint64_t i = 0; i < w.size(); i++)
            if (!known[i]) { store[i] = w[i]; known[i] = true; unknown--; }
    }
    return SQ_OK;
}

int dedup_run(sq_dedup *d, sq_batch *b1, sq_batch *b2)
{
    sq_ctx *ctx = d->ctx;
    const uint64_t n = b1->n;
    if (n == 0) return SQ_OK;
    unsigned long long *d_hashes = nullptr;
    unsigned char *d_special = nullptr;
    SQ_HIP(hipMalloc((void **)&d_hashes, n * 8));
    SQ_HIP(hipMalloc((void **)&d_special, n));
    DedupParams P{};
    P.buf1 = b1->d_buf; P.metas1 = b1->d_metas;
    P.buf2 = b2 ? b2->d_buf : nullptr; P.metas2 = b2 ? b2->d_metas : nullptr;
    P.n = n;
    P.front_len = d->front_len; P.back_len = d->back_len;
    P.front_off = d->front_off; P.back_off = d->back_off;
    P.hashes = d_hashes; P.special = d_special;
    hipLaunchKernelGGL(k_dedup_hash, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, P);
    SQ_HIP(hipGetLastError());
    /* only hashes that pass the mask in force at the start of the batch can matter;
       the mask only ever gets stricter (H3) */
    DedupKeep keep{(1ULL << d->modulo_bits) - 1, d_hashes, d_special};
    unsigned long long *d_idx = nullptr;
    uint64_t n_keep = 0;
    int rc = ordered_select(ctx, n, keep, &d_idx, &n_keep);
    if (rc) return rc;
    std::vector<unsigned long long> idx(n_keep), hashes(n_keep);
    std::vector<unsigned char> special(n_keep);
    if (n_keep) {
        unsigned long long *d_kh = nullptr;
        unsigned char *d_ks = nullptr;
        SQ_HIP(hipMalloc((void **)&d_kh, n_keep * 8));
        SQ_HIP(hipMalloc((void **)&d_ks, n_keep));
        hipLaunchKernelGGL(k_dedup_gather, dim3(blocks_for(n_keep)), dim3(256), 0, ctx->stream, d_idx,
                           n_keep, d_hashes, d_special, d_kh, d_ks);
        SQ_HIP(hipMemcpyAsync(idx.data(), d_idx, n_keep * 8, hipMemcpyDeviceToHost, ctx->stream));
        SQ_HIP(hipMemcpyAsync(hashes.data(), d_kh, n_keep * 8, hipMemcpyDeviceToHost, ctx->stream));
        SQ_HIP(hipMemcpyAsync(special.data(), d_ks, n_keep, hipMemcpyDeviceToHost, ctx->stream));
        SQ_HIP(hipStreamSynchronize(ctx->stream));
        (void)hipFree(d_kh);
        (void)hipFree(d_ks);
    }
    (void)hipFree(d_idx); (void)hipFree(d_hashes); (void)hipFree(d_special);
    const uint64_t fp_len = d->front_len + d->back_len;
    for (uint64_t e = 0; e < n_keep; e++) {
        if (e + 12 < n_keep) { /* the slot a hash lands in is known ahead: hide the table's cache misses */
            const uint64_t hn = hashes[e + 12];
            const uint64_t slot = (hn >> d->modulo_bits) & (d->table_size - 1);
            __builtin_prefetch(&d->count[slot]);
            __builtin_prefetch(&d->hash[slot]);
        }
        const uint64_t r = idx[e];
        uint64_t h = hashes[e];
        if (special[e]) {
            std::vector<uint8_t> store;
            rc = store_after_pair(d, b1, b2, r, store);
            if (rc) return rc;
            sq_meta m1, m2;
            if (!b1->h_metas.empty()) { m1 = b1->h_metas[r]; m2 = b2->h_metas[r]; }
            else {
                SQ_HIP(hipMemcpy(&m1, b1->d_metas + r, sizeof(sq_meta), hipMemcpyDeviceToHost));
                SQ_HIP(hipMemcpy(&m2, b2->d_metas + r, sizeof(sq_meta), hipMemcpyDeviceToHost));
            }
            const uint8_t *sp = store.data();
            h = murmur3_x64_64([&](uint64_t i) { return sp[i]; }, fp_len,
                               ((uint64_t)m1.sequence_length + m2.sequence_length) >> 6);
        }
        dedup_insert(d, h);
    }
    if (b2) { /* carry the store into the next batch (usually one step: the last
                 pair rewrote all of it) */
        std::vector<uint8_t> store;
        rc = store_after_pair(d, b1, b2, n - 1, store);
        if (rc) return rc;
        d->store = store;
    }
    return SQ_OK;
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

SQ_EXPORT int64_t sq_dedup_duplication_counts(sq_dedup *d, uint64_t *out, size_t cap)
{
    size_t n = 0;
    for (uint64_t i = 0; i < d->table_size; i++) { /* :4736-4744 slot order */
        if (!d->count[i]) continue;
        if (out && n < cap) out[n] = d->count[i];
        n++;
    }
    return (int64_t)n;
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
};

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
    SQ_HIP_NULL(hipMemset(z->d_max, 0, 8));
    for (int w = 0; w < 2; w++) {
        IszTable &T = z->tab[w];
        SQ_HIP_NULL(hipMalloc((void **)&T.hash, ISZ_TABLE_SIZE * 8));
        SQ_HIP_NULL(hipMalloc((void **)&T.count, ISZ_TABLE_SIZE * 8));
        SQ_HIP_NULL(hipMalloc((void **)&T.rank, ISZ_TABLE_SIZE * 8));
        SQ_HIP_NULL(hipMalloc((void **)&T.ready, ISZ_TABLE_SIZE * 4));
        SQ_HIP_NULL(hipMalloc((void **)&T.key, ISZ_TABLE_SIZE * 32));
        SQ_HIP_NULL(hipMalloc((void **)&T.n_distinct, 16));
        SQ_HIP_NULL(hipMalloc((void **)&T.overflow, 4));
        T.n_events = T.n_distinct + 1;
        SQ_HIP_NULL(hipMemset(T.hash, 0, ISZ_TABLE_SIZE * 8));
        SQ_HIP_NULL(hipMemset(T.count, 0, ISZ_TABLE_SIZE * 8));
        SQ_HIP_NULL(hipMemset(T.rank, 0xFF, ISZ_TABLE_SIZE * 8));
        SQ_HIP_NULL(hipMemset(T.ready, 0, ISZ_TABLE_SIZE * 4));
        SQ_HIP_NULL(hipMemset(T.n_distinct, 0, 16));
        SQ_HIP_NULL(hipMemset(T.overflow, 0, 4));
    }
    return z;
}

SQ_EXPORT void sq_insertsize_free(sq_insertsize *z)
{
    if (!z) return;
    (void)hipStreamSynchronize(z->ctx->stream);
    if (z->d_sizes) (void)hipFree(z->d_sizes);
    if (z->d_max) (void)hipFree(z->d_max);
    for (int w = 0; w < 2; w++) {
        IszTable &T = z->tab[w];
        for (void *p : {(void *)T.hash, (void *)T.count, (void *)T.rank, (void *)T.ready, (void *)T.key,
                        (void *)T.n_distinct, (void *)T.overflow})
            if (p) (void)hipFree(p);
    }
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

int isz_rebuild_tables(sq_insertsize *z)
{
    sq_ctx *ctx = z->ctx;
    SQ_HIP(hipStreamSynchronize(ctx->stream));
    for (int w = 0; w < 2; w++) {
        const IszTable &T = z->tab[w];
        std::vector<unsigned long long> hash(ISZ_TABLE_SIZE), count(ISZ_TABLE_SIZE), rank(ISZ_TABLE_SIZE);
        std::vector<unsigned long long> key(ISZ_TABLE_SIZE * 4);
        SQ_HIP(hipMemcpy(hash.data(), T.hash, ISZ_TABLE_SIZE * 8, hipMemcpyDeviceToHost));
        SQ_HIP(hipMemcpy(count.data(), T.count, ISZ_TABLE_SIZE * 8, hipMemcpyDeviceToHost));
        SQ_HIP(hipMemcpy(rank.data(), T.rank, ISZ_TABLE_SIZE * 8, hipMemcpyDeviceToHost));
        SQ_HIP(hipMemcpy(key.data(), T.key, ISZ_TABLE_SIZE * 32, hipMemcpyDeviceToHost));
        std::vector<uint64_t> used;
        for (uint64_t i = 0; i < ISZ_TABLE_SIZE; i++)
            if (hash[i] && count[i]) used.push_back(i);
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

SQ_EXPORT int sq_insertsize_add_batch_pair(sq_insertsize *z, sq_batch *b1, sq_batch *b2)
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
    P.n = n; P.insert_sizes = z->d_sizes; P.max_insert = z->d_max;
    P.tab[0] = z->tab[0]; P.tab[1] = z->tab[1];
    P.rank_base = z->total_reads;
    P.closed = z->closed ? 1 : 0;
    P.lds_sizes = (uint32_t)std::min<size_t>(z->cap, 8192);
    hipLaunchKernelGGL(k_insert_size, dim3(blocks_for(n, 2048)), dim3(256), P.lds_sizes * 4, ctx->stream, P);
    SQ_HIP(hipGetLastError());
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
                         (unsigned long long)(ISZ_TABLE_SIZE / 4 * 3));
            return SQ_ERR_MEMORY;
        }
        if (ctx->pinned[32] >= z->max_adapters && ctx->pinned[33] >= z->max_adapters) z->closed = true;
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
