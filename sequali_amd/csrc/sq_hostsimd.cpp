/* sq_hostsimd.cpp -- the two byte scans of the host-side FASTQ parser (newline positions, first
 * byte >= 0x80) with AVX2 when the CPU has it.  Plain C++ (no HIP): the device pass of a .hip file
 * cannot see x86 intrinsics.  Used by sq_split_range / sq_first_non_ascii (sq_api.hip). */
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <immintrin.h>

namespace {

/* *high: set to the offset (base + ..) of the first byte >= 0x80 among the bytes looked at, if
   there is one and *high was UINT32_MAX */
__attribute__((target("avx2"))) size_t scan_nl_avx2(const uint8_t *p, size_t n, uint32_t base, uint32_t *out, size_t cap, size_t *scanned,
                                                    uint32_t *high)
{
    const __m256i nl = _mm256_set1_epi8('\n');
    size_t k = 0, i = 0;
    for (; i + 32 <= n && k + 32 <= cap; i += 32) {
        const __m256i v = _mm256_loadu_si256((const __m256i *)(p + i));
        uint32_t m = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, nl));
        const uint32_t h = (uint32_t)_mm256_movemask_epi8(v);
        if (__builtin_expect(h != 0, 0) && *high == UINT32_MAX) *high = base + (uint32_t)i + (uint32_t)__builtin_ctz(h);
        while (m) {
            out[k++] = base + (uint32_t)i + (uint32_t)__builtin_ctz(m);
            m &= m - 1;
        }
    }
    for (; i < n && k < cap && (i + 32 > n); i++) {
        if (p[i] == '\n') out[k++] = base + (uint32_t)i;
        if ((p[i] & 0x80) && *high == UINT32_MAX) *high = base + (uint32_t)i;
    }
    *scanned = i;
    return k;
}

size_t scan_nl_plain(const uint8_t *p, size_t n, uint32_t base, uint32_t *out, size_t cap, size_t *scanned, uint32_t *high)
{
    size_t k = 0, i = 0;
    for (; i < n && k < cap; i++) {
        if (p[i] == '\n') out[k++] = base + (uint32_t)i;
        if ((p[i] & 0x80) && *high == UINT32_MAX) *high = base + (uint32_t)i;
    }
    *scanned = i;
    return k;
}

__attribute__((target("avx2"))) int64_t first_non_ascii_avx2(const uint8_t *p, size_t n)
{
    size_t i = 0;
    for (; i + 128 <= n; i += 128) {
        const __m256i a = _mm256_loadu_si256((const __m256i *)(p + i)), b = _mm256_loadu_si256((const __m256i *)(p + i + 32));
        const __m256i c = _mm256_loadu_si256((const __m256i *)(p + i + 64)), d = _mm256_loadu_si256((const __m256i *)(p + i + 96));
        if (_mm256_movemask_epi8(_mm256_or_si256(_mm256_or_si256(a, b), _mm256_or_si256(c, d)))) break;
    }
    for (; i < n; i++)
        if (p[i] & 0x80) return (int64_t)i;
    return -1;
}

/* scan_nl_avx2 over src while its bytes go to dst with streaming stores (nobody reads dst through the cache: the copy engines
   and the device do), and the byte behind every newline noted from src (0: it lies behind the n bytes) */
__attribute__((target("avx2"))) size_t copy_scan_nl_avx2(uint8_t *dst, const uint8_t *src, size_t n, uint32_t base, uint32_t *out, uint8_t *after,
                                                         size_t cap, size_t *copied, uint32_t *high)
{
    const __m256i nl = _mm256_set1_epi8('\n');
    size_t k = 0, i = 0;
    auto byte = [&](size_t j) {
        const uint8_t c = src[j];
        dst[j] = c;
        if (c == '\n') { out[k] = base + (uint32_t)j; after[k++] = j + 1 < n ? src[j + 1] : 0; }
        if ((c & 0x80) && *high == UINT32_MAX) *high = base + (uint32_t)j;
    };
    for (; i < n && k < cap && ((uintptr_t)(dst + i) & 31); i++) byte(i);
    for (; i + 64 <= n && k + 64 <= cap; i += 64) {
        const __m256i a = _mm256_loadu_si256((const __m256i *)(src + i)), b = _mm256_loadu_si256((const __m256i *)(src + i + 32));
        _mm256_stream_si256((__m256i *)(dst + i), a);
        _mm256_stream_si256((__m256i *)(dst + i + 32), b);
        uint64_t m = (uint64_t)(uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(a, nl)) | ((uint64_t)(uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(b, nl)) << 32);
        const uint64_t h = (uint64_t)(uint32_t)_mm256_movemask_epi8(a) | ((uint64_t)(uint32_t)_mm256_movemask_epi8(b) << 32);
        if (__builtin_expect(h != 0, 0) && *high == UINT32_MAX) *high = base + (uint32_t)i + (uint32_t)__builtin_ctzll(h);
        while (m) {
            const size_t j = i + (size_t)__builtin_ctzll(m);
            out[k] = base + (uint32_t)j;
            after[k++] = j + 1 < n ? src[j + 1] : 0;
            m &= m - 1;
        }
    }
    for (; i < n && k < cap && (i + 64 > n); i++) byte(i);
    _mm_sfence();
    *copied = i;
    return k;
}

const bool g_avx2 = __builtin_cpu_supports("avx2");

}  // namespace

/* positions (base + offset in p) of the next newlines of p[0, n): fills out[0, cap), sets *scanned
 * to the bytes looked at (every newline in front of it has been reported) and *high to the
 * position of the first byte >= 0x80 among them (unchanged if none); returns the count */
size_t sq_scan_newlines(const uint8_t *p, size_t n, uint32_t base, uint32_t *out, size_t cap, size_t *scanned, uint32_t *high)
{
    return g_avx2 ? scan_nl_avx2(p, n, base, out, cap, scanned, high) : scan_nl_plain(p, n, base, out, cap, scanned, high);
}

/* sq_scan_newlines of src[0, n) while those bytes are copied to dst; after[k]: the byte behind newline k, 0 where that lies
 * behind the n bytes.  *copied: the bytes done (all n unless out is full) */
size_t sq_copy_scan_newlines(uint8_t *dst, const uint8_t *src, size_t n, uint32_t base, uint32_t *out, uint8_t *after, size_t cap, size_t *copied,
                             uint32_t *high)
{
    if (g_avx2) return copy_scan_nl_avx2(dst, src, n, base, out, after, cap, copied, high);
    size_t k = 0, i = 0;
    for (; i < n && k < cap; i++) {
        const uint8_t c = src[i];
        dst[i] = c;
        if (c == '\n') { out[k] = base + (uint32_t)i; after[k++] = i + 1 < n ? src[i + 1] : 0; }
        if ((c & 0x80) && *high == UINT32_MAX) *high = base + (uint32_t)i;
    }
    *copied = i;
    return k;
}

int64_t sq_first_non_ascii_fast(const uint8_t *p, size_t n)
{
    if (g_avx2) return first_non_ascii_avx2(p, n);
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        memcpy(&w, p + i, 8);
        if (w & 0x8080808080808080ULL) break;
    }
    for (; i < n; i++)
        if (p[i] & 0x80) return (int64_t)i;
    return -1;
}
