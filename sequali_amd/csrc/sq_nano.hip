/*
 * sq_nano.hip -- NanoStats (_qcmodule.c:4804-5430): one lane per record
 * reads the nanopore metadata of a read, from its BAM tags when it has any, else from
 * its FASTQ header, and fills one 40-byte NanoInfo.  accumulated_error_rate comes from
 * the meta in HBM, where the QCMetrics pass wrote it on the same stream (:2126, :5314).
 *
 * What the reference makes depend on the order of the reads is settled after the
 * kernel: everything stops at the first record that raises (records in front of it stay
 * counted) or, permanently, at the first header that does not parse (:5302-5312); and
 * minimum_time restarts after a timestamp of 0 (:5319-5321).
 */
#include <algorithm>

#include "sq_common.h"

namespace {

/* status word of a record: code | chars << 8; a pi tag of the wrong length also sets
 * bit 7 of the code and stores its length in the upper 24 bits (then no chars) */
enum { NANO_OK = 0, NANO_TRUNCATED = 1, NANO_ARRAY_TYPE = 2, NANO_UNKNOWN_TYPE = 3, NANO_WRONG_TYPECODE = 4,
       NANO_CH_NOT_INT = 5, NANO_BAD_HEADER = 6, NANO_PI_WARNING = 0x80 };

struct Bytes { /* bounded view of the batch buffer: reads past it give 0 */
    const uint8_t *buf;
    uint64_t len;
    __device__ uint8_t at(uint64_t off) const { return off < len ? buf[off] : (uint8_t)0; }
};

/* unsigned_decimal_integer_from_string :159-180 */
__device__ long long nano_decimal(const Bytes &B, uint64_t off, uint64_t n)
{
    if (n < 1 || n > 18) return -1;
    unsigned long long r = 0;
    for (uint64_t i = 0; i < n; i++) {
        uint8_t c = (uint8_t)(B.at(off + i) - '0');
        if (c > 9) return -1;
        r = r * 10 + c;
    }
    return (long long)r;
}

/* posix_gm_time :247-262 */
__device__ long long nano_gm_time(long long year, long long month, long long mday, long long hour,
                                  long long minute, long long second)
{
    if (year < 1970 || month < 1 || month > 12) return -1;
    year -= 1900;
    const int cum[12] = {0, 31, 59, 90, 120, 151, 181, 212, 243, 273, 304, 334};
    const long long yday = cum[month - 1] + mday - 1;
    return second + minute * 60 + hour * 3600 + yday * 86400 + (year - 70) * 31536000 +
           ((year - 69) / 4) * 86400 - ((year - 1) / 100) * 86400 + ((year + 299) / 400) * 86400;
}

/* time_string_to_timestamp :271-322 */
__device__ long long nano_timestamp(const Bytes &B, uint64_t s)
{
    long long year = nano_decimal(B, s, 4), month = nano_decimal(B, s + 5, 2), day = nano_decimal(B, s + 8, 2);
    long long hour = nano_decimal(B, s + 11, 2), minute = nano_decimal(B, s + 14, 2);
    const long long second = nano_decimal(B, s + 17, 2);
    if ((year | month | day | hour | minute | second) < 0 || B.at(s + 4) != '-' || B.at(s + 7) != '-' ||
        B.at(s + 10) != 'T' || B.at(s + 13) != ':' || B.at(s + 16) != ':')
        return -1;
    uint64_t tz = s + 19;
    if (B.at(tz) == '.') {
        uint64_t digits = 0;
        while ((uint8_t)(B.at(s + 20 + digits) - '0') <= 9) digits++;
        tz += digits + 1;
    }
    const uint8_t sign = B.at(tz);
    if (sign == '+' || sign == '-') {
        const long long oh = nano_decimal(B, tz + 1, 2), om = nano_decimal(B, tz + 4, 2);
        if ((oh | om) < 0 || B.at(tz + 3) != ':') return -1;
        if (sign == '+') { hour += oh; minute += om; }
        else { hour -= oh; minute -= om; }
    } else if (sign != 'Z') {
        return -1;
    }
    return nano_gm_time(year, month, day, hour, minute, second);
}

/* NanoInfo_from_header :5005-5052 */
__device__ bool nano_from_header(const Bytes &B, uint64_t name, uint64_t n, sq_nanoinfo &info)
{
    const uint64_t end = name + n;
    uint64_t cursor = name;
    while (cursor < end && B.at(cursor) != ' ') cursor++;
    if (cursor >= end) return false;
    cursor++;
    int32_t channel = -1;
    long long start = -1;
    while (cursor < end) {
        uint64_t eq = cursor;
        while (eq < end && B.at(eq) != '=') eq++;
        if (eq >= end) return false;
        const uint64_t name_len = eq - cursor, value = eq + 1;
        uint64_t value_end = value;
        while (value_end < end && B.at(value_end) != ' ') value_end++;
        if (name_len == 2 && B.at(cursor) == 'c' && B.at(cursor + 1) == 'h') {
            channel = (int32_t)nano_decimal(B, value, value_end - value);
        } else if (name_len == 10) {
            const char want[11] = "start_time";
            bool same = true;
            for (int i = 0; i < 10; i++) same &= B.at(cursor + i) == (uint8_t)want[i];
            if (same) start = nano_timestamp(B, value);
        }
        cursor = value_end + 1;
    }
    if (channel == -1 || start == -1) return false;
    info.channel_id = channel;
    info.start_time = start;
    return true;
}

__device__ uint32_t load_le(const Bytes &B, uint64_t off, int bytes)
{
    uint32_t v = 0;
    for (int i = 0; i < bytes; i++) v |= (uint32_t)B.at(off + i) << (8 * i);
    return v;
}

/* tag_length :5077-5143; < 0: -(status word) */
__device__ long long nano_tag_length(const Bytes &B, uint64_t tag, uint64_t max)
{
    if (max < 4) return -(long long)NANO_TRUNCATED;
    uint8_t type = B.at(tag + 2);
    uint64_t value = tag + 3, value_len;
    bool is_array = false;
    uint64_t count = 1;
    if (type == 'B') {
        is_array = true;
        value = tag + 8;
        type = B.at(tag + 3);
        if (max < 8) return -(long long)NANO_TRUNCATED;
        count = load_le(B, tag + 4, 4);
    }
    switch (type) {
        case 'A': case 'c': case 'C': value_len = 1; break;
        case 's': case 'S': value_len = 2; break;
        case 'I': case 'i': case 'f': value_len = 4; break;
        case 'Z': case 'H': {
            if (is_array) return -(long long)(NANO_ARRAY_TYPE | ((uint32_t)type << 8));
            uint64_t z = value;
            const uint64_t stop = tag + max; /* memchr(value_start, 0, maximum_tag_length - 3) */
            while (z < stop && B.at(z) != 0) z++;
            if (z >= stop) return -(long long)NANO_TRUNCATED;
            value_len = z - value + 1;
            break;
        }
        default: return -(long long)(NANO_UNKNOWN_TYPE | ((uint32_t)type << 8));
    }
    const uint64_t len = (value - tag) + count * value_len;
    if (len > max) return -(long long)NANO_TRUNCATED;
    return (long long)len;
}

__device__ int hex_value(uint8_t c)
{
    if (c >= '0' && c <= '9') return c - '0';
    const uint8_t l = c | 0x20;
    return (l >= 'a' && l <= 'f') ? l - 'a' + 10 : -1;
}

/* uuid4_hash :5155-5182 */
__device__ unsigned long long nano_uuid4_hash(const Bytes &B, uint64_t u)
{
    if (B.at(u + 8) != '-' || B.at(u + 13) != '-' || B.at(u + 14) != '4' || B.at(u + 18) != '-' ||
        B.at(u + 23) != '-' || B.at(u + 36) != 0)
        return 0;
    unsigned long long first = 0, last = 0;
    for (int i = 0; i < 8; i++) {
        const int v = hex_value(B.at(u + i));
        if (v < 0) return 0;
        first = first * 16 + (unsigned)v;
    }
    for (int i = 28; i < 36; i++) {
        const int v = hex_value(B.at(u + i));
        if (v < 0) return 0;
        last = last * 16 + (unsigned)v;
    }
    return (first << 32) | (last & 0xFFFFFFFFULL);
}

/* TagInfo_from_tags :5205-5259; returns the status word */
__device__ uint32_t nano_from_tags(const Bytes &B, uint64_t tags, uint64_t n, sq_nanoinfo &info)
{
    info.channel_id = -1;
    info.duration = 0.0f;
    info.start_time = 0;
    info.parent_id_hash = 0;
    uint32_t warned = 0;
    while (n > 0) {
        const long long len = nano_tag_length(B, tags, n);
        if (len < 0) return (uint32_t)(-len);
        const uint8_t a = B.at(tags), b = B.at(tags + 1), type = B.at(tags + 2);
        if (a == 'c' && b == 'h') {
            long long v;
            switch (type) { /* get_tag_int_value :5054-5075 */
                case 'c': v = (int8_t)B.at(tags + 3); break;
                case 'C': v = B.at(tags + 3); break;
                case 's': v = (int16_t)load_le(B, tags + 3, 2); break;
                case 'S': v = (uint16_t)load_le(B, tags + 3, 2); break;
                case 'i': v = (int32_t)load_le(B, tags + 3, 4); break;
                case 'I': v = load_le(B, tags + 3, 4); break;
                default: return NANO_CH_NOT_INT;
            }
            info.channel_id = (int32_t)v;
        } else if (a == 's' && b == 't') {
            if (type != 'Z') return NANO_WRONG_TYPECODE | ('s' << 8) | ('t' << 16) | ((uint32_t)type << 24);
            info.start_time = nano_timestamp(B, tags + 3);
        } else if (a == 'd' && b == 'u') {
            if (type != 'f') return NANO_WRONG_TYPECODE | ('d' << 8) | ('u' << 16) | ((uint32_t)type << 24);
            info.duration = __uint_as_float(load_le(B, tags + 3, 4));
        } else if (a == 'p' && b == 'i') {
            if (type != 'Z') return NANO_WRONG_TYPECODE | ('p' << 8) | ('i' << 16) | ((uint32_t)type << 24);
            if (len - 4 != 36) warned = NANO_PI_WARNING | ((uint32_t)std::min<long long>(len - 4, 0xFFFFFF) << 8);
            else info.parent_id_hash = nano_uuid4_hash(B, tags + 3);
        }
        tags += (uint64_t)len;
        n -= (uint64_t)len;
    }
    return warned;
}

/* NanoStats_add_meta :5269-5324 for every record of a batch */
__global__ void k_nano_parse(const uint8_t *buf, uint64_t buf_len, const sq_meta *metas, uint64_t n,
                             sq_nanoinfo *infos, uint32_t *status, unsigned long long *first_stop)
{
    const Bytes B{buf, buf_len};
    for (uint64_t r = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; r < n;
         r += (uint64_t)gridDim.x * blockDim.x) {
        const sq_meta m = metas[r];
        sq_nanoinfo info;
        info.start_time = 0; info.duration = 0.0f; info.channel_id = 0; info.pad_ = 0; info.parent_id_hash = 0;
        info.length = m.sequence_length;
        uint32_t st;
        if (m.tags_length) st = nano_from_tags(B, m.record_start + m.tags_offset, m.tags_length, info);
        else st = nano_from_header(B, m.record_start, m.name_length, info) ? NANO_OK : NANO_BAD_HEADER;
        info.cumulative_error_rate = m.accumulated_error_rate;
        infos[r] = info;
        status[r] = st;
        if (st & 0x7F) atomicMin(first_stop, (unsigned long long)r);
    }
}

/* over the counted records: maximum, the last record with a timestamp of 0 (+1), pi warnings */
__global__ void k_nano_scan1(const sq_nanoinfo *infos, const uint32_t *status, uint64_t n, uint64_t n_status,
                             long long *max_time, unsigned long long *last_zero, unsigned long long *n_warned,
                             long long *all_min)
{
    long long mx = INT64_MIN, mn = INT64_MAX;
    unsigned long long lz = 0, w = 0;
    for (uint64_t r = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; r < n_status;
         r += (uint64_t)gridDim.x * blockDim.x) {
        if (status[r] & NANO_PI_WARNING) w++;
        if (r >= n) continue;
        const long long t = infos[r].start_time;
        mx = t > mx ? t : mx;
        mn = t < mn ? t : mn;
        if (t == 0) lz = r + 1;
    }
    /* per wave first: every thread would hit the same four addresses */
    for (int off = 32; off > 0; off >>= 1) {
        const long long omx = __shfl_xor(mx, off), omn = __shfl_xor(mn, off);
        const unsigned long long olz = __shfl_xor(lz, off);
        mx = omx > mx ? omx : mx;
        mn = omn < mn ? omn : mn;
        lz = olz > lz ? olz : lz;
        w += __shfl_xor(w, off);
    }
    if ((threadIdx.x & 63) != 0) return;
    if (mx != INT64_MIN) atomicMax(max_time, mx);
    if (mn != INT64_MAX) atomicMin(all_min, mn);
    if (lz) atomicMax(last_zero, lz);
    if (w) atomicAdd(n_warned, w);
}

/* minimum over the counted records behind the last timestamp of 0 */
__global__ void k_nano_scan2(const sq_nanoinfo *infos, uint64_t from, uint64_t n, long long *min_time)
{
    long long mn = INT64_MAX;
    for (uint64_t r = from + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; r < n;
         r += (uint64_t)gridDim.x * blockDim.x) {
        const long long t = infos[r].start_time;
        mn = t < mn ? t : mn;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const long long o = __shfl_xor(mn, off);
        mn = o < mn ? o : mn;
    }
    if ((threadIdx.x & 63) == 0 && mn != INT64_MAX) atomicMin(min_time, mn);
}

int nano_blocks(uint64_t n)
{
    uint64_t b = (n + 255) / 256;
    return (int)std::max<uint64_t>(1, std::min<uint64_t>(b, 8192));
}

} // namespace

struct sq_nanostats {
    sq_ctx *ctx;
    bool skipped = false;
    std::string skipped_reason;
    uint64_t number_of_reads = 0;
    int64_t min_time = 0, max_time = 0;
    sq_nanoinfo *d_infos = nullptr;
    size_t cap = 0;
    uint32_t *d_status = nullptr;
    size_t status_cap = 0;
    unsigned long long *d_scalars = nullptr; /* first_stop, max, last_zero, warned, min behind the last zero, min */
    std::vector<uint64_t> warnings;          /* counted pi lengths of the last add */
};

SQ_EXPORT sq_nanostats *sq_nanostats_new(sq_ctx *ctx)
{
    sq_nanostats *s = new sq_nanostats();
    s->ctx = ctx;
    SQ_HIP_NULL(hipMalloc((void **)&s->d_scalars, 6 * 8));
    return s;
}

SQ_EXPORT void sq_nanostats_free(sq_nanostats *s)
{
    if (!s) return;
    (void)hipStreamSynchronize(s->ctx->stream);
    for (void *p : {(void *)s->d_infos, (void *)s->d_status, (void *)s->d_scalars})
        if (p) (void)hipFree(p);
    delete s;
}

SQ_EXPORT int sq_nanostats_add_batch(sq_nanostats *s, sq_batch *b)
{
    sq_ctx *ctx = s->ctx;
    s->warnings.clear();
    const uint64_t n = b->n;
    if (s->skipped || n == 0) return SQ_OK; /* :5271 */
    auto reserve = [&](uint64_t records) -> int {   /* room for the NanoInfo and the status of `records` more records */
        if (s->number_of_reads + records > s->cap) {
            int rc = sq_grow_device(ctx, &s->d_infos, &s->cap, std::max<size_t>(s->number_of_reads + records, 2 * s->cap));
            if (rc) return rc;
        }
        if (records > s->status_cap) {
            if (s->d_status) { SQ_HIP(hipStreamSynchronize(ctx->stream)); SQ_HIP(hipFree(s->d_status)); }
            SQ_HIP(hipMalloc((void **)&s->d_status, records * 4));
            s->status_cap = records;
        }
        return SQ_OK;
    };
    const long long init[6] = {-1, INT64_MIN, 0, 0, INT64_MAX, INT64_MAX};
    /* The module stops for good at the first header that is no nanopore header (:5302-5312) -- for the reads of any other
       instrument that is record 0 of the first array it is handed.  A large batch is therefore tried on its first records
       first: where they already hold the stop, nothing behind them counts, and neither the rest of the batch is parsed
       nor room made for it (25 M Illumina headers: 13 ms and 1 GB, once per run). */
    uint64_t n_run = n;
    if (n > 4096) {
        if (int rc = reserve(256)) return rc;
        SQ_HIP(hipMemcpyAsync(s->d_scalars, init, sizeof init, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_nano_parse, dim3(nano_blocks(256)), dim3(256), 0, ctx->stream, b->d_buf, (uint64_t)b->buf_len,
                           b->d_metas, (uint64_t)256, s->d_infos + s->number_of_reads, s->d_status, s->d_scalars);
        SQ_HIP(hipMemcpyAsync(&ctx->pinned[48], s->d_scalars, 8, hipMemcpyDeviceToHost, ctx->stream));
        SQ_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->pinned[48] != ~0ULL) n_run = 256;
    }
    if (int rc = reserve(n_run)) return rc;
    sq_nanoinfo *infos = s->d_infos + s->number_of_reads;
    SQ_HIP(hipMemcpyAsync(s->d_scalars, init, sizeof init, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_nano_parse, dim3(nano_blocks(n_run)), dim3(256), 0, ctx->stream, b->d_buf, (uint64_t)b->buf_len,
                       b->d_metas, n_run, infos, s->d_status, s->d_scalars);
    SQ_HIP(hipGetLastError());
    SQ_HIP(hipMemcpyAsync(&ctx->pinned[48], s->d_scalars, 8, hipMemcpyDeviceToHost, ctx->stream));
    SQ_HIP(hipStreamSynchronize(ctx->stream));
    const uint64_t stop = ctx->pinned[48];
    const uint64_t counted = stop == ~0ULL ? n_run : stop;
    uint32_t stop_status = 0;
    if (stop != ~0ULL) SQ_HIP(hipMemcpy(&stop_status, s->d_status + stop, 4, hipMemcpyDeviceToHost));
    /* a pi warning of the record that raised was issued before it raised */
    const uint64_t n_status = stop == ~0ULL ? n_run : stop + 1;
    hipLaunchKernelGGL(k_nano_scan1, dim3(nano_blocks(n_status)), dim3(256), 0, ctx->stream, infos, s->d_status,
                       counted, n_status, (long long *)(s->d_scalars + 1), s->d_scalars + 2, s->d_scalars + 3,
                       (long long *)(s->d_scalars + 5));
    SQ_HIP(hipMemcpyAsync(&ctx->pinned[49], s->d_scalars + 1, 24, hipMemcpyDeviceToHost, ctx->stream));
    SQ_HIP(hipMemcpyAsync(&ctx->pinned[53], s->d_scalars + 5, 8, hipMemcpyDeviceToHost, ctx->stream));
    SQ_HIP(hipStreamSynchronize(ctx->stream));
    const long long batch_max = (long long)ctx->pinned[49];
    const uint64_t last_zero = ctx->pinned[50], warned = ctx->pinned[51];
    if (counted > last_zero) {
        hipLaunchKernelGGL(k_nano_scan2, dim3(nano_blocks(counted - last_zero)), dim3(256), 0, ctx->stream, infos,
                           last_zero, counted, (long long *)(s->d_scalars + 4));
        SQ_HIP(hipMemcpyAsync(&ctx->pinned[52], s->d_scalars + 4, 8, hipMemcpyDeviceToHost, ctx->stream));
        SQ_HIP(hipStreamSynchronize(ctx->stream));
    }
    if (counted) {
        /* :5315-5322: max is a plain maximum; min is the fold m = (m == 0 || t < m) ? t : m.
           Once m is negative only a smaller t moves it; while it is not, a timestamp of 0
           (no st tag) restarts it and a negative one takes it over */
        if (batch_max > s->max_time) s->max_time = batch_max;
        const long long batch_min = (long long)ctx->pinned[53];
        const long long suffix_min = counted > last_zero ? (long long)ctx->pinned[52] : 0;
        if (batch_min < 0) s->min_time = s->min_time < 0 ? std::min<long long>(s->min_time, batch_min) : batch_min;
        else if (s->min_time < 0) { /* stays */ }
        else if (last_zero) s->min_time = suffix_min;
        else if (s->min_time == 0 || batch_min < s->min_time) s->min_time = batch_min;
    }
    s->number_of_reads += counted;
    if (warned) {
        std::vector<uint32_t> st(n_status);
        SQ_HIP(hipMemcpy(st.data(), s->d_status, n_status * 4, hipMemcpyDeviceToHost));
        for (uint32_t w : st)
            if (w & NANO_PI_WARNING) s->warnings.push_back(w >> 8);
    }
    if (stop == ~0ULL) return SQ_OK;
    const uint32_t code = stop_status & 0x7F;
    const char c1 = (char)(stop_status >> 8), c2 = (char)(stop_status >> 16), c3 = (char)(stop_status >> 24);
    switch (code) {
        case NANO_BAD_HEADER: { /* :5302-5312: the module stops for good, no exception */
            sq_meta m;
            SQ_HIP(hipMemcpy(&m, b->d_metas + stop, sizeof m, hipMemcpyDeviceToHost));
            std::string name(m.name_length, ' ');
            if (m.name_length)
                SQ_HIP(hipMemcpy(&name[0], b->d_buf + m.record_start, m.name_length, hipMemcpyDeviceToHost));
            s->skipped = true;
            s->skipped_reason = "Can not parse header: " + sq_py_repr(name);
            return SQ_OK;
        }
        case NANO_TRUNCATED: sq_set_error("truncated tags"); return SQ_ERR_VALUE;
        case NANO_ARRAY_TYPE: sq_set_error("Invalid type for array %c", c1); return SQ_ERR_VALUE;
        case NANO_UNKNOWN_TYPE: sq_set_error("Unknown tag type %c", c1); return SQ_ERR_VALUE;
        case NANO_WRONG_TYPECODE:
            sq_set_error("Wrong tag type for '%c%c' expected '%c' got '%c'", c1, c2, (c1 == 'd') ? 'f' : 'Z', c3);
            return SQ_ERR_RUNTIME;
        default: /* :5221: NULL without an exception set */
            sq_set_error("ch tag holds no integer (the reference returns an error without setting an exception)");
            return SQ_ERR_SYSTEM;
    }
}

SQ_EXPORT int sq_nanostats_add(sq_nanostats *s, const uint8_t *buf, size_t buf_len, const sq_meta *metas, size_t n)
{
    sq_batch *b = sq_batch_upload(s->ctx, buf, buf_len, metas, n);
    if (!b) return SQ_ERR_MEMORY;
    int rc = sq_nanostats_add_batch(s, b);
    sq_batch_free(b);
    return rc;
}

SQ_EXPORT uint64_t sq_nanostats_number_of_reads(sq_nanostats *s) { return s->number_of_reads; }
SQ_EXPORT int64_t sq_nanostats_minimum_time(sq_nanostats *s) { return s->min_time; }
SQ_EXPORT int64_t sq_nanostats_maximum_time(sq_nanostats *s) { return s->max_time; }
SQ_EXPORT const char *sq_nanostats_skipped_reason(sq_nanostats *s)
{
    return s->skipped ? s->skipped_reason.c_str() : nullptr;
}

SQ_EXPORT int64_t sq_nanostats_infos(sq_nanostats *s, sq_nanoinfo *out, size_t cap)
{
    const uint64_t n = s->number_of_reads;
    if (!out || cap < n || n == 0) return (int64_t)n;
    SQ_HIP(hipStreamSynchronize(s->ctx->stream));
    SQ_HIP(hipMemcpy(out, s->d_infos, n * sizeof(sq_nanoinfo), hipMemcpyDeviceToHost));
    return (int64_t)n;
}

SQ_EXPORT int64_t sq_nanostats_last_warnings(sq_nanostats *s, uint64_t *lengths, size_t cap)
{
    const size_t n = s->warnings.size();
    if (lengths && cap >= n) std::copy(s->warnings.begin(), s->warnings.end(), lengths);
    return (int64_t)n;
}
