/*
 * sq_synth_core.h -- counter-based synthetic FASTQ, the
 * build's own replacement for the reference's scripts/fastq_create.py.
 *
 * Every byte of record i is a pure function of (kind, seed, i, byte offset),
 * written with integer arithmetic only, so the host loop (sq_synth_host) and
 * the HIP kernel (sq_synth_device) produce identical bytes and records can be
 * generated in any order / on any rank.
 *
 * Illumina (kind 0 = R1 / single end, kind 1 = R2 of the same pair, kind 3 = kind 0 with the
 * reads ordered by tile, 65536 in a row, as a sequencer writes them):
 *   pair i has a fragment F_i: length ~ N(300,60) clipped to [40,600] for 92 %
 *   of the pairs, uniform [40,150) for 8 % (adapter read-through); 10 % of the
 *   pairs re-use the fragment of an earlier pair (duplicates); bases uniform
 *   ACGT.  R1 = F[:150], R2 = revcomp(F)[:150]; a read longer than the
 *   fragment continues with the Illumina adapter (R1: AGATCGGAAGAGCACACGTCTG
 *   AACTCCAGTCA, R2: AGATCGGAAGAGCGTCGTGTAGGGAAAGAGTGT) and then poly-G.  Each
 *   read base turns into N with p = 1/1024.  Qualities per base from
 *   {Q37 'F', Q25 ':', Q11 ',', Q2 '#'} with weights .80/.12/.06/.02, the Q37
 *   weight falling linearly to .55 at the last base.  Header
 *   SIM:1:FCX:<lane>:<tile>:<x>:<y> <1|2>:N:0:ATCCGA, 96 tile ids, all fields
 *   fixed width so that every record is 348 bytes.
 * Nanopore (kind 2): length = clip(8000 * exp(0.6 z), 200, 100000), z ~ N(0,1)
 *   (fixed-point exp2), bases uniform ACGT, qualities Q3..Q35 around a
 *   per-read mean, header <uuid> runid=<hex> read=<i> ch=<n> start_time=...
 *   (fixed width, 139 bytes).
 */
#ifndef SQ_SYNTH_CORE_H
#define SQ_SYNTH_CORE_H

#include <stdint.h>

#ifdef __HIPCC__
#define SQ_HD __host__ __device__ inline
#else
#define SQ_HD static inline
#endif

#define SQ_SYNTH_READ_LEN 150
#define SQ_SYNTH_ILLUMINA_NAME 42
#define SQ_SYNTH_NANOPORE_NAME 139

SQ_HD uint64_t sqs_mix(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

/* one 64-bit random word per (seed, stream, index, counter) */
SQ_HD uint64_t sqs_rand(uint64_t seed, uint64_t stream, uint64_t index, uint64_t ctr)
{
    return sqs_mix(sqs_mix(sqs_mix(seed ^ (stream * 0xD1342543DE82EF95ULL)) + index) + ctr);
}

enum { SQS_FRAG = 1, SQS_DUP = 2, SQS_BASE = 3, SQS_NMASK = 4, SQS_QUAL = 5, SQS_HEAD = 6, SQS_LEN = 7 };

/* Irwin-Hall: sum of 12 uniform u16 -> approximately N(393210, 65536^2) */
SQ_HD int64_t sqs_normal_q16(uint64_t seed, uint64_t stream, uint64_t index)
{
    int64_t s = 0;
    for (int k = 0; k < 3; k++) {
        uint64_t r = sqs_rand(seed, stream, index, 100 + k);
        s += (int64_t)(r & 0xFFFF) + (int64_t)((r >> 16) & 0xFFFF) +
             (int64_t)((r >> 32) & 0xFFFF) + (int64_t)(r >> 48);
    }
    return s - 393210; /* z in Q16: z = value / 65536 */
}

/* the pair whose fragment pair i uses (follows the duplicate chain) */
SQ_HD uint64_t sqs_source_pair(uint64_t seed, uint64_t i)
{
    for (int hop = 0; hop < 64 && i > 0; hop++) {
        uint64_t r = sqs_rand(seed, SQS_DUP, i, 0);
        if ((r & 0x3FF) >= 102) break; /* ~10 % duplicates */
        i = (r >> 10) % i;
    }
    return i;
}

SQ_HD uint32_t sqs_fragment_length(uint64_t seed, uint64_t src)
{
    uint64_t r = sqs_rand(seed, SQS_FRAG, src, 0);
    if ((r & 0xFF) < 20) /* ~8 % short inserts */
        return 40 + (uint32_t)((r >> 8) % 110);
    int64_t z = sqs_normal_q16(seed, SQS_FRAG, src);
    int64_t len = 300 + (60 * z) / 65536;
    if (len < 40) len = 40;
    if (len > 600) len = 600;
    return (uint32_t)len;
}

SQ_HD uint8_t sqs_fragment_base(uint64_t seed, uint64_t src, uint32_t k)
{
    uint64_t r = sqs_rand(seed, SQS_BASE, src, k >> 5);
    return (uint8_t)"ACGT"[(r >> (2 * (k & 31))) & 3];
}

SQ_HD uint8_t sqs_complement(uint8_t c)
{
    return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A';
}

/* base p (0..149) of read `mate` (0 = R1, 1 = R2) of pair i */
SQ_HD uint8_t sqs_illumina_base(uint64_t seed, uint64_t i, uint64_t src, uint32_t flen, int mate,
                                uint32_t p)
{
    const char *ad = mate ? "AGATCGGAAGAGCGTCGTGTAGGGAAAGAGTGT" : "AGATCGGAAGAGCACACGTCTGAACTCCAGTCA";
    uint8_t c;
    if (p < flen)
        c = mate ? sqs_complement(sqs_fragment_base(seed, src, flen - 1 - p))
                 : sqs_fragment_base(seed, src, p);
    else if (p - flen < 33)
        c = (uint8_t)ad[p - flen];
    else
        c = 'G';
    uint64_t r = sqs_rand(seed, SQS_NMASK + 16 * (uint64_t)mate, i, p >> 2);
    if (((r >> (16 * (p & 3))) & 0x3FF) == 0) c = 'N';
    return c;
}

SQ_HD uint8_t sqs_illumina_qual(uint64_t seed, uint64_t i, int mate, uint32_t p, uint32_t L = SQ_SYNTH_READ_LEN)
{
    uint64_t r = sqs_rand(seed, SQS_QUAL + 16 * (uint64_t)mate, i, p >> 2);
    uint32_t u = (uint32_t)(r >> (16 * (p & 3))) & 0xFFFF;
    /* Q37 weight 0.80 -> 0.55 over the read, in 1/65536 units */
    uint32_t w37 = 52429 - (16384 * p) / (L > 1 ? L - 1 : 1);
    if (u < w37) return 'F';
    uint32_t v = ((u - w37) * 10) / (65536 - w37); /* 0..9 over the rest: 6:3:1 */
    return v < 6 ? ':' : v < 9 ? ',' : '#';
}

SQ_HD void sqs_put_dec(uint8_t *dst, uint64_t v, int width)
{
    for (int k = width - 1; k >= 0; k--) {
        dst[k] = (uint8_t)('0' + v % 10);
        v /= 10;
    }
}

/* header of pair i, mate 0/1: exactly SQ_SYNTH_ILLUMINA_NAME bytes */
SQ_HD void sqs_illumina_name(uint64_t seed, uint64_t i, int mate, uint8_t *dst, int by_tile = 0)
{
    const char *pre = "SIM:1:FCX:";
    const char *post = ":N:0:ATCCGA";
    uint64_t r = sqs_rand(seed, SQS_HEAD, i, 0);
    /* by_tile: the order a sequencer writes, 65536 reads of a tile in a row */
    uint32_t t = by_tile ? (uint32_t)((i >> 16) % 96) : (uint32_t)((r >> 8) % 96);
    uint32_t tile = (t / 48 + 1) * 1000 + ((t / 24) % 2 + 1) * 100 + (t % 24) + 1;
    for (int k = 0; k < 10; k++) dst[k] = (uint8_t)pre[k];
    dst[10] = (uint8_t)('1' + (r & 3));
    dst[11] = ':';
    sqs_put_dec(dst + 12, tile, 4);
    dst[16] = ':';
    sqs_put_dec(dst + 17, 10000 + (r >> 20) % 20000, 5);
    dst[22] = ':';
    sqs_put_dec(dst + 23, 100000 + i % 900000, 6);
    dst[29] = ' ';
    dst[30] = (uint8_t)('1' + mate);
    for (int k = 0; k < 11; k++) dst[31 + k] = (uint8_t)post[k];
}

/* ---- nanopore ---------------------------------------------------------- */
/* 2^(x/65536) * 65536 for x in Q16, integer only (cubic on the fraction) */
SQ_HD uint64_t sqs_exp2_q16(int64_t x)
{
    int64_t ip = x >> 16; /* floor */
    uint64_t f = (uint64_t)(x & 0xFFFF);
    /* 2^f ~ 1 + f*(0.6931472 + f*(0.2402265 + f*0.0555041)) in Q16 */
    uint64_t p = 3638; /* 0.0555041 * 65536 */
    p = 15743 + ((p * f) >> 16);
    p = 45426 + ((p * f) >> 16);
    p = 65536 + ((p * f) >> 16);
    if (ip >= 0) return p << ip;
    return p >> (-ip);
}

SQ_HD uint32_t sqs_nanopore_length(uint64_t seed, uint64_t i)
{
    int64_t z = sqs_normal_q16(seed, SQS_LEN, i);
    /* exp(0.6 z) = 2^(0.6 z / ln 2) = 2^(0.8656170 z) */
    int64_t e = (z * 56729) >> 16;
    uint64_t len = (8000 * sqs_exp2_q16(e)) >> 16;
    if (len < 200) len = 200;
    if (len > 100000) len = 100000;
    return (uint32_t)len;
}

SQ_HD uint8_t sqs_nanopore_base(uint64_t seed, uint64_t i, uint32_t p)
{
    uint64_t r = sqs_rand(seed, SQS_BASE + 32, i, p >> 5);
    return (uint8_t)"ACGT"[(r >> (2 * (p & 31))) & 3];
}

SQ_HD uint8_t sqs_nanopore_qual(uint64_t seed, uint64_t i, uint32_t p)
{
    uint64_t m = sqs_rand(seed, SQS_QUAL + 32, i, 0xFFFFFFFFu);
    int q = 8 + (int)(m % 20);
    uint64_t r = sqs_rand(seed, SQS_QUAL + 32, i, p >> 3);
    int n = (int)((r >> (8 * (p & 7))) & 0xFF);
    q += (n & 7) + ((n >> 3) & 7) - 7; /* triangular noise in [-7, 7] */
    if (q < 3) q = 3;
    if (q > 35) q = 35;
    return (uint8_t)(33 + q);
}

SQ_HD void sqs_nanopore_name(uint64_t seed, uint64_t i, uint8_t *dst)
{
    const char *hex = "0123456789abcdef";
    uint64_t a = sqs_rand(seed, SQS_HEAD + 32, i, 0), b = sqs_rand(seed, SQS_HEAD + 32, i, 1);
    int o = 0;
    for (int k = 0; k < 32; k++) {
        if (k == 8 || k == 12 || k == 16 || k == 20) dst[o++] = '-';
        uint64_t w = k < 16 ? a : b;
        dst[o++] = (uint8_t)hex[(w >> (4 * (k & 15))) & 15];
    }
    const char *s1 = " runid=";
    for (int k = 0; k < 7; k++) dst[o++] = (uint8_t)s1[k];
    uint64_t run0 = sqs_mix(seed), run1 = sqs_mix(seed + 1), run2 = sqs_mix(seed + 2);
    for (int k = 0; k < 40; k++) {
        uint64_t w = k < 16 ? run0 : k < 32 ? run1 : run2;
        dst[o++] = (uint8_t)hex[(w >> (4 * (k & 15))) & 15];
    }
    const char *s2 = " read=";
    for (int k = 0; k < 6; k++) dst[o++] = (uint8_t)s2[k];
    sqs_put_dec(dst + o, i, 10);
    o += 10;
    const char *s3 = " ch=";
    for (int k = 0; k < 4; k++) dst[o++] = (uint8_t)s3[k];
    sqs_put_dec(dst + o, 1 + (a >> 40) % 2048, 4);
    o += 4;
    const char *s4 = " start_time=2021-09-30T11:34:08Z";
    for (int k = 0; k < 32; k++) dst[o++] = (uint8_t)s4[k];
}

/* ---- record geometry ---------------------------------------------------- */
/* kind: bits 0-7 the kind proper (0 R1, 1 R2, 2 nanopore, 3 R1 by tile, 4 R2 by tile), bits 8-23 the read length of
   the Illumina kinds (0: SQ_SYNTH_READ_LEN) */
SQ_HD int sqs_base_kind(int kind) { return kind & 0xFF; }
SQ_HD int sqs_kind_mate(int kind) { return sqs_base_kind(kind) == 1 || sqs_base_kind(kind) == 4; }
SQ_HD int sqs_kind_by_tile(int kind) { return sqs_base_kind(kind) == 3 || sqs_base_kind(kind) == 4; }
SQ_HD uint32_t sqs_name_length(int kind) { return sqs_base_kind(kind) == 2 ? SQ_SYNTH_NANOPORE_NAME : SQ_SYNTH_ILLUMINA_NAME; }
SQ_HD uint32_t sqs_read_length(int kind, uint64_t seed, uint64_t i)
{
    if (sqs_base_kind(kind) == 2) return sqs_nanopore_length(seed, i);
    return (kind >> 8) ? (uint32_t)(kind >> 8) & 0xFFFF : SQ_SYNTH_READ_LEN;
}
/* '@' name '\n' seq '\n+\n' qual '\n' */
SQ_HD uint64_t sqs_record_bytes(int kind, uint64_t seed, uint64_t i)
{
    return 1 + (uint64_t)sqs_name_length(kind) + 1 + 2 * (uint64_t)sqs_read_length(kind, seed, i) + 4;
}

#endif
