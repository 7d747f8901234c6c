/*
 * sq_pair.hip -- the builds of k_span (sq_span_kernel.h) that carry other modules' work through QCMetrics' pass, and
 * what stands behind them: config 3 of BASELINE.json ((QCMetrics + PerTileQuality) x 2 + InsertSizeMetrics on pairs)
 * read every batch four times -- k_tile_parse, k_span, k_ptspan, k_isz_span -- and was bound by those reads.
 *
 *   k_span<NW, ., PT>   PerTileQuality_add_meta (_qcmodule.c:3123-3222) inside the pass: the tile id from the header
 *                       bytes of the lines the pass fetches anyway (illumina_header_to_tile_id :3088-3121), the error
 *                       rates of a run of reads of one tile summed in registers, runs staged (PtRun + a row of sums)
 *   k_pt_tail           the same for the records behind the last full span: a wave per record, a run of one read
 *   k_pt_runs_assign    the table row of every staged run (the tile map of sq_qc.hip: first come, first served)
 *   k_pt_fold           runs added to PerTileQuality's tables
 *
 * Nothing reaches the tables before the whole batch has parsed: the reference stops counting at the first header
 * that does not parse (:3137-3148), and a batch that holds one is counted by the older route (sq_qc.hip), which
 * knows where to stop.
 */
#include "sq_span_kernel.h"

namespace {

/* the records behind the last full span (fewer than 16): a wave per record */
__global__ void __launch_bounds__(64) k_pt_tail(PassParams P, uint64_t first, uint64_t n)
{
    const uint64_t r = first + blockIdx.x;
    if (r >= n) return;
    const uint32_t lane = threadIdx.x, U = P.uniform_len;
    const sq_meta m = P.metas[r];
    long long tile = 0;
    if (lane == 0) {
        tile = tile_id_of(P.buf + m.record_start, m.name_length);
        if (tile < 0) atomicMin(P.pt_bad, (unsigned long long)(P.pt_first_index + r));
        if (P.pt_tiles) P.pt_tiles[r] = tile;
    }
    tile = __shfl(tile, 0);
    if (tile < 0 || !P.pt_runs) return;
    uint32_t idx = 0;
    if (lane == 0) idx = atomicAdd(P.pt_nruns, 1u);
    idx = __shfl(idx, 0);
    if (idx >= P.pt_runs_cap) return;
    if (lane == 0) {
        PtRun run;
        run.tile = tile;
        run.reads = 1;
        run.pad = 0;
        P.pt_runs[idx] = run;
    }
    const uint8_t *q = P.buf + m.record_start + m.qualities_offset;
    for (uint32_t pos = lane; pos < U; pos += 64) {
        const uint32_t ph = (uint32_t)q[pos] - 33u;   /* :3189-3220; what is no phred character: NaN, as in k_span */
        P.pt_run_sums[(uint64_t)idx * U + pos] = ph <= SQ_PHRED_MAX ? __longlong_as_double((long long)c_error_rate_bits[ph])
                                                                    : __longlong_as_double(0x7FF8000000000000LL);
    }
}

/* the table row of every run: a lane per run, the lock-free tile map of k_tile_assign (sq_qc.hip) */
__global__ void __launch_bounds__(256) k_pt_runs_assign(PtRun *runs, uint32_t n_runs, long long *keys, int *vals, int *n_slots, int *overflow)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_runs) return;
    const long long tile = runs[i].tile;
    uint32_t idx = (uint32_t)(((unsigned long long)tile * 0x9E3779B97F4A7C15ULL) >> 48) & (TILE_MAP_SIZE - 1);
    int slot = -1;
    for (uint32_t probes = 0; slot < 0 && probes < 4 * TILE_MAP_SIZE; probes++) {
        long long k = __hip_atomic_load(&keys[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (k == TILE_EMPTY) {
            const long long old = (long long)atomicCAS((unsigned long long *)&keys[idx], (unsigned long long)TILE_EMPTY, (unsigned long long)tile);
            if (old == TILE_EMPTY) {
                slot = atomicAdd(n_slots, 1);
                __hip_atomic_store(&vals[idx], slot, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            k = old;
        }
        if (k == tile) {
            const int v = __hip_atomic_load(&vals[idx], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            if (v >= 0) slot = v;   /* else: the lane that made the entry publishes it presently */
        } else {
            idx = (idx + 1) & (TILE_MAP_SIZE - 1);
        }
    }
    if (slot < 0) { *overflow = 1; return; }
    runs[i].pad = (uint32_t)slot;
}

/* runs to the tables: a workgroup per run (PerTileQuality_add_meta :3186-3220 summed over the run's reads) */
__global__ void __launch_bounds__(256) k_pt_fold(const PtRun *runs, const double *sums, uint32_t U, double *errors, unsigned long long *len_counts, uint64_t cap)
{
    const PtRun run = runs[blockIdx.x];
    const uint64_t row = (uint64_t)run.pad * cap;
    if (threadIdx.x == 0) atomicAdd(&len_counts[row + (U - 1)], (unsigned long long)run.reads);
    for (uint32_t pos = threadIdx.x; pos < U; pos += blockDim.x)
        unsafeAtomicAdd(&errors[row + pos], sums[(uint64_t)blockIdx.x * U + pos]);
}

template <int NW, int PAIR>
int launch_pt(sq_ctx *ctx, const PassParams &P, int waves, size_t lds, int grid)
{
    static bool attr = false;
    if (!attr) {
        SQ_HIP(hipFuncSetAttribute((const void *)k_span<NW, false, false, SPAN_W4, false, false, true, PAIR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    sq_route(ctx, PAIR == 2 ? "k_span<%d,QCPT_scan,uniform,both>" : PAIR == 1 ? "k_span<%d,QCPT_ends,uniform,both>" : "k_span<%d,QCPT,uniform,both>", NW);
    hipLaunchKernelGGL((k_span<NW, false, false, SPAN_W4, false, false, true, PAIR>), dim3(grid), dim3(waves * 64), lds, ctx->stream, P, 0u);
    SQ_HIP(hipGetLastError());
    return SQ_OK;
}
template <int NW>
int launch_pt_any(sq_ctx *ctx, const PassParams &P, int pair, int waves, size_t lds, int grid)
{
    return pair == 2 ? launch_pt<NW, 2>(ctx, P, waves, lds, grid) : pair == 1 ? launch_pt<NW, 1>(ctx, P, waves, lds, grid) : launch_pt<NW, 0>(ctx, P, waves, lds, grid);
}

} // namespace

/* QCMetrics over the first 16 * (n / 16) records of a batch of one read length with PerTileQuality's tile ids and
 * staged runs on the way (P.pt_*), the records behind them by k_pt_tail (PerTileQuality only: QCMetrics' share of
 * those is the caller's).  *done = records QCMetrics has counted (0: the kernel does not take this batch and nothing
 * has been queued). */
int sq_span_launch_pt(sq_ctx *ctx, const PassParams &P, int pair, uint64_t *done)
{
    *done = 0;
    const uint32_t U = P.uniform_len;
    if (!U || U > 32 * SPAN_NW_MAX || P.n < SPAN_R) return SQ_OK;
    if (pair && (U < 16 || (pair == 2 && P.pair_L2 < 16))) return SQ_OK;   /* the needles are 16 bases */
    const int nw = (int)((U + 31) / 32);
    int waves = span_max_waves(nw, false, false, false, true);
    while (waves >= 4 && span_lds_layout(nw, U, 0, 0, 0, waves, false, false, false, pair == 2).total > 160 * 1024) waves--;
    if (waves < 4) return SQ_OK;
    const size_t lds = span_lds_layout(nw, U, 0, 0, 0, waves, false, false, false, pair == 2).total;
    PassParams C = P;
    C.n = (P.n / SPAN_R) * SPAN_R;
    const uint64_t nspans = C.n / SPAN_R;
    const int grid = (int)std::max<uint64_t>(1, std::min<uint64_t>((nspans + waves - 1) / waves, (uint64_t)ctx->num_cus));
    int rc;
    switch (nw) {
        case 1: rc = launch_pt_any<1>(ctx, C, pair, waves, lds, grid); break;
        case 2: rc = launch_pt_any<2>(ctx, C, pair, waves, lds, grid); break;
        case 3: rc = launch_pt_any<3>(ctx, C, pair, waves, lds, grid); break;
        case 4: rc = launch_pt_any<4>(ctx, C, pair, waves, lds, grid); break;
        case 5: rc = launch_pt_any<5>(ctx, C, pair, waves, lds, grid); break;
        case 6: rc = launch_pt_any<6>(ctx, C, pair, waves, lds, grid); break;
        case 7: rc = launch_pt_any<7>(ctx, C, pair, waves, lds, grid); break;
        default: rc = launch_pt_any<8>(ctx, C, pair, waves, lds, grid); break;
    }
    if (rc) return rc;
    if (C.n < P.n) {
        hipLaunchKernelGGL(k_pt_tail, dim3((unsigned)(P.n - C.n)), dim3(64), 0, ctx->stream, P, (uint64_t)C.n, (uint64_t)P.n);
        SQ_HIP(hipGetLastError());
    }
    *done = C.n;
    return SQ_OK;
}

/* (the scan of read 1's pass covers the pairs of the full spans; the caller hands the rest to k_insert_size) */

/* the staged runs [0, n_runs) get their table rows (runs[i].pad); *n_slots afterwards = rows in use */
int sq_pt_runs_assign(sq_ctx *ctx, PtRun *runs, uint32_t n_runs, long long *keys, int *vals, int *n_slots, int *overflow)
{
    if (!n_runs) return SQ_OK;
    hipLaunchKernelGGL(k_pt_runs_assign, dim3((n_runs + 255) / 256), dim3(256), 0, ctx->stream, runs, n_runs, keys, vals, n_slots, overflow);
    SQ_HIP(hipGetLastError());
    return SQ_OK;
}

int sq_pt_fold(sq_ctx *ctx, const PtRun *runs, const double *sums, uint32_t n_runs, uint32_t U, double *errors, unsigned long long *len_counts, uint64_t cap)
{
    if (!n_runs) return SQ_OK;
    sq_route(ctx, "k_pt_fold");
    hipLaunchKernelGGL(k_pt_fold, dim3(n_runs), dim3(256), 0, ctx->stream, runs, sums, U, errors, len_counts, cap);
    SQ_HIP(hipGetLastError());
    return SQ_OK;
}

/* What k_span<PT> makes of a header (illumina_header_to_tile_id, _qcmodule.c:3088-3121): the first 64 bytes in registers
 * (tile_id_of_words<8>), a longer header or a tile of 9 .. 18 digits byte by byte (tile_id_of) -- the same functions,
 * compiled for the host, so that the parse can be checked without a GPU (tests/test_boundary_cpu.py).  `name` must have 64
 * readable bytes (the kernel's loads fetch 64 whatever the name's length; what lies behind the name must not matter). */
SQ_EXPORT int64_t sq_test_tile_of_header(const uint8_t *name, uint32_t n)
{
    long long tile = -2;
    if (n <= 64) {
        uint64_t w[8];
        memcpy(w, name, 64);
        tile = tile_id_of_words<8>(w, n);
    }
    if (tile == -2) tile = tile_id_of(name, n);
    return tile;
}
/* the same through the parse the four lanes of a quad share (quad_tile_id_host, sq_pass.h): what the kernel's DPP exchanges
   compute, emulated lane by lane; the byte-by-byte parse where that one declines */
SQ_EXPORT int64_t sq_test_tile_of_header_quad(const uint8_t *name, uint32_t n)
{
    long long tile = quad_tile_id_host(name, n);
    if (tile == QUAD_TILE_SLOW) tile = tile_id_of(name, n);
    return tile;
}
