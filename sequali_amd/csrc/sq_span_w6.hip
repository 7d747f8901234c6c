/*
 * sq_span_w6.hip -- the builds of k_span (sq_span_kernel.h) whose automaton is restarted SIX dwords (24 positions) in front
 * of a lane's quarter instead of three: adapters of 14 to 25 characters (AdapterCounter takes up to 64,
 * _qcmodule.c:2549-2591; the reference's own list holds 12-mers).  A quarter must be at least as long as the restart, so
 * these builds exist from 3 windows (65 positions) on: Q4 = 2 NW + 1 >= 7 dwords.  Round 5 ran them against the oracle
 * (tests/test_gpu_span_edges.py::test_adapters_of_14_to_25_characters_on_every_quarter_seam) and against k_wide
 * (profiles/r5/exp_w6.txt): the default from 129 bases on for batches of one read length and for every length-sorted batch
 * (SQ_SPAN_W6, sq_common.h).
 */
#include "sq_span_kernel.h"

namespace {

constexpr int W6 = 6;

template <int NW, bool SEG, bool SPLIT>
int launch_w6(sq_ctx *ctx, const PassParams &P0, uint32_t n_ad, int waves, size_t lds, int grid)
{
    PassParams P = P0;
    static bool attr = false;
    if (!attr) {
        SQ_HIP(hipFuncSetAttribute((const void *)k_span<NW, true, SEG, W6, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    sq_route(ctx, "k_span<%d,AD,%s,%s,w6>", NW, SEG ? "sorted" : "uniform", SPLIT ? "split" : "both");
    hipLaunchKernelGGL((k_span<NW, true, SEG, W6, SPLIT>), dim3(grid), dim3(waves * 64), lds, ctx->stream, P, n_ad);
    SQ_HIP(hipGetLastError());
    return SQ_OK;
}

template <int NW, bool SEG, bool SPLIT>
bool spills_w6()
{
    static int spills = -1;
    if (spills < 0) {
        hipFuncAttributes fa{};
        spills = hipFuncGetAttributes(&fa, (const void *)k_span<NW, true, SEG, W6, SPLIT>) == hipSuccess && fa.localSizeBytes > 0 ? 1 : 0;
    }
    return spills != 0;
}

}  // namespace

/* is there a build for this shape (one wave for both streams: 3 .. 5 windows; a wave per stream: 3 .. 8), and does it keep its
   registers */
bool sq_span_w6_exists(int nw, bool seg, bool split)
{
    if (nw < 3) return false;
    if (!split) return nw <= SPAN_NW_AD;
    return nw <= 8;
}

bool sq_span_w6_spills(int nw, bool seg, bool split)
{
    if (!sq_span_w6_exists(nw, seg, split)) return true;
#define W6_CASE(N, S, P) if (nw == N && seg == S && split == P) return spills_w6<N, S, P>();
    W6_CASE(3, false, false) W6_CASE(4, false, false) W6_CASE(5, false, false)
    W6_CASE(3, true, false) W6_CASE(4, true, false) W6_CASE(5, true, false)
    W6_CASE(3, false, true) W6_CASE(4, false, true) W6_CASE(5, false, true) W6_CASE(6, false, true) W6_CASE(7, false, true) W6_CASE(8, false, true)
    W6_CASE(3, true, true) W6_CASE(4, true, true) W6_CASE(5, true, true) W6_CASE(6, true, true) W6_CASE(7, true, true) W6_CASE(8, true, true)
#undef W6_CASE
    return true;
}

int sq_span_launch_w6(int nw, bool seg, bool split, sq_ctx *ctx, const PassParams &P, uint32_t n_ad, int waves, size_t lds, int grid)
{
#define W6_CASE(N, S, Q) if (nw == N && seg == S && split == Q) return launch_w6<N, S, Q>(ctx, P, n_ad, waves, lds, grid);
    W6_CASE(3, false, false) W6_CASE(4, false, false) W6_CASE(5, false, false)
    W6_CASE(3, true, false) W6_CASE(4, true, false) W6_CASE(5, true, false)
    W6_CASE(3, false, true) W6_CASE(4, false, true) W6_CASE(5, false, true) W6_CASE(6, false, true) W6_CASE(7, false, true) W6_CASE(8, false, true)
    W6_CASE(3, true, true) W6_CASE(4, true, true) W6_CASE(5, true, true) W6_CASE(6, true, true) W6_CASE(7, true, true) W6_CASE(8, true, true)
#undef W6_CASE
    sq_set_error("k_span: no build with a six-dword restart for %d windows", nw);
    return SQ_ERR_SYSTEM;
}
