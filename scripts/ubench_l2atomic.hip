// round 5: what do f64 atomic adds into a small table cost when every CU issues them -- PerTileQuality's table
// [tiles][positions] (96 x 150 doubles = 115 KB) for reads of RANDOM tiles cannot live in k_span's LDS; could the pass add to
// it in the L2 instead?  A wave-instruction adds 64 doubles: lanes 0-31 to 32 consecutive positions of one random tile's row,
// lanes 32-63 to those of another (what lane (h, pl) of k_span holds per window).  Variants: one table for the device (agent
// scope), one table per XCD with workgroup-scope atomics (the XCD's own L2 performs them; a merge afterwards), and plain
// stores of the same shape for comparison.   Needed: 25 M reads x 150 positions = 3.75 G adds inside a 3.3 ms pass.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o scripts/build/ubench_l2atomic scripts/ubench_l2atomic.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int TILES = 96, U = 160;   // rows of 160 doubles (1280 bytes)

__device__ __forceinline__ uint32_t xcc_id()
{
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xF;
}

template <int MODE>   // 0: agent-scope atomics, one table; 1: workgroup-scope atomics, a table per XCD; 2: plain stores (no atomic); 3: agent scope, a table per XCD
__global__ void __launch_bounds__(768) k(double *table, uint64_t iters, uint32_t *xcc_seen)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, pl = lane & 31;
    const uint32_t xcc = xcc_id();
    if (threadIdx.x == 0) atomicOr(&xcc_seen[blockIdx.x], 1u << xcc);
    double *t = table + (MODE == 1 || MODE == 3 ? (size_t)xcc * TILES * U : 0);
    uint32_t rng = (blockIdx.x * 12 + wave) * 2654435761u + 12345u;
    const double v = 1e-3 * (1 + (lane & 7));
    for (uint64_t it = 0; it < iters; it++) {
        // a span: 16 rows x 5 windows = 40 wave-instructions (two rows each)
#pragma unroll 1
        for (int k = 0; k < 8; k++) {
            rng = rng * 1664525u + 1013904223u;
            const uint32_t tile_a = (rng >> 8) % TILES, tile_b = (rng >> 20) % TILES;
            double *row = t + (size_t)(h ? tile_b : tile_a) * U + pl;
#pragma unroll
            for (int w = 0; w < 5; w++) {
                double *p = row + 32 * w;
                if (MODE == 0 || MODE == 3) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else if (MODE == 1) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else *(volatile double *)p = v;
            }
        }
    }
}

int main()
{
    double *table;
    uint32_t *seen;
    const size_t bytes = (size_t)16 * TILES * U * sizeof(double);
    CK(hipMalloc(&table, bytes));
    CK(hipMalloc(&seen, 4096 * 4));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const uint64_t iters = 2000;   // spans per wave
    const char *names[4] = {"agent scope, one table", "workgroup scope, a table per XCD (checked below)", "plain stores (not atomic)", "agent scope, a table per XCD"};
    for (int mode = 0; mode < 4; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            CK(hipMemset(table, 0, bytes));
            CK(hipMemset(seen, 0, 4096 * 4));
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0));
            if (mode == 0) k<0><<<cus, 768>>>(table, iters, seen);
            else if (mode == 1) k<1><<<cus, 768>>>(table, iters, seen);
            else if (mode == 2) k<2><<<cus, 768>>>(table, iters, seen);
            else k<3><<<cus, 768>>>(table, iters, seen);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double adds = (double)cus * 12 * iters * 40 * 64;
            if (rep == 1) {
                std::vector<double> hst(16 * TILES * U);
                CK(hipMemcpy(hst.data(), table, bytes, hipMemcpyDeviceToHost));
                double sum = 0;
                for (double x : hst) sum += x;
                const double want = (double)cus * 12 * iters * 40 * 8 * (1e-3 * 36);   // per wave-instruction: lanes' values 1..8 e-3, eight times each
                std::vector<uint32_t> hs(4096);
                CK(hipMemcpy(hs.data(), seen, 4096 * 4, hipMemcpyDeviceToHost));
                uint32_t all = 0;
                for (int i = 0; i < cus; i++) all |= hs[i];
                printf("%-50s %8.3f ms  %7.1f G adds/s  (3.75 G adds: %.2f ms)  sum/expected %.9f  XCDs seen 0x%x\n", names[mode], ms, adds / ms / 1e6, 3.75e9 / (adds / ms) , sum / want, all);
            }
        }
    }
    return 0;
}
