// Span streaming into LDS: every wave takes contiguous spans of SPAN bytes (16 FASTQ records of
// ~348 bytes) out of a big buffer, one span landing in LDS (LDS-DMA, or 16-byte loads through
// registers) while the span before it is consumed (ds_read_b128 + add).  Reports TB/s for
// several occupancies.  hipcc --offload-arch=gfx950 -O3 -o scripts/build/ubench_span scripts/ubench_span.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define SQ_LDS __attribute__((address_space(3)))

constexpr uint32_t SPAN = 5552 + 16;   /* 16 records of 347 bytes, + alignment slack */
constexpr uint32_t SLOT = 6144;        /* LDS bytes per landing buffer: 6 x 1 KiB */
constexpr uint32_t STEP = 5552;        /* distance between span starts */

template <int MODE>   /* 0: LDS-DMA, 1: registers + ds_write_b128, 2: LDS-DMA nt */
__global__ void __launch_bounds__(1024) k_span(const uint8_t *buf, uint64_t nspans, unsigned long long *out, int work)
{
    extern __shared__ __align__(16) uint8_t smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    uint8_t *ring = smem + (size_t)wave * 2 * SLOT;
    const uint64_t total_waves = (uint64_t)gridDim.x * nwaves;
    /* a workgroup takes nwaves consecutive spans per round */
    uint64_t s = (uint64_t)blockIdx.x * nwaves + wave;
    unsigned long long acc = 0;
    int cur = 0;
    auto issue = [&](uint64_t sp, int slot) {
        const uint64_t a0 = (7 + sp * STEP) & ~15ull;
        const uint8_t *g = buf + a0 + lane * 16;
        uint8_t *l = ring + slot * SLOT;
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int k = 0; k < 6; k++) {
                if (k < 5 || lane * 16 + 5120 < SPAN)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + k * 1024),
                                                     (SQ_LDS void *)(l + k * 1024), 16, 0, MODE == 2 ? 2 : 0);
            }
        }
    };
    if (MODE == 1) {
        /* register staging: loads of span k+1 are issued before the consume of span k, written after */
        u32x4 r[6];
        auto load = [&](uint64_t sp) {
            const uint64_t a0 = (7 + sp * STEP) & ~15ull;
            const uint8_t *g = buf + a0 + lane * 16;
#pragma unroll
            for (int k = 0; k < 6; k++) {
                r[k] = (u32x4){0, 0, 0, 0};
                if (k < 5 || lane * 16 + 5120 < SPAN) r[k] = *(const u32x4 *)(g + k * 1024);
            }
        };
        if (s < nspans) load(s);
        while (s < nspans) {
            uint8_t *l = ring + cur * SLOT;
#pragma unroll
            for (int k = 0; k < 6; k++) *(u32x4 *)(l + k * 1024 + lane * 16) = r[k];
            const uint64_t sn = s + total_waves;
            if (sn < nspans) load(sn);
            for (int w = 0; w < work; w++)
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    if (k * 1024 + lane * 16 < SPAN) {
                        const u32x4 v = *(const u32x4 *)(l + k * 1024 + lane * 16);
                        acc += (unsigned long long)v.x + v.y + v.z + v.w;
                    }
                }
            cur ^= 1;
            s = sn;
        }
    } else {
        if (s < nspans) issue(s, 0);
        while (s < nspans) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const uint64_t sn = s + total_waves;
            if (sn < nspans) issue(sn, cur ^ 1);
            const uint8_t *l = ring + cur * SLOT;
            for (int w = 0; w < work; w++)
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    if (k * 1024 + lane * 16 < SPAN) {
                        const u32x4 v = *(const u32x4 *)(l + k * 1024 + lane * 16);
                        acc += (unsigned long long)v.x + v.y + v.z + v.w;
                    }
                }
            cur ^= 1;
            s = sn;
        }
    }
    atomicAdd(out, acc);
}

int main()
{
    const uint64_t nspans = 1500000;                 /* 8.3 GB */
    const uint64_t bytes = 7 + nspans * STEP + 8192;
    uint8_t *d; unsigned long long *d_out;
    CK(hipMalloc(&d, bytes)); CK(hipMalloc(&d_out, 8));
    /* fill with a pattern whose per-span sums the host can reproduce: dword i = i * 2654435761 */
    {
        std::vector<uint32_t> h(1 << 22);
        for (uint64_t off = 0; off < bytes; off += (uint64_t)h.size() * 4) {
            const uint64_t n = std::min<uint64_t>(h.size() * 4, bytes - off);
            for (uint64_t i = 0; i < n / 4; i++) h[i] = (uint32_t)((off / 4 + i) * 2654435761u);
            CK(hipMemcpy(d + off, h.data(), n & ~3ull, hipMemcpyHostToDevice));
        }
    }
    /* expected: sum over spans of the dwords of [a0, a0 + SPAN) */
    unsigned long long expect = 0;
    for (uint64_t sp = 0; sp < nspans; sp++) {
        const uint64_t a0 = (7 + sp * STEP) & ~15ull;
        unsigned long long acc = 0;
        for (uint64_t i = a0 / 4; i < (a0 + SPAN) / 4; i++) acc += (uint32_t)(i * 2654435761u);
        expect += acc;
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; mode++)
        for (int waves : {8, 10, 12, 16})
            for (int work : {1, 8}) {
                const size_t lds = (size_t)waves * 2 * SLOT;
                if (lds > 160 * 1024) continue;
                auto fn = mode == 0 ? k_span<0> : mode == 1 ? k_span<1> : k_span<2>;
                CK(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                float best = 1e9;
                unsigned long long got = 0;
                for (int rep = 0; rep < 3; rep++) {
                    CK(hipMemset(d_out, 0, 8));
                    CK(hipEventRecord(e0));
                    hipLaunchKernelGGL(fn, dim3(256), dim3(waves * 64), lds, 0, d, nspans, d_out, work);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    best = std::min(best, ms);
                    CK(hipMemcpy(&got, d_out, 8, hipMemcpyDeviceToHost));
                }
                printf("mode %d (%s) waves/CU %2d consume x%d: %.3f ms = %.2f TB/s  checksum %s\n", mode,
                       mode == 0 ? "LDS-DMA" : mode == 1 ? "registers" : "LDS-DMA nt", waves, work, best,
                       (double)nspans * STEP / best / 1e9, got == expect * work ? "ok" : "MISMATCH");
            }
    return 0;
}
