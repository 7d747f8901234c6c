// LDS-DMA with unaligned per-row sources: every wave takes 16 consecutive "records" of REC bytes,
// and brings the 304 bytes behind byte 37 of each record (sequence + separator + qualities) to a
// 16-byte aligned row of its LDS slot: 16 rows x 19 pieces of 16 bytes = 304 lane loads = 5
// global_load_lds_dwordx4.  The source addresses have any byte alignment.  Reports TB/s of the
// record text covered (16 x REC per span) and checks a sample of row sums against the host.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define SQ_LDS __attribute__((address_space(3)))
constexpr uint32_t REC = 347, ROWB = 304, PIECES = 19, SLOT = 16 * ROWB + 256; /* 5 x 1024 = 5120 >= 4864 */
constexpr uint32_t SAMPLE = 4096;

__host__ __device__ inline uint8_t pat(uint64_t g) { return (uint8_t)((g * 0x9E3779B97F4A7C15ull) >> 56); }
__global__ void k_fill(uint8_t *buf, uint64_t n)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) buf[i] = pat(i);
}

template <int NT>
__global__ void __launch_bounds__(1024) k_rows(const uint8_t *buf, uint64_t nspans, unsigned long long *out, unsigned long long *sample)
{
    extern __shared__ __align__(16) uint8_t smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    uint8_t *ring = smem + (size_t)wave * 2 * SLOT;
    const uint64_t total_waves = (uint64_t)gridDim.x * nwaves;
    uint64_t s = (uint64_t)blockIdx.x * nwaves + wave;
    unsigned long long acc = 0;
    int cur = 0;
    auto issue = [&](uint64_t sp, int slot) {
        uint8_t *l = ring + slot * SLOT;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const uint32_t piece = k * 64 + lane;           /* 0 .. 319, 304 used */
            const uint32_t row = piece / PIECES, pk = piece - row * PIECES;
            const uint8_t *g = buf + (sp * 16 + row) * (uint64_t)REC + 37 + pk * 16;
            if (piece < 16 * PIECES)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (SQ_LDS void *)(l + k * 1024), 16, 0, NT ? 2 : 0);
        }
    };
    if (s < nspans) issue(s, 0);
    while (s < nspans) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint64_t sn = s + total_waves;
        if (sn < nspans) issue(sn, cur ^ 1);
        const uint8_t *l = ring + cur * SLOT;
        /* lane = (row, quarter): sum the dwords of its quarter of the row */
        const uint32_t row = lane >> 2, c = lane & 3;
        unsigned long long rs = 0;
        for (uint32_t d = c; d < ROWB / 4; d += 4) rs += *(const uint32_t *)(l + row * ROWB + d * 4);
        rs += __shfl_xor(rs, 1); rs += __shfl_xor(rs, 2);
        if (c == 0) { acc += rs; if (s < SAMPLE) sample[s * 16 + row] = rs; }
        cur ^= 1;
        s = sn;
    }
    atomicAdd(out, acc);
}

int main()
{
    const uint64_t nspans = 1500000;
    const uint64_t bytes = nspans * 16 * REC + 4096;
    uint8_t *d; unsigned long long *d_out, *d_sample;
    CK(hipMalloc(&d, bytes)); CK(hipMalloc(&d_out, 8)); CK(hipMalloc(&d_sample, SAMPLE * 16 * 8));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, d, bytes);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> expect(SAMPLE * 16);
    for (uint64_t sp = 0; sp < SAMPLE; sp++)
        for (uint32_t row = 0; row < 16; row++) {
            unsigned long long rs = 0;
            const uint64_t g0 = (sp * 16 + row) * (uint64_t)REC + 37;
            for (uint32_t dw = 0; dw < ROWB / 4; dw++) {
                uint32_t v = 0;
                for (int b = 0; b < 4; b++) v |= (uint32_t)pat(g0 + dw * 4 + b) << (8 * b);
                rs += v;
            }
            expect[sp * 16 + row] = rs;
        }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int nt = 0; nt < 2; nt++)
        for (int waves : {8, 12, 16}) {
            const size_t lds = (size_t)waves * 2 * SLOT;
            if (lds > 160 * 1024) continue;
            auto fn = nt ? k_rows<1> : k_rows<0>;
            CK(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            float best = 1e9;
            int bad = 0;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipMemset(d_out, 0, 8)); CK(hipMemset(d_sample, 0, SAMPLE * 16 * 8));
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(fn, dim3(256), dim3(waves * 64), lds, 0, d, nspans, d_out, d_sample);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                best = std::min(best, ms);
            }
            std::vector<unsigned long long> got(SAMPLE * 16);
            CK(hipMemcpy(got.data(), d_sample, SAMPLE * 16 * 8, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < got.size(); i++) bad += got[i] != expect[i];
            printf("row DMA%s waves/CU %2d: %.3f ms = %.2f TB/s of record text; sampled row sums: %d wrong of %zu\n", nt ? " nt" : "   ", waves, best,
                   (double)nspans * 16 * REC / best / 1e9, bad, got.size());
        }
    return 0;
}
