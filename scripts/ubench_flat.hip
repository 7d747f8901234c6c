// Calibration of FETCH_SIZE: k_flat reads a buffer of known size exactly once, as one linear
// stream of 16-byte loads (64 lanes x 16 B = 1 KB per wave instruction), and sums it.
// Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and compare FETCH_SIZE x 1024 with the
// bytes printed here (profiles/r2b/fetch_calibration.txt).
// Build: hipcc --offload-arch=gfx950 -O3 scripts/ubench_flat.hip -o scripts/build/ubench_flat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void __launch_bounds__(256) k_flat(const uint4 *buf, uint64_t n16, unsigned long long *sink)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long acc = 0;
    for (; i < n16; i += stride) {
        const uint4 v = buf[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 0x123456789ULL) *sink = acc;
}
__global__ void k_fill(uint32_t *buf, uint64_t n)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) buf[i] = (uint32_t)(i * 2654435761u);
}
int main(int argc, char **argv)
{
    const uint64_t bytes = (argc > 1 ? strtoull(argv[1], 0, 10) : 8ULL) << 30;
    uint4 *buf; unsigned long long *sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 8));
    k_fill<<<4096, 256>>>((uint32_t *)buf, bytes / 4);
    CK(hipDeviceSynchronize());
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(a));
        k_flat<<<256 * 8, 256>>>(buf, bytes / 16, sink);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("k_flat: %llu bytes in %.3f ms = %.2f TB/s\n", (unsigned long long)bytes, ms, bytes / ms / 1e9);
    }
    return 0;
}
