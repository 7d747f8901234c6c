// What ds_read_b64_tr_b8 delivers: every lane hands in the address of an 8-byte chunk; the
// program prints, per lane, the LDS byte addresses its 8 result bytes came from.
// Build: hipcc --offload-arch=gfx950 -O2 scripts/ubench_tr8.hip -o scripts/build/ubench_tr8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const uint32_t *addr, uint32_t *out, int mode)
{
    __shared__ __align__(16) uint8_t lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = mode ? (uint8_t)(i >> 8) : (uint8_t)i;
    __syncthreads();
    uint32_t a = (uint32_t)(uintptr_t)lds + addr[threadIdx.x];
    u32x2 v;
    asm volatile("ds_read_b64_tr_b8 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    out[2 * threadIdx.x] = v.x;
    out[2 * threadIdx.x + 1] = v.y;
}
int main()
{
    uint32_t *da, *dout;
    hipMalloc(&da, 256); hipMalloc(&dout, 512);
    for (int pattern = 0; pattern < 2; pattern++) {
        std::vector<uint32_t> a(64), lo(128), hi(128);
        for (int l = 0; l < 64; l++) a[l] = pattern == 0 ? 8 * l : 24 * l; // chunks in lane order / rows 24 bytes apart
        hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice);
        k<<<1, 64>>>(da, dout, 0); hipMemcpy(lo.data(), dout, 512, hipMemcpyDeviceToHost);
        k<<<1, 64>>>(da, dout, 1); hipMemcpy(hi.data(), dout, 512, hipMemcpyDeviceToHost);
        printf("pattern %d (lane l hands in byte address %d l)\n", pattern, pattern == 0 ? 8 : 24);
        for (int l = 0; l < 64; l++) {
            printf("lane %2d:", l);
            for (int b = 0; b < 8; b++) {
                int src = ((lo[2 * l + b / 4] >> (8 * (b % 4))) & 0xFF) | (((hi[2 * l + b / 4] >> (8 * (b % 4))) & 0xFF) << 8);
                int step = pattern == 0 ? 8 : 24;
                printf("  l%-2d+%d", src / step, src % step);
            }
            printf("\n");
        }
    }
    return 0;
}
