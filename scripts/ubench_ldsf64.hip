// Rate of ds_add_f64 / ds_add_u32 / ds_read_b64 per CU: W waves per workgroup (one workgroup per CU),
// every lane its own address (conflict free), no waits inside the loop.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/ubench_ldsf64.hip -o scripts/build/ubench_ldsf64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int ITER = 4000;
template <int MODE>
__global__ void k(unsigned long long *out)
{
    extern __shared__ double lds[];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 0.0;
    __syncthreads();
    const unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) double *)lds + 8 * (threadIdx.x & 63) + 512 * (threadIdx.x >> 6);
    double v = 1.0;
    unsigned one = 1;
    long long t0 = clock64();
    for (int i = 0; i < ITER; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (MODE == 0) asm volatile("ds_add_f64 %0, %1" :: "v"(a), "v"(v) : "memory");
            else if (MODE == 1) asm volatile("ds_add_u32 %0, %1" :: "v"(a), "v"(one) : "memory");
            else { double r; asm volatile("ds_read_b64 %0, %1" : "=v"(r) : "v"(a) : "memory"); }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = (unsigned long long)(t1 - t0);
}
int main()
{
    unsigned long long *d, h[256];
    CK(hipMalloc(&d, 256 * 8));
    const char *names[3] = {"ds_add_f64", "ds_add_u32", "ds_read_b64"};
    for (int mode = 0; mode < 3; mode++)
        for (int waves : {1, 4, 6, 8, 16}) {
            if (mode == 0) k<0><<<256, waves * 64, 32768>>>(d);
            else if (mode == 1) k<1><<<256, waves * 64, 32768>>>(d);
            else k<2><<<256, waves * 64, 32768>>>(d);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
            double avg = 0; for (int i = 0; i < 256; i++) avg += (double)h[i]; avg /= 256;
            printf("%-12s %2d waves per CU: %.2f cycles per wave instruction and CU (s_memtime ticks: x clock ratio)\n", names[mode], waves, avg / (ITER * 8.0 * waves));
        }
    return 0;
}
