// Does LDS take unaligned ds_read_b32 / b64 / b128 and ds_write_b32 on gfx950, and at what cost?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
constexpr int ITER = 1000;
template <int W, int MIS>   /* W: 4, 8, 16 bytes per lane; MIS: byte misalignment */
__global__ void __launch_bounds__(1024) k(unsigned long long *out, uint32_t *vals)
{
    __shared__ __align__(16) uint8_t lds[32768];
    for (int i = threadIdx.x; i < 32768; i += blockDim.x) lds[i] = (uint8_t)(i * 7 + (i >> 8));
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    /* 64 lanes at a stride of 348 bytes (records), + MIS */
    const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)lds + (lane * 348 + MIS) % 24000;
    uint32_t acc = 0;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (W == 4) { uint32_t v; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(a), "i"(j * 16)); asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v)); acc += v; }
            if (W == 8) { u32x2 v; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(a), "i"(j * 16)); asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v)); acc += v.x + v.y; }
            if (W == 16) { u32x4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "i"(j * 16)); asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v)); acc += v.x + v.y + v.z + v.w; }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (lane == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    if (blockIdx.x == 0 && threadIdx.x < 64) vals[threadIdx.x] = acc;
}
template <int MIS>
__global__ void k_write(uint32_t *vals)
{
    __shared__ __align__(16) uint8_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)lds + lane * 8 + MIS;
    asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" :: "v"(a), "v"(0x44332211u + lane) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) vals[i] = ((uint32_t *)lds)[i];
}
template <int W, int MIS> void run(unsigned long long *d_out, uint32_t *d_vals)
{
    std::vector<unsigned long long> h(256 * 16);
    std::vector<uint32_t> hv(64);
    hipLaunchKernelGGL((k<W, MIS>), dim3(256), dim3(1024), 0, 0, d_out, d_vals);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), d_out, 256 * 16 * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hv.data(), d_vals, 256, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    /* expected value for lane 0..63 */
    int bad = 0;
    for (int lane = 0; lane < 64; lane++) {
        uint32_t acc = 0;
        for (int j = 0; j < 8; j++)
            for (int w = 0; w < W / 4; w++) {
                uint32_t v = 0;
                for (int b = 0; b < 4; b++) { int i = (lane * 348 + MIS) % 24000 + j * 16 + w * 4 + b; v |= (uint32_t)(uint8_t)(i * 7 + (i >> 8)) << (8 * b); }
                acc += v;
            }
        if (acc * ITER != hv[lane]) bad++;
    }
    printf("ds_read_b%-3d misaligned by %d: %6.2f cycles per wave instruction and CU (16 waves), values %s\n", W * 8, MIS,
           (double)h[h.size() / 2] / (ITER * 8.0 * 16), bad ? "WRONG" : "ok");
}
int main()
{
    unsigned long long *d_out; uint32_t *d_vals;
    CK(hipMalloc(&d_out, 256 * 16 * 8)); CK(hipMalloc(&d_vals, 4096));
    run<4, 0>(d_out, d_vals); run<4, 1>(d_out, d_vals); run<4, 2>(d_out, d_vals); run<4, 3>(d_out, d_vals);
    run<8, 0>(d_out, d_vals); run<8, 1>(d_out, d_vals); run<8, 4>(d_out, d_vals);
    run<16, 0>(d_out, d_vals); run<16, 1>(d_out, d_vals); run<16, 4>(d_out, d_vals); run<16, 8>(d_out, d_vals);
    std::vector<uint32_t> hv(1024);
    hipLaunchKernelGGL((k_write<1>), dim3(1), dim3(64), 0, 0, d_vals);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hv.data(), d_vals, 4096, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int lane = 0; lane < 64; lane++) {
        uint32_t v = 0x44332211u + lane;
        uint32_t w0 = hv[lane * 2], w1 = hv[lane * 2 + 1];
        if (w0 != (v << 8) || w1 != (v >> 24)) bad++;
    }
    printf("ds_write_b32 misaligned by 1: %s (lane 0 words %08x %08x)\n", bad ? "WRONG" : "ok", hv[0], hv[1]);
    return 0;
}
