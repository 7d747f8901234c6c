// Do VALU, SALU and LDS instructions of different waves of one SIMD issue side by side?  Loops of 16
// independent instructions in several mixes, 1..4 waves per SIMD; cycles per instruction and SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int ITER = 2000;
enum { M_VALU16, M_V8_NOP8, M_V8_SALU8, M_V12_LDS4, M_V8_NOP4_LDS4, M_V8_WAIT8, M_V12_LDS4_WAIT, M_N };
const char *names[M_N] = {"16 v_perm", "8 v_perm + 8 s_nop", "8 v_perm + 8 s_add", "12 v_perm + 4 ds_read_b32", "8 v_perm + 4 s_nop + 4 ds_read_b32",
    "8 v_perm + 8 s_waitcnt(no-op)", "12 v_perm + 4 (ds_read + wait)"};
#define VP(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r[(i) & 15]) : "v"(one), "v"(sel));
#define NOP asm volatile("s_nop 0");
#define SADD asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc) :: "scc");
#define LDSR(i) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(q[(i) & 3]) : "v"(a32), "i"(((i) & 15) * 256));
#define WAITN asm volatile("s_waitcnt lgkmcnt(15)");
#define WAIT0 asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]));
template <int M> __global__ void __launch_bounds__(1024) k(unsigned long long *out, uint32_t seed)
{
    __shared__ uint32_t lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i;
    __syncthreads();
    uint32_t r[16], q[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 16; i++) r[i] = seed + i * 77 + threadIdx.x;
    uint32_t one = 1, sel = 0x07060504u - (threadIdx.x & 3), sacc = 0;
    const uint32_t a32 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)lds + 4 * threadIdx.x % 16384;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < ITER; it++) {
        if (M == M_VALU16) { VP(0) VP(1) VP(2) VP(3) VP(4) VP(5) VP(6) VP(7) VP(8) VP(9) VP(10) VP(11) VP(12) VP(13) VP(14) VP(15) }
        if (M == M_V8_NOP8) { VP(0) NOP VP(1) NOP VP(2) NOP VP(3) NOP VP(4) NOP VP(5) NOP VP(6) NOP VP(7) NOP }
        if (M == M_V8_SALU8) { VP(0) SADD VP(1) SADD VP(2) SADD VP(3) SADD VP(4) SADD VP(5) SADD VP(6) SADD VP(7) SADD }
        if (M == M_V12_LDS4) { VP(0) VP(1) VP(2) LDSR(0) VP(3) VP(4) VP(5) LDSR(1) VP(6) VP(7) VP(8) LDSR(2) VP(9) VP(10) VP(11) LDSR(3) WAIT0 }
        if (M == M_V8_NOP4_LDS4) { VP(0) NOP VP(1) LDSR(0) VP(2) NOP VP(3) LDSR(1) VP(4) NOP VP(5) LDSR(2) VP(6) NOP VP(7) LDSR(3) WAIT0 }
        if (M == M_V8_WAIT8) { VP(0) WAITN VP(1) WAITN VP(2) WAITN VP(3) WAITN VP(4) WAITN VP(5) WAITN VP(6) WAITN VP(7) WAITN }
        if (M == M_V12_LDS4_WAIT) { VP(0) VP(1) VP(2) LDSR(0) WAIT0 VP(3) VP(4) VP(5) LDSR(1) WAIT0 VP(6) VP(7) VP(8) LDSR(2) WAIT0 VP(9) VP(10) VP(11) LDSR(3) WAIT0 }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    uint32_t x = sacc ^ q[0] ^ q[1] ^ q[2] ^ q[3];
#pragma unroll
    for (int i = 0; i < 16; i++) x ^= r[i];
    if (x == 0x12345u) out[1 << 20] = x;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int M> void run(unsigned long long *d_out, std::vector<unsigned long long> &h)
{
    printf("%-40s", names[M]);
    for (int wps : {1, 2, 3, 4}) {
        const int threads = 256 * wps;
        CK(hipMemset(d_out, 0, 256 * 16 * 8));
        hipLaunchKernelGGL(k<M>, dim3(256), dim3(threads), 0, 0, d_out, 12345u);
        hipLaunchKernelGGL(k<M>, dim3(256), dim3(threads), 0, 0, d_out, 12345u);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), d_out, 256 * 16 * 8, hipMemcpyDeviceToHost));
        std::vector<unsigned long long> v;
        for (int b = 0; b < 256; b++) for (int w = 0; w < 4 * wps; w++) v.push_back(h[b * 16 + w]);
        std::sort(v.begin(), v.end());
        printf("  w%d: %5.2f", wps, (double)v[v.size() / 2] / ((double)ITER * 16 * wps));
    }
    printf("   cycles per instruction (of 16 per iteration) and SIMD\n");
}
int main()
{
    unsigned long long *d_out;
    CK(hipMalloc(&d_out, ((1 << 20) + 16) * 8));
    std::vector<unsigned long long> h(256 * 16);
    run<M_VALU16>(d_out, h); run<M_V8_NOP8>(d_out, h); run<M_V8_SALU8>(d_out, h); run<M_V12_LDS4>(d_out, h);
    run<M_V8_NOP4_LDS4>(d_out, h); run<M_V8_WAIT8>(d_out, h); run<M_V12_LDS4_WAIT>(d_out, h);
    return 0;
}
