// Is the VALU issue rate of 8-byte encodings bound by instruction fetch?  Same adds in e32 / e64
// encodings, block sizes 4 / 16 / 64 instructions per loop iteration, 1..8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int ITER = 2000;
#define REP4(S, b) S(b + 0) S(b + 1) S(b + 2) S(b + 3)
#define REP16(S, b) REP4(S, b) REP4(S, b + 4) REP4(S, b + 8) REP4(S, b + 12)
#define REP64(S) REP16(S, 0) REP16(S, 16) REP16(S, 32) REP16(S, 48)
enum { E32, E64, E64_3OP, MIX, NOP, F64, MUL, E32_64BLK, E64_64BLK, E64_4BLK, MOV32, SDWA_DIFF, T_N };
const char *names[T_N] = {"v_add_u32 e32 x16", "v_add_u32 e64 x16", "v_add3_u32 x16", "e32/e64 alternating x16", "s_nop 0 x16", "v_add_f64 x16", "v_mul_lo_u32 x16",
    "v_add_u32 e32 x64", "v_add_u32 e64 x64", "v_add_u32 e64 x4", "v_mov_b32 e32 x16", "v_xor_b32 e64 dst!=src x16"};
template <int T> __global__ void __launch_bounds__(1024) k(unsigned long long *out, uint32_t seed)
{
    uint32_t r[16];
#pragma unroll
    for (int i = 0; i < 16; i++) r[i] = seed + i * 77 + threadIdx.x;
    double d[8];
#pragma unroll
    for (int i = 0; i < 8; i++) d[i] = 1.0 + i + threadIdx.x;
    uint32_t one = 1;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < ITER; it++) {
        if (T == E32) {
#define S(i) asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(r[(i) & 15]) : "v"(one));
            REP16(S, 0)
#undef S
        } else if (T == E64) {
#define S(i) asm volatile("v_add_u32_e64 %0, %0, %1" : "+v"(r[(i) & 15]) : "v"(one));
            REP16(S, 0)
#undef S
        } else if (T == E64_3OP) {
#define S(i) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(r[(i) & 15]) : "v"(one));
            REP16(S, 0)
#undef S
        } else if (T == MIX) {
#define S(i) if ((i) & 1) asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(r[(i) & 15]) : "v"(one)); else asm volatile("v_add_u32_e64 %0, %0, %1" : "+v"(r[(i) & 15]) : "v"(one));
            REP16(S, 0)
#undef S
        } else if (T == NOP) {
#define S(i) asm volatile("s_nop 0");
            REP16(S, 0)
#undef S
        } else if (T == F64) {
#define S(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[(i) & 7]) : "v"(d[((i) + 1) & 7]));
            REP16(S, 0)
#undef S
        } else if (T == MUL) {
#define S(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[(i) & 15]) : "v"(one));
            REP16(S, 0)
#undef S
        } else if (T == E32_64BLK) {
#define S(i) asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(r[(i) & 15]) : "v"(one));
            REP64(S)
#undef S
        } else if (T == E64_64BLK) {
#define S(i) asm volatile("v_add_u32_e64 %0, %0, %1" : "+v"(r[(i) & 15]) : "v"(one));
            REP64(S)
#undef S
        } else if (T == E64_4BLK) {
#define S(i) asm volatile("v_add_u32_e64 %0, %0, %1" : "+v"(r[(i) & 15]) : "v"(one));
            REP4(S, 0)
#undef S
        } else if (T == MOV32) {
#define S(i) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(r[(i) & 15]) : "v"(one));
            REP16(S, 0)
#undef S
        } else if (T == SDWA_DIFF) {
#define S(i) asm volatile("v_xor_b32_e64 %0, %1, %2" : "=v"(r[(i) & 15]) : "v"(r[((i) + 5) & 15]), "v"(one));
            REP16(S, 0)
#undef S
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) x ^= r[i];
#pragma unroll
    for (int i = 0; i < 8; i++) x ^= (uint32_t)__double_as_longlong(d[i]);
    if (x == 0x12345u) out[1 << 20] = x;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int T> void run(unsigned long long *d_out, std::vector<unsigned long long> &h)
{
    const int per_iter = (T == E32_64BLK || T == E64_64BLK) ? 64 : (T == E64_4BLK ? 4 : 16);
    printf("%-30s", names[T]);
    for (int wps : {1, 2, 3, 4, 8}) {
        const int blocks_per_cu = wps == 8 ? 2 : 1;
        const int threads = 256 * wps / blocks_per_cu;
        CK(hipMemset(d_out, 0, 512 * 16 * 8));
        hipLaunchKernelGGL(k<T>, dim3(256 * blocks_per_cu), dim3(threads), 0, 0, d_out, 12345u);
        hipLaunchKernelGGL(k<T>, dim3(256 * blocks_per_cu), dim3(threads), 0, 0, d_out, 12345u);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), d_out, 512 * 16 * 8, hipMemcpyDeviceToHost));
        std::vector<unsigned long long> v;
        for (int b = 0; b < 256 * blocks_per_cu; b++) for (int w = 0; w < threads / 64; w++) v.push_back(h[b * 16 + w]);
        std::sort(v.begin(), v.end());
        const double med = (double)v[v.size() / 2];
        printf("  w%d: %5.2f", wps, med / ((double)ITER * per_iter * wps));
    }
    printf("   (cycles per instruction and SIMD)\n");
}
int main()
{
    unsigned long long *d_out;
    CK(hipMalloc(&d_out, ((1 << 20) + 16) * 8));
    std::vector<unsigned long long> h(512 * 16);
    run<E32>(d_out, h); run<E64>(d_out, h); run<E64_3OP>(d_out, h); run<MIX>(d_out, h); run<NOP>(d_out, h); run<F64>(d_out, h); run<MUL>(d_out, h);
    run<E32_64BLK>(d_out, h); run<E64_64BLK>(d_out, h); run<E64_4BLK>(d_out, h); run<MOV32>(d_out, h); run<SDWA_DIFF>(d_out, h);
    return 0;
}
