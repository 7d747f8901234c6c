// round 5: does gfx950 have the quad sliding-window SAD instructions, what do they compute, and what do they cost to issue?
// (the overlap scan of InsertSizeMetrics inside k_span<PAIR=2> slides a 4-byte compare one base at a time: 18 lane operations
// per position; v_qsad_pk_u16_u8 compares 4 positions against a 4-byte reference in one instruction and accumulates)
//   hipcc --offload-arch=gfx950 -O3 -o scripts/build/ubench_qsad scripts/ubench_qsad.hip && scripts/build/ubench_qsad
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int ITER = 2000;
#define REP16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
enum { T_ADD, T_QSAD, T_MQSAD_PK, T_MQSAD_U32, T_SAD, T_ALIGNBYTE, T_PKMIN, T_CMP, T_N };
const char *names[T_N] = {"v_add_u32", "v_qsad_pk_u16_u8", "v_mqsad_pk_u16_u8", "v_mqsad_u32_u8", "v_sad_u8", "v_alignbyte_b32", "v_pk_min_u16", "v_cmp_eq_u32 (e64 -> sgpr)"};

template <int T>
__global__ void __launch_bounds__(1024) k(unsigned long long *out, uint32_t seed)
{
    uint32_t r[16];
    unsigned long long w[16];
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 x[8];
#pragma unroll
    for (int i = 0; i < 16; i++) { r[i] = seed + i * 77 + threadIdx.x; w[i] = ((unsigned long long)r[i] << 32) | (r[i] * 31u); }
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = u32x4{r[i], r[i] + 1, r[i] + 2, r[i] + 3};
    uint32_t one = 0x01020304u + threadIdx.x;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < ITER; it++) {
        if (T == T_ADD) {
#define S(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(one));
            REP16(S)
#undef S
        } else if (T == T_QSAD) {
#define S(i) asm volatile("v_qsad_pk_u16_u8 %0, %0, %1, %0" : "+v"(w[i]) : "v"(one));
            REP16(S)
#undef S
        } else if (T == T_MQSAD_PK) {
#define S(i) asm volatile("v_mqsad_pk_u16_u8 %0, %0, %1, %0" : "+v"(w[i]) : "v"(one));
            REP16(S)
#undef S
        } else if (T == T_MQSAD_U32) {
#define S(i) asm volatile("v_mqsad_u32_u8 %0, %1, %2, %0" : "+v"(x[i & 7]) : "v"(w[i]), "v"(one));
            REP16(S)
#undef S
        } else if (T == T_SAD) {
#define S(i) asm volatile("v_sad_u8 %0, %0, %1, %0" : "+v"(r[i]) : "v"(one));
            REP16(S)
#undef S
        } else if (T == T_ALIGNBYTE) {
#define S(i) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(r[i]) : "v"(one));
            REP16(S)
#undef S
        } else if (T == T_PKMIN) {
#define S(i) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(r[i]) : "v"(one));
            REP16(S)
#undef S
        } else if (T == T_CMP) {
#define S(i) { unsigned long long m; asm volatile("v_cmp_eq_u32_e64 %0, %1, %2" : "=s"(m) : "v"(r[i]), "v"(one)); asm volatile("" :: "s"(m)); }
            REP16(S)
#undef S
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    unsigned long long sink = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) sink += r[i] + w[i];
#pragma unroll
    for (int i = 0; i < 8; i++) sink += x[i].x + x[i].y + x[i].z + x[i].w;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = sink; }
    else if (sink == 0x1234567887654321ull) out[3] = sink;
}

__global__ void k_sem(const unsigned long long *in, unsigned long long *out)
{
    // in[0] = 8 data bytes, in[1] = reference (low dword), in[2] = accumulator
    unsigned long long d = in[0], acc = in[2], q, mq;
    uint32_t ref = (uint32_t)in[1];
    asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %3" : "=&v"(q) : "v"(d), "v"(ref), "v"(acc));
    asm volatile("v_mqsad_pk_u16_u8 %0, %1, %2, %3" : "=&v"(mq) : "v"(d), "v"(ref), "v"(acc));
    out[0] = q; out[1] = mq;
}

template <int T> void run(unsigned long long *d_out, int waves_per_simd)
{
    const int threads = 256 * waves_per_simd > 1024 ? 1024 : 256 * waves_per_simd;
    k<T><<<1, threads>>>(d_out, 12345);   // one workgroup on one CU: waves_per_simd waves on each of its 4 SIMDs
    CK(hipDeviceSynchronize());
    unsigned long long h[2];
    CK(hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost));
    printf("%-28s %d waves/SIMD: %.2f cycles per wave instruction and SIMD\n", names[T], threads / 256, (double)h[0] / (ITER * 16.0 * (threads / 256)));
}

int main()
{
    unsigned long long *d_out, *d_in;
    CK(hipMalloc(&d_out, 4096)); CK(hipMalloc(&d_in, 64));
    // semantics: data bytes 10 20 30 40 50 60 70 80, reference 30 40 50 60 (the window at byte offset 2 matches), accumulator 1 2 3 4
    unsigned char data[8] = {10, 20, 30, 40, 50, 60, 70, 80}, ref[8] = {30, 40, 50, 60, 0, 0, 0, 0};
    unsigned short acc[4] = {1, 2, 3, 4};
    unsigned long long in[3];
    memcpy(&in[0], data, 8); memcpy(&in[1], ref, 8); memcpy(&in[2], acc, 8);
    CK(hipMemcpy(d_in, in, 24, hipMemcpyHostToDevice));
    k_sem<<<1, 64>>>(d_in, d_out);
    CK(hipDeviceSynchronize());
    unsigned long long h[2];
    CK(hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost));
    unsigned short q[4], mq[4];
    memcpy(q, &h[0], 8); memcpy(mq, &h[1], 8);
    printf("v_qsad_pk_u16_u8 : %u %u %u %u   (expected, SAD of the window at byte offset k against the reference, + acc[k]: ", q[0], q[1], q[2], q[3]);
    for (int k = 0; k < 4; k++) { int s = acc[k]; for (int b = 0; b < 4; b++) s += abs((int)data[k + b] - (int)ref[b]); printf("%d ", s); }
    printf(")\nv_mqsad_pk_u16_u8: %u %u %u %u   (masked: reference bytes that are 0 do not count)\n", mq[0], mq[1], mq[2], mq[3]);
    for (int wps : {1, 2, 4}) {
        run<T_ADD>(d_out, wps); run<T_QSAD>(d_out, wps); run<T_MQSAD_PK>(d_out, wps); run<T_MQSAD_U32>(d_out, wps);
        run<T_SAD>(d_out, wps); run<T_ALIGNBYTE>(d_out, wps); run<T_PKMIN>(d_out, wps); run<T_CMP>(d_out, wps);
    }
    return 0;
}
