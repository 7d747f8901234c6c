// Issue-rate microbenchmark for gfx950: how many cycles a SIMD spends per wave instruction of
// the kinds the fused pass is made of, at 1 / 2 / 4 / 8 waves per SIMD.  Build + run:
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/ubench_issue scripts/ubench_issue.hip && gpurun_out/ubench_issue
// Every test is a loop of ITER iterations over a block of 16 independent instructions
// (or the dependent chain named in the test); cycles come from s_memtime inside the kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int ITER = 2000;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define REP16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)

enum { T_ADD, T_SDWA, T_PERM, T_F64, T_MAD24, T_LSHLADD, T_XAD, T_BFE, T_MUL32,
       T_DSR32, T_DSR8, T_DSR64, T_DSR16, T_DSADD, T_DSADD_SAME, T_DSW32, T_DSW128, T_DSR128,
       T_MIX_VALU_DSR, T_MIX_VALU_DSADD, T_CMP_BCNT, T_DEP_LDS16, T_DEP_F64, T_SALU, T_MIX3, T_N };
const char *names[T_N] = {"v_add_u32", "v_or_b32_sdwa", "v_perm_b32", "v_add_f64", "v_mad_u32_u24", "v_lshl_add_u32", "v_xad_u32", "v_bfe_u32", "v_mul_lo_u32",
       "ds_read_b32", "ds_read_u8", "ds_read_b64", "ds_read_u16", "ds_add_u32 (64 addrs)", "ds_add_u32 (lane pairs same addr)", "ds_write_b32", "ds_write_b128", "ds_read_b128",
       "8 v_add + 8 ds_read_b32", "8 v_add + 8 ds_add_u32", "8 (v_cmp_sdwa + s_bcnt1)", "dependent or_sdwa+ds_read_u16 chain", "dependent v_add_f64 chain", "s_add_u32", "8 v_add + 4 ds_read + 4 ds_add"};

template <int T>
__global__ void __launch_bounds__(1024) k(unsigned long long *out, uint32_t seed)
{
    __shared__ uint32_t lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (i * 16) & 0x3ff0; /* u16 chain: next address */
    __syncthreads();
    uint32_t r[16];
#pragma unroll
    for (int i = 0; i < 16; i++) r[i] = seed + i * 77 + threadIdx.x;
    double d[8];
#pragma unroll
    for (int i = 0; i < 8; i++) d[i] = 1.0 + i + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t a32 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)lds + 4 * threadIdx.x % 16384;
    uint32_t a8 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)lds + threadIdx.x % 16384;
    uint32_t a64 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)lds + 8 * threadIdx.x % 16384;
    uint32_t a128 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)lds + 16 * threadIdx.x % 16384;
    uint32_t asame = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)lds + 4 * (threadIdx.x >> 1) % 16384;
    uint32_t st = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)lds + ((lane * 16) & 0x3ff0);
    uint32_t one = 1, c3 = 3, sel = 0x07060504u - (lane & 3);
    unsigned long long sacc = 0;
    unsigned long long t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < ITER; it++) {
        if (T == T_ADD) {
#define S(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(one));
            REP16(S)
#undef S
        } else if (T == T_SDWA) {
#define S(i) asm volatile("v_or_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(r[i]) : "v"(one));
            REP16(S)
#undef S
        } else if (T == T_PERM) {
#define S(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(one), "v"(sel));
            REP16(S)
#undef S
        } else if (T == T_F64) {
#define S(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i & 7]) : "v"(d[(i + 1) & 7]));
            REP16(S)
#undef S
        } else if (T == T_MAD24) {
#define S(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r[i]) : "v"(c3), "v"(one));
            REP16(S)
#undef S
        } else if (T == T_LSHLADD) {
#define S(i) asm volatile("v_lshl_add_u32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(one), "v"(c3));
            REP16(S)
#undef S
        } else if (T == T_XAD) {
#define S(i) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(c3), "v"(one));
            REP16(S)
#undef S
        } else if (T == T_BFE) {
#define S(i) asm volatile("v_bfe_u32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(one), "v"(c3));
            REP16(S)
#undef S
        } else if (T == T_MUL32) {
#define S(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[i]) : "v"(c3));
            REP16(S)
#undef S
        } else if (T == T_DSR32) {
#define S(i) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r[i]) : "v"(a32), "i"(i * 256));
            REP16(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (T == T_DSR8) {
#define S(i) asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(r[i]) : "v"(a8), "i"(i * 64));
            REP16(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (T == T_DSR16) {
#define S(i) asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(r[i]) : "v"(a32), "i"(i * 64));
            REP16(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (T == T_DSR64) {
            unsigned long long q[8];
#define S(i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(q[i & 7]) : "v"(a64), "i"(i * 512));
            REP16(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]) :: "memory");
            r[0] ^= (uint32_t)(q[0] ^ q[1] ^ q[2] ^ q[3] ^ q[4] ^ q[5] ^ q[6] ^ q[7]);
        } else if (T == T_DSR128) {
            u32x4 q0, q1, q2, q3;
#define S(i) if ((i & 3) == 0) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q0) : "v"(a128), "i"((i & 3) * 1024)); \
             else if ((i & 3) == 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q1) : "v"(a128), "i"((i & 3) * 1024)); \
             else if ((i & 3) == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q2) : "v"(a128), "i"((i & 3) * 1024)); \
             else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q3) : "v"(a128), "i"((i & 3) * 1024));
            REP16(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) :: "memory");
            r[0] ^= q0.x ^ q1.x ^ q2.x ^ q3.x;
        } else if (T == T_DSADD) {
#define S(i) asm volatile("ds_add_u32 %0, %1 offset:%2" :: "v"(a32), "v"(one), "i"(i * 256) : "memory");
            REP16(S)
#undef S
        } else if (T == T_DSADD_SAME) {
#define S(i) asm volatile("ds_add_u32 %0, %1 offset:%2" :: "v"(asame), "v"(one), "i"(i * 256) : "memory");
            REP16(S)
#undef S
        } else if (T == T_DSW32) {
#define S(i) asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(a32), "v"(r[i]), "i"(i * 256) : "memory");
            REP16(S)
#undef S
        } else if (T == T_DSW128) {
            u32x4 q = {r[0], r[1], r[2], r[3]};
#define S(i) asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(a128), "v"(q), "i"((i & 3) * 1024) : "memory");
            REP16(S)
#undef S
        } else if (T == T_MIX_VALU_DSR) {
#define S(i) if (i & 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(one)); else asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r[i]) : "v"(a32), "i"(i * 256));
            REP16(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (T == T_MIX_VALU_DSADD) {
#define S(i) if (i & 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(one)); else asm volatile("ds_add_u32 %0, %1 offset:%2" :: "v"(a32), "v"(one), "i"(i * 256) : "memory");
            REP16(S)
#undef S
        } else if (T == T_MIX3) {
#define S(i) if (i & 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(one)); else if (i & 2) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r[i]) : "v"(a32), "i"(i * 256)); else asm volatile("ds_add_u32 %0, %1 offset:%2" :: "v"(a32), "v"(one), "i"(i * 256) : "memory");
            REP16(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (T == T_CMP_BCNT) {
            unsigned long long m; uint32_t cnt;
#define S(i) if (i < 8) { asm volatile("v_cmp_eq_u32_sdwa %0, %1, %2 src0_sel:BYTE_1 src1_sel:DWORD" : "=s"(m) : "v"(r[i]), "v"(one)); \
                          asm volatile("s_bcnt1_i32_b64 %0, %1" : "=s"(cnt) : "s"(m) : "scc"); sacc += cnt; }
            REP16(S)
#undef S
        } else if (T == T_DEP_LDS16) {
#define S(i) asm volatile("v_or_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\tds_read_u16 %0, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(st) : "v"(one) : "memory");
            REP16(S)
#undef S
        } else if (T == T_DEP_F64) {
#define S(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[0]) : "v"(d[1]));
            REP16(S)
#undef S
        } else if (T == T_SALU) {
            uint32_t s = (uint32_t)sacc;
#define S(i) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s) :: "scc");
            REP16(S)
#undef S
            sacc = s;
        }
    }
    unsigned long long t1;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    uint32_t x = st ^ (uint32_t)sacc;
#pragma unroll
    for (int i = 0; i < 16; i++) x ^= r[i];
#pragma unroll
    for (int i = 0; i < 8; i++) x ^= (uint32_t)__double_as_longlong(d[i]);
    if (x == 0x12345u) out[1 << 20] = x;
    if (lane == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int T> void run(unsigned long long *d_out, std::vector<unsigned long long> &h)
{
    const int per_iter = (T == T_CMP_BCNT) ? 8 : 16;
    printf("%-40s", names[T]);
    for (int wps : {1, 2, 4}) {
        const int threads = 256 * wps; /* one workgroup per CU, wps waves per SIMD */
        CK(hipMemset(d_out, 0, 256 * 8 * 16 * 8));
        hipLaunchKernelGGL(k<T>, dim3(256), dim3(threads), 0, 0, d_out, 12345u);
        hipLaunchKernelGGL(k<T>, dim3(256), dim3(threads), 0, 0, d_out, 12345u);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), d_out, 256 * 16 * 8, hipMemcpyDeviceToHost));
        std::vector<unsigned long long> v;
        for (int b = 0; b < 256; b++) for (int w = 0; w < 4 * wps; w++) v.push_back(h[b * 16 + w]);
        std::sort(v.begin(), v.end());
        const double med = (double)v[v.size() / 2];
        /* cycles the SIMD spends per wave instruction = elapsed / (instructions per wave * waves per SIMD) */
        printf("  wps=%d: %6.2f cyc/inst/SIMD (wave: %7.1f)", wps, med / ((double)ITER * per_iter * wps), med / ((double)ITER * per_iter));
    }
    printf("\n");
}

int main()
{
    unsigned long long *d_out;
    CK(hipMalloc(&d_out, ((1 << 20) + 16) * 8));
    std::vector<unsigned long long> h(256 * 16);
    run<T_ADD>(d_out, h); run<T_SDWA>(d_out, h); run<T_PERM>(d_out, h); run<T_F64>(d_out, h); run<T_MAD24>(d_out, h);
    run<T_LSHLADD>(d_out, h); run<T_XAD>(d_out, h); run<T_BFE>(d_out, h); run<T_MUL32>(d_out, h);
    run<T_DSR32>(d_out, h); run<T_DSR8>(d_out, h); run<T_DSR64>(d_out, h); run<T_DSR16>(d_out, h); run<T_DSR128>(d_out, h);
    run<T_DSADD>(d_out, h); run<T_DSADD_SAME>(d_out, h); run<T_DSW32>(d_out, h); run<T_DSW128>(d_out, h);
    run<T_MIX_VALU_DSR>(d_out, h); run<T_MIX_VALU_DSADD>(d_out, h); run<T_MIX3>(d_out, h); run<T_CMP_BCNT>(d_out, h);
    run<T_DEP_LDS16>(d_out, h); run<T_DEP_F64>(d_out, h); run<T_SALU>(d_out, h);
    return 0;
}
