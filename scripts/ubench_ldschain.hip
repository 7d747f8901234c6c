// What does a dependent LDS table walk cost next to other LDS traffic?  k_span's automaton is a
// chain of `v_add_u32_sdwa addr, state, code` + `ds_read_u16` steps (56 per span of 16 reads) and
// runs at ~205 cycles per step in the loaded kernel against 68 alone.  This bench
// separates the candidates: every wave walks a chain of CH dependent reads per round and issues,
// between two chain steps, IND independent LDS operations (reads, or ds_add_u32) and VAL
// independent VALU instructions; waves per CU from 4 to 16.  Reports cycles per chain step (wave
// time / steps) and LDS instructions per cycle and CU.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/build/ubench_ldschain scripts/ubench_ldschain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int IND, int VAL, int KIND>   /* KIND 0: independent ds_read_b32, 1: ds_add_u32, 2: ds_read_b64 */
__global__ void __launch_bounds__(1024) k_chain(uint32_t *out, int rounds, unsigned long long *cycles)
{
    extern __shared__ __align__(16) uint8_t smem[];
    uint16_t *tab = (uint16_t *)smem;                 /* 4096 entries: a random walk over itself (byte offsets) */
    uint32_t *pool = (uint32_t *)(smem + 8192);       /* 8192 dwords of independent traffic */
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) tab[i] = (uint16_t)(((i * 2654435761u) >> 13) & 0x1FFE);
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) pool[i] = i;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t a = (lane * 34 + wave * 6) & 0x1FFE;     /* chain state: byte offset into tab */
    uint32_t pa = 8192 + 4 * ((lane + 64 * wave) & 8191);   /* lanes on consecutive dwords: conflict free */
    uint32_t acc = 0, v0 = lane, v1 = wave, v2 = 3, v3 = 5;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rounds; r++) {
#pragma unroll
        for (int s = 0; s < 4; s++) {
            uint32_t e;
            asm volatile("ds_read_u16 %0, %1" : "=v"(e) : "v"(a) : "memory");
            uint32_t x[IND > 0 ? IND : 1];
#pragma unroll
            for (int k = 0; k < IND; k++) {
                if (KIND == 0) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(x[k]) : "v"(pa), "i"(256 * k) : "memory");
                else if (KIND == 1) asm volatile("ds_add_u32 %0, %1 offset:%2" :: "v"(pa), "v"(v2), "i"(256 * k) : "memory");
                else { uint64_t y; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(y) : "v"(pa), "i"(512 * k) : "memory"); x[k] = (uint32_t)y; }
            }
#pragma unroll
            for (int k = 0; k < VAL; k++) {   /* independent VALU work, four chains */
                if ((k & 3) == 0) v0 = v0 * 3 + v1; else if ((k & 3) == 1) v1 = v1 + (v2 >> 1); else if ((k & 3) == 2) v2 = v2 ^ (v3 << 2); else v3 = v3 + v0;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("" : "+v"(e));
            if (KIND != 1)
#pragma unroll
                for (int k = 0; k < IND; k++) { asm volatile("" : "+v"(x[k])); acc += x[k]; }
            a = e;   /* the next step depends on this read */
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) atomicAdd(cycles, t1 - t0);
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + acc + v0 + v1 + v2 + v3;
}

template <int IND, int VAL, int KIND>
void run(int waves, uint32_t *d_out, unsigned long long *d_cyc)
{
    const int rounds = 20000;
    const size_t lds = 8192 + 32768 + 2048;
    CK(hipFuncSetAttribute((const void *)k_chain<IND, VAL, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9; unsigned long long cyc = 0;
    for (int rep = 0; rep < 2; rep++) {
        CK(hipMemset(d_cyc, 0, 8));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_chain<IND, VAL, KIND>), dim3(256), dim3(waves * 64), lds, 0, d_out, rounds, d_cyc);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) { best = ms; CK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost)); }
    }
    const double steps = 4.0 * rounds, per_wave = (double)cyc / (256.0 * waves);
    const double lds_per_cycle = (double)waves * steps * (1 + IND) / per_wave;
    printf("waves %2d  ind %2d (%s)  valu %2d : %7.1f cycles per chain step, %5.3f LDS instr / cycle / CU, %.3f ms\n", waves, IND,
           KIND == 0 ? "read_b32" : KIND == 1 ? "add_u32 " : "read_b64", VAL, per_wave / steps, lds_per_cycle, best);
}

int main()
{
    uint32_t *d_out; unsigned long long *d_cyc;
    CK(hipMalloc(&d_out, 256 * 1024 * 4)); CK(hipMalloc(&d_cyc, 8));
    for (int waves : {1, 4, 8, 12, 16}) {
        run<0, 0, 0>(waves, d_out, d_cyc);
        run<0, 12, 0>(waves, d_out, d_cyc);
        run<2, 0, 0>(waves, d_out, d_cyc);
        run<4, 0, 0>(waves, d_out, d_cyc);
        run<4, 12, 0>(waves, d_out, d_cyc);
        run<8, 0, 0>(waves, d_out, d_cyc);
        run<4, 0, 1>(waves, d_out, d_cyc);
        run<4, 12, 1>(waves, d_out, d_cyc);
        run<4, 0, 2>(waves, d_out, d_cyc);
        run<4, 24, 2>(waves, d_out, d_cyc);
    }
    return 0;
}
