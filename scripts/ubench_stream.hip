// What memory delivers to the access patterns of this library's kernels (MI355X): one read of an 8 GiB buffer as
//   flat      grid-stride, 16 bytes per lane, registers (ubench_flat)
//   chunks    a workgroup walks chunks of CH bytes of its own (the way k_span's workgroups take their spans), registers
//   dma       the same chunks by LDS-DMA (global_load_lds_dwordx4: 1 KB per wave instruction, two slots per wave)
//   dma+5     the same, every wave's source 5 bytes off a 16-byte boundary (records lie anywhere)
//   rows      chunks, but only 300 of every 341 bytes (the sequence and quality lines of a record, not its name)
//   meta      40-byte structs, a lane per struct, three loads (16 + 16 + 8 bytes) as the compiler does for `sq_meta m = metas[i]`
//   dma x2    every KB by LDS-DMA twice, the second time GAP steps later (does the L2 keep what an LDS-DMA brought?)
//   gather    rows of 208 bytes (13 pieces: a read of 100 bases, its sequence and its qualities) from random records by
//             LDS-DMA, four rows per wave instruction: what k_span over length-sorted rows asks memory for
// with 4 / 8 / 16 waves per CU.   hipcc --offload-arch=gfx950 -O3 -o scripts/build/ubench_stream scripts/ubench_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void dma16(const uint8_t *g, uint32_t lds_dst)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_dst) : "memory");
}

template <int MODE>   /* 0 flat, 1 chunks, 2 dma, 3 dma + 5, 4 rows, 5 meta, 6 dma twice (1 step apart), 7 dma twice (4 steps apart) */
__global__ void __launch_bounds__(1024) k_stream(const uint8_t *buf, uint64_t bytes, uint32_t chunk, unsigned long long *sink)
{
    extern __shared__ __align__(16) uint8_t smem[];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, W = blockDim.x >> 6;
    unsigned long long acc = 0;
    if (MODE == 0) {
        const uint4 *p = (const uint4 *)buf;
        for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < bytes / 16; i += (uint64_t)gridDim.x * blockDim.x) {
            const uint4 v = p[i];
            acc += v.x + v.y + v.z + v.w;
        }
    } else if (MODE == 5) {
        for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < bytes / 40; i += (uint64_t)gridDim.x * blockDim.x) {
            uint4 a, b; uint2 c;
            __builtin_memcpy(&a, buf + 40 * i, 16);
            __builtin_memcpy(&b, buf + 40 * i + 16, 16);
            __builtin_memcpy(&c, buf + 40 * i + 32, 8);
            acc += a.x + b.y + c.x;
        }
    } else {
        /* wave w of workgroup b: chunks (b W + w) + k (grid W) of `chunk` bytes, 1 KB per step */
        const uint64_t n_chunks = bytes / chunk;
        const uint32_t slot = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)smem + wave * 2048;
        for (uint64_t c = (uint64_t)blockIdx.x * W + wave; c < n_chunks; c += (uint64_t)gridDim.x * W) {
            const uint8_t *src = buf + c * chunk + (MODE == 3 ? 5 : 0);
            const uint32_t steps = (chunk - 16) / 1024;
            if (MODE == 1 || MODE == 4) {
                for (uint32_t s = 0; s < steps; s++) {
                    uint64_t off = (uint64_t)s * 1024 + 16 * lane;
                    if (MODE == 4) { off = off / 300 * 341 + off % 300 + 37; if (off + 16 > chunk) continue; }
                    uint4 v;
                    __builtin_memcpy(&v, src + off, 16);
                    acc += v.x + v.y + v.z + v.w;
                }
            } else if (MODE == 8) {
                /* `steps` x 4 rows per chunk; row number -> a record somewhere in the buffer */
                const uint64_t n_rec = bytes / 341;
                const uint32_t row_in = lane / 13, piece = lane % 13;
                for (uint32_t s = 0; s < steps; s++) {
                    const uint64_t rowno = (c * steps + s) * 4 + row_in;
                    const uint64_t rec = (rowno * 0x9E3779B97F4A7C15ULL >> 20) % n_rec;
                    if (lane < 52) dma16(buf + rec * 341 + 37 + 16 * piece, __builtin_amdgcn_readfirstlane(slot + 1024 * (s & 1)));
                    asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                acc += *(const uint32_t *)(smem + wave * 2048 + 4 * lane);
            } else if (MODE >= 6) {
                const uint32_t gap = MODE == 6 ? 1 : 4;
                for (uint32_t s = 0; s < steps + gap; s++) {
                    if (s < steps) dma16(src + (uint64_t)s * 1024 + 16 * lane, __builtin_amdgcn_readfirstlane(slot));
                    if (s >= gap) dma16(src + (uint64_t)(s - gap) * 1024 + 16 * lane, __builtin_amdgcn_readfirstlane(slot + 1024));
                    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                acc += *(const uint32_t *)(smem + wave * 2048 + 4 * lane);
            } else {
                dma16(src + 16 * lane, __builtin_amdgcn_readfirstlane(slot));
                for (uint32_t s = 0; s < steps; s++) {
                    if (s + 1 < steps) dma16(src + (uint64_t)(s + 1) * 1024 + 16 * lane, __builtin_amdgcn_readfirstlane(slot + 1024 * ((s + 1) & 1)));
                    if (s + 1 < steps) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    acc += *(const uint32_t *)(smem + wave * 2048 + 1024 * (s & 1) + 4 * lane);
                }
            }
        }
    }
    if (acc == 0x123456789ULL) *sink = acc;
}

__global__ void k_fill(uint32_t *buf, uint64_t n)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) buf[i] = (uint32_t)(i * 2654435761u);
}

template <int MODE> void run(const char *name, const uint8_t *buf, uint64_t bytes, uint32_t chunk, int waves, unsigned long long *sink)
{
    CK(hipFuncSetAttribute((const void *)k_stream<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k_stream<MODE>), dim3(256 * (MODE == 0 || MODE == 5 ? 4 : 1)), dim3(waves * 64), 16 * 2048, 0, buf, bytes, chunk, sink);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    /* gather: every KB step asks for 4 rows of 208 bytes = 832 bytes (they touch 2.6 lines of 128 bytes each: 1333 bytes) */
    const double useful = MODE == 4 ? bytes * (300.0 / 341.0) : MODE == 8 ? bytes * (832.0 / 1024.0) : (double)bytes;
    printf("%-8s chunk %6u B, %2d waves per workgroup: %7.3f ms = %5.2f TB/s of the bytes asked for\n", name, chunk, waves, best, useful / best / 1e9);
}

int main()
{
    const uint64_t bytes = 8ULL << 30;
    uint8_t *buf; unsigned long long *sink;
    CK(hipMalloc(&buf, bytes + 4096)); CK(hipMalloc(&sink, 8));
    k_fill<<<4096, 256>>>((uint32_t *)buf, bytes / 4);
    CK(hipDeviceSynchronize());
    for (int waves : {4, 8, 16}) {
        run<0>("flat", buf, bytes, 0, waves, sink);
        run<5>("meta", buf, bytes, 0, waves, sink);
        for (uint32_t chunk : {5456u + 16u, 43648u + 16u}) {   /* one span of 16 records / eight */
            run<1>("chunks", buf, bytes, chunk, waves, sink);
            run<2>("dma", buf, bytes, chunk, waves, sink);
            run<3>("dma+5", buf, bytes, chunk, waves, sink);
            run<4>("rows", buf, bytes, chunk, waves, sink);
            run<6>("dma x2/1", buf, bytes, chunk, waves, sink);
            run<7>("dma x2/4", buf, bytes, chunk, waves, sink);
            run<8>("gather", buf, bytes, chunk, waves, sink);
        }
    }
    return 0;
}
