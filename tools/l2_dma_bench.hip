// Microbenchmark: how fast can a CU pull L2-resident operand tiles into LDS (or VGPRs)?
// Modes: sharing pattern of the source region among blocks, LDS-DMA vs plain loads, waves per block.
// Build: hipcc --offload-arch=gfx950 -O3 tools/l2_dma_bench.hip -o /tmp/l2bench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef __attribute__((address_space(3))) void lds_void_t;

__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_addr) : "memory");
}

// Each block streams `iters` stages of STAGE_KB KiB. region_of_block decides which part of the
// buffer a block reads: share = number of consecutive blocks reading the same region.
template <int NWAVES, bool USE_DMA, int ROW_BYTES>
__global__ __launch_bounds__(NWAVES * 64) void stream_kernel(const char* __restrict__ buf, size_t region_bytes,
                                                            int share, int n_regions, int stage_bytes, int iters,
                                                            float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int region = (blockIdx.x / share) % n_regions;
    const char* base = buf + (size_t)region * region_bytes;
    const unsigned smem_base = (unsigned)(size_t)(lds_void_t*)smem;
    const int instr_per_stage = stage_bytes / 1024;          // 1 KiB per wave-instruction
    const int ipw = instr_per_stage / NWAVES;
    // lane -> (row, chunk) like the conv kernel: ROW_BYTES contiguous per row, rows 4 KiB apart
    const int cpr = ROW_BYTES / 16, rpi = 64 / cpr;
    const int lrow = lane / cpr, lch = lane % cpr;
    float acc = 0.f;
    const int stages = 3;
    for (int it = 0; it < iters; ++it) {
        const int slot = it % stages;
        for (int i = 0; i < ipw; ++i) {
            const int inst = i * NWAVES + wave;
            // source: row (inst*rpi + lrow), rows strided by 4096 B within the region, column window moves with it
            const size_t row = (size_t)(inst * rpi + lrow);
            const size_t off = (row * 4096 + (size_t)((it * ROW_BYTES) % 4096) + lch * 16) % region_bytes;
            if (USE_DMA) {
                dma16(base + off, smem_base + slot * stage_bytes + inst * 1024);
            } else {
                const float4 v = *reinterpret_cast<const float4*>(base + off);
                acc += v.x + v.w;
            }
        }
        if (USE_DMA) {
            if (it >= 2) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(12) : "memory");   // ~2 stages in flight
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 12345.f) sink[0] = acc;
}

template <int NWAVES, bool USE_DMA, int ROW_BYTES>
void run(const char* name, const char* d, size_t total_bytes, size_t region_bytes, int share, int stage_kb, int blocks, float* sink) {
    const int n_regions = (int)(total_bytes / region_bytes);
    const int stage_bytes = stage_kb * 1024;
    const int iters = 64;
    auto k = stream_kernel<NWAVES, USE_DMA, ROW_BYTES>;
    const int lds = USE_DMA ? 3 * stage_bytes : 0;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds > 0 ? lds : 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, dim3(blocks), dim3(NWAVES * 64), lds, 0, d, region_bytes, share, n_regions, stage_bytes, iters, sink);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(k, dim3(blocks), dim3(NWAVES * 64), lds, 0, d, region_bytes, share, n_regions, stage_bytes, iters, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)blocks * iters * stage_bytes * reps;
    printf("%-44s waves %d stage %3d KiB blocks %5d share %3d regions %4d : %7.2f TB/s  (%.1f B/clk/CU @2.1GHz)\n", name, NWAVES, stage_kb,
           blocks, share, n_regions, bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 256 / 2.1e9);
}

int main() {
    const size_t total = 64ull << 20;     // 64 MiB: fits the 256 MiB MALL, spills the 32 MiB of L2s
    char* d; CK(hipMalloc(&d, total)); CK(hipMemset(d, 1, total));
    float* sink; CK(hipMalloc(&sink, 4));
    // footprint small enough to live in L2 (4 MiB per XCD): 16 regions of 1 MiB, each shared by many blocks
    run<8, true, 128>("DMA rows128, 1MiB regions, share 16", d, 16ull << 20, 1ull << 20, 16, 48, 256, sink);
    run<8, true, 128>("DMA rows128, 1MiB regions, share 64", d, 16ull << 20, 1ull << 20, 64, 48, 256, sink);
    run<8, true, 128>("DMA rows128, 1MiB regions, share 1", d, 16ull << 20, 1ull << 20, 1, 48, 256, sink);
    run<8, true, 128>("DMA rows128, 256KiB regions(64MiB tot) share 1", d, 64ull << 20, 256ull << 10, 1, 48, 256, sink);
    run<8, true, 128>("DMA rows128, ONE 1MiB region (all blocks)", d, 1ull << 20, 1ull << 20, 1, 48, 256, sink);
    run<8, true, 128>("DMA rows128, share 16, 2 blocks/CU (24K stage)", d, 16ull << 20, 1ull << 20, 16, 24, 512, sink);
    run<4, true, 128>("DMA rows128, 4 waves, share 16", d, 16ull << 20, 1ull << 20, 16, 48, 256, sink);
    run<16, true, 128>("DMA rows128, 16 waves, share 16", d, 16ull << 20, 1ull << 20, 16, 48, 256, sink);
    run<8, true, 1024>("DMA rows1024 (1KiB contiguous), share 16", d, 16ull << 20, 1ull << 20, 16, 48, 256, sink);
    run<8, true, 256>("DMA rows256, share 16", d, 16ull << 20, 1ull << 20, 16, 48, 256, sink);
    run<8, false, 128>("VGPR loads rows128, share 16", d, 16ull << 20, 1ull << 20, 16, 48, 256, sink);
    run<16, false, 128>("VGPR loads rows128, 16 waves, share 16", d, 16ull << 20, 1ull << 20, 16, 48, 256, sink);
    run<16, false, 1024>("VGPR loads contiguous, 16 waves x4 blocks", d, 16ull << 20, 1ull << 20, 16, 48, 1024, sink);
    return 0;
}
