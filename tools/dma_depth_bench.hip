// Microbenchmark: LDS-DMA delivery into ONE block of 4 (or 8) waves per CU as a function of the requests a wave keeps in
// flight, the contiguous bytes per matrix row of a request (64 / 128 / 256) and the instruction form (global_load_lds vs
// buffer_load ... lds).  Access pattern of a 256 x 256 GEMM tile: operand A = 256 rows of a [512 x K] matrix that every block
// reads (L2 resident), operand B = 256 rows of a [M x K] matrix that two neighbouring blocks share (first touch from HBM).
// Build: hipcc --offload-arch=gfx950 -O3 tools/dma_depth_bench.hip -o tools/dma_depth_bench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef __attribute__((address_space(3))) void lds_void_t;

template <int IMM>
__device__ __forceinline__ void dma_global(const void* sbase, unsigned voff, unsigned lds_base) {
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_base), "n"(IMM) : "scc");
}
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int IMM>
__device__ __forceinline__ void dma_buffer(i32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_base) {
    asm volatile("s_add_u32 m0, %3, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" : : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_base), "n"(IMM) : "scc");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// One "step" = 32 KiB per block (A image 16 KiB + B image 16 KiB), i.e. 32 / NW one-KiB requests per wave; DEPTH = requests a
// wave leaves in flight after each step's issue.  ROWB = contiguous bytes per matrix row inside one request.
template <int NW, int ROWB, int DEPTH, bool BUF, bool BARRIER>
__global__ __launch_bounds__(NW * 64) void k(const char* __restrict__ a, const char* __restrict__ b, int K2 /* row bytes */, int steps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void_t*)smem + wave * 1024);
    constexpr int CPR = ROWB / 16, RPI = 64 / CPR;            // chunks per row, rows per request
    constexpr int PER_WAVE = 16 / NW;                          // requests per wave per operand per step (16 KiB image)
    const int lrow = lane / CPR, lch = lane % CPR;
    // tile of this block: B rows [256 * (blockIdx.x / 2), +256) (two blocks share a pixel tile), A rows [256 * (blockIdx.x & 1), +256)
    const char* ab = a + (size_t)(blockIdx.x & 1) * 256 * K2;
    const char* bb = b + (size_t)(blockIdx.x >> 1) * 256 * K2;
    unsigned voff[PER_WAVE];
#pragma unroll
    for (int e = 0; e < PER_WAVE; ++e) voff[e] = (unsigned)(((e * NW + wave) * RPI + lrow) * K2 + lch * 16);
    // a 16 KiB image with ROWB bytes per row covers 16384 / ROWB rows: for ROWB = 128 two sub-steps cover all 256 rows
    const unsigned long long pa = (unsigned long long)ab, pb = (unsigned long long)bb;
    const i32x4 ra = {(int)(pa & 0xffffffffu), (int)((pa >> 32) & 0xffffu), 256 * K2, 0x00020000};
    const i32x4 rb = {(int)(pb & 0xffffffffu), (int)((pb >> 32) & 0xffffu), 256 * K2, 0x00020000};
    for (int s = 0; s < steps; ++s) {
        // which rows / columns: ROWB 64: all 256 rows, 64 B of K per step.  ROWB 128: rows alternate halves, 128 B of K per 2 steps
        const int rows_per_img = 16384 / ROWB;
        const int sub = s % (256 / rows_per_img);
        const unsigned col = (unsigned)((s / (256 / rows_per_img)) * ROWB % K2) + (unsigned)(sub * rows_per_img) * K2;
        const int slot = (s & 3) * 32768;
#pragma unroll
        for (int e = 0; e < PER_WAVE; ++e) {
            if (BUF) {
                dma_buffer<0>(ra, voff[e], col, lds_base + slot + e * NW * 1024);
                dma_buffer<16384>(rb, voff[e], col, lds_base + slot + e * NW * 1024);
            } else {
                dma_global<0>(ab + col, voff[e], lds_base + slot + e * NW * 1024);
                dma_global<16384>(bb + col, voff[e], lds_base + slot + e * NW * 1024);
            }
        }
        wait_vm<DEPTH>();
        if (BARRIER) asm volatile("s_barrier" ::: "memory");
    }
    wait_vm<0>();
}

template <int NW, int ROWB, int DEPTH, bool BUF, bool BARRIER>
void run(const char* a, const char* b, int K2, int blocks) {
    auto kern = k<NW, ROWB, DEPTH, BUF, BARRIER>;
    const int lds = 4 * 32768;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int steps = K2 / 64;             // one pass over K like the GEMM (K2 bytes per row / 64 B per step)
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(NW * 64), lds, 0, a, b, K2, steps);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(NW * 64), lds, 0, a, b, K2, steps);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)blocks * steps * 32768.0 * reps;
    printf("waves %d  row %3d B  in flight/wave %2d + %2d  %s %s : %7.1f us  %6.2f TB/s\n", NW, ROWB, DEPTH, 32 / NW, BUF ? "buffer_load" : "global_load",
           BARRIER ? "barrier" : "       ", ms / reps * 1e3, bytes / (ms * 1e-3) / 1e12);
}

int main() {
    const int K2 = 4096;                   // K = 2048 fp16
    const int blocks = 512;                // 256 pixel tiles x 2 cout tiles
    char *a, *b;
    CK(hipMalloc(&a, (size_t)512 * K2)); CK(hipMalloc(&b, (size_t)256 * 256 * K2));
    CK(hipMemset(a, 1, (size_t)512 * K2)); CK(hipMemset(b, 1, (size_t)256 * 256 * K2));
    run<4, 64, 8, false, false>(a, b, K2, blocks);
    run<4, 64, 16, false, false>(a, b, K2, blocks);
    run<4, 64, 24, false, false>(a, b, K2, blocks);
    run<4, 64, 40, false, false>(a, b, K2, blocks);
    run<4, 64, 16, false, true>(a, b, K2, blocks);
    run<4, 128, 8, false, false>(a, b, K2, blocks);
    run<4, 128, 16, false, false>(a, b, K2, blocks);
    run<4, 128, 24, false, false>(a, b, K2, blocks);
    run<4, 128, 40, false, false>(a, b, K2, blocks);
    run<4, 256, 16, false, false>(a, b, K2, blocks);
    run<4, 64, 16, true, false>(a, b, K2, blocks);
    run<4, 128, 16, true, false>(a, b, K2, blocks);
    run<4, 128, 40, true, false>(a, b, K2, blocks);
    run<8, 64, 8, false, false>(a, b, K2, blocks);
    run<8, 128, 8, false, false>(a, b, K2, blocks);
    run<8, 128, 16, false, false>(a, b, K2, blocks);
    run<8, 128, 16, true, false>(a, b, K2, blocks);
    return 0;
}
