// Persistent, weight-stationary, software-pipelined 1x1 convolution (c_in <= 512, c_out in 256-channel slabs).
//
// These layers (projection shortcut, conv1, conv3 of resnet_v2 block1 at 64x64: reference resnet_v2.py:120-138)
// move 170-340 MB per launch against 2-17 GFLOP: they are pure memory streams.  The tiled kernel
// (conv_igemm_f16_dma.hip) runs them as 4096 short-lived blocks whose load -> MFMA -> transpose -> store
// chains do not overlap (measured: 77-94 us where a plain copy of the same bytes takes 30-60 us, and removing
// either the stores or the MFMAs barely helps).  Here ONE block per CU stays resident and walks over pixel
// tiles (64 pixels x all 256 output channels = full 512-byte NHWC rows):
//   * the weights live in registers as MFMA A-fragments for the whole launch (32 VGPRs per lane);
//   * the next tile's input rows (8 KB) and shortcut rows (32 KB) are LDS-DMA'd while the current tile is
//     computed: each wave reads back only the shortcut bytes it requested itself, so they need no barrier;
//   * ordering is by counted s_waitcnt vmcnt(N): the only younger VMEM operations than the next tile's
//     loads are the current tile's stores, whose number per wave is fixed;
//   * the barriers are raw s_barrier (a __syncthreads() would drain the stores: vmcnt(0));
//   * optional second output (64 channels):
//       MODE2 = 1  extra weight rows on the SAME input (conv1 of the unit next to its projection shortcut,
//                  reference resnet_v2.py:122-128): 4 more MFMAs on the B fragments already in registers;
//       MODE2 = 2  conv1 of the NEXT unit on this launch's output (after the shortcut add), pre-activation
//                  applied in the row-wise pass, second GEMM from the LDS-resident tile (resnet_v2.py:119,127).
//       PSC        (with MODE2 = 2, block1/unit_1) the unit's PROJECTION shortcut is computed here instead of being
//                  read: a second 8 KB input tile (the unit's raw 64-channel input), Wsc in registers next to W3,
//                  sc = fp16(Wsc . fp16(relu(x * scale + shift)) + bias_sc) (resnet_v2.py:119,122-125) added to
//                  fp16(conv3 + bias) in registers -- 128 instead of 512 shortcut bytes per pixel, and the launch that
//                  wrote the shortcut tensor (512 more) is gone.
//       OUTM / REB (round 5, block1): the 256-channel residual stream of block1's stride-1 units is never materialised.
//                  x_1 = fp16(Wsc . pre(x0) + bsc) + fp16(W3_1 . t2_1 + b3_1) and x_2 = x_1 + fp16(W3_2 . t2_2 + b3_2) are
//                  functions of three 64-channel tensors (128 B per pixel each, against 512 B for a stored x): the launch of unit 1
//                  (PSC) keeps x_1 on chip -- it only feeds the next unit's conv1 (OUTM = 1: no store of `out`) -- and the
//                  launch of unit 2 REBUILDS it (REB: a third 8 KB input tile t2_1 and W3_1 in registers next to Wsc and W3_2;
//                  the same MFMAs in the same k order and the same fp16 roundings as the launch that would have stored it: the
//                  same bits), adds its own conv3 and again keeps the sum on chip.  Where the next unit is strided (its
//                  shortcut reads every second pixel, reference resnet_v2.py:113-121) the launch writes exactly those pixels as a
//                  compact [n, h/2, w/2, 256] tensor (OUTM = 2).  block1 at batch 256: 3.29 -> 2.08 GB per forward.
//       CB = 512   (K = 128, block2) ALL 512 output channels of a 32-pixel tile in one block (wave tile 64 couts x 32 pixels,
//                  W3 = 64 VGPRs per lane) so that MODE2 = 2 works there too: the next unit's conv1 (512 -> 128) needs every
//                  channel of a pixel.  Its weights W1' [128][512] live in REGISTERS as well -- wave w owns output rows
//                  16 w .. 16 w + 15 and runs 2 x 16 v_mfma_f32_16x16x32_f16 over the whole K = 512 from the LDS tile: no
//                  split K, no cross-wave reduction.  block2's conv1 launches (24 us each at batch 64, re-reading the 67 MB
//                  residual stream the conv3 launch just wrote) are gone.
// Arithmetic is that of the tiled kernel: fp32 accumulate, fp16(conv + bias), then the fp16 shortcut add.
#include <cstdlib>
#include <type_traits>

#include "metro_common.h"

namespace metro {

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

__device__ __attribute__((aligned(16))) unsigned int g_zero_page_pw[4];   // zero-initialised

struct Pw64Args {
    const half_t* in;          // [m_total][64]
    const half_t* w;           // [256][64]
    const float* bias;         // [256]
    const half_t* pro_scale;   // [64]  (PRO)
    const half_t* pro_shift;
    const half_t* residual;    // [m_total][256]  (RES)
    half_t* out;               // [m_total][256]
    const half_t* w2;          // MODE2 = 1: [64][64];  MODE2 = 2: [64][256] (CB = 256) / [128][512] (CB = 512)
    const float* bias2;        // [64]
    const half_t* scale2;      // [256]  (MODE2 = 2)
    const half_t* shift2;
    half_t* out2;              // [m_total][64]
    const half_t* x_sc;        // PSC: [m_total][64] raw unit input; pro_scale / pro_shift are ITS pre-activation
    const half_t* w_sc;        // PSC: [256][64]
    const float* bias_sc;      // PSC: [256]
    const half_t* in_b;        // REB: [m_total][64] conv2 output of the PREVIOUS unit
    const half_t* w_b;         // REB: [256][64] conv3 weights of the previous unit
    const float* bias_b;       // REB: [256]
    half_t* out_sub;           // OUTM = 2: compact [n][h_sub][w_sub][256]: pixels (sub_off + 2 i, sub_off + 2 j) of `out`
    int sub_off, h_sub, w_sub, lw_out;   // lw_out = log2(w_out) (the sub-sampled store needs a power of two)
    int m_total, n_tiles;
    int c_out;                 // CB * (number of CB-channel slabs); block b serves slab b % halves
    // sub-sampled shortcut (units with stride 2, reference resnet_v2.py:113-118: max_pool2d 1x1 stride 2 of the
    // unit input): output pixel (ho, wo) adds input pixel (res_off + 2*ho, res_off + 2*wo) of a res_h x res_w map
    int res_stride, res_off, res_h, res_w, h_out, w_out;
};

namespace pw {
constexpr int NW = 8, NT = 512;
// WM = waves along the output channels of a block, CB = output channels per block:
//   WM 4, CB 256: wave tile 64 couts x 32 pixels, 64-pixel tiles (weights K/2 VGPRs per lane: K <= 128)
//   WM 8, CB 256: wave tile 32 couts x 32 pixels, 32-pixel tiles (K/4 VGPRs: K <= 512)
//   WM 8, CB 512: wave tile 64 couts x 32 pixels, 32-pixel tiles (K/2 VGPRs: K = 128), every channel of a pixel in one block
template <int K, int WM, int CB = 256>
struct Lay {
    static constexpr int WN = NW / WM;
    static constexpr int NI = CB / 32 / WM;         // 32-row MFMA tiles per wave
    static constexpr int TN = 32 * WN;              // pixels per tile
    static constexpr int KS = K / 64;               // input tile = KS swizzled slices of TN rows x 64 fp16
    static constexpr int SL_BYTES = TN * 128;
    static constexpr int X_BYTES = KS * SL_BYTES;
    static constexpr int XI = X_BYTES / 1024 / NW;  // input DMA instructions per wave per tile
    static constexpr int CPR = CB / 8;              // 16-byte chunks per output row
    static constexpr int RI = TN * CPR / NT;        // row-wise iterations = shortcut DMA instructions = stores per wave
    static constexpr int OUT_ROW = CB * 2 + 16;     // padded rows of the [pixel][cout] tile
    static constexpr int OUT_BYTES = TN * OUT_ROW;
    static constexpr int RES_BYTES = TN * CB * 2;
    static constexpr int C2 = CB == 512 ? 128 : 64; // channels of a second output
    // bias[CB] f32 | bias2[128] f32 | pro scale[K] | pro shift[K] fp16 | bias_sc[256] f32 (PSC) | bias_b[256] f32 (REB)
    static constexpr int BIAS2_OFF = CB * 4, PRO_OFF = BIAS2_OFF + 512, BSC_OFF = PRO_OFF + 4 * K, BB_OFF = BSC_OFF + 1024;
    static constexpr int PAR_BYTES = BB_OFF + 1024;
    static constexpr int X_OFF = 0;                 // 2 buffers
    static constexpr int OUT_OFF = X_OFF + 2 * X_BYTES;
    static constexpr int PAR_OFF = OUT_OFF + OUT_BYTES;
    static constexpr int RES_OFF = PAR_OFF + PAR_BYTES;   // 2 buffers (RES)
    static_assert(X_BYTES % (1024 * NW) == 0, "input tile must split evenly over the waves");
    static_assert(NI >= 1 && RI >= 1 && (TN * CPR) % NT == 0, "tile / thread mismatch");
};
template <int K, int WM, bool RES, int MODE2, bool PSC = false, int CB = 256, bool REB = false>
constexpr int lds_bytes() {
    return Lay<K, WM, CB>::RES_OFF + (RES ? 2 * Lay<K, WM, CB>::RES_BYTES : PSC ? (REB ? 4 : 2) * Lay<K, WM, CB>::X_BYTES
                                      : (MODE2 == 1 && K > 64) ? Lay<K, WM, CB>::C2 * (2 * K + 16) + Lay<K, WM, CB>::TN * (2 * Lay<K, WM, CB>::C2 + 16) : 0);
}
}  // namespace pw

__device__ __forceinline__ int pw_swz(int row) { return (row >> 1) & 7; }

__device__ __forceinline__ void pw_dma16(const void* gsrc, unsigned lds_addr) {
    asm volatile(
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, off"
        :
        : "v"(gsrc), "s"(lds_addr));
}
template <int N>
__device__ __forceinline__ void pw_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// vmcnt(BASE + extra), extra in [0, MAXX] wave-uniform: the wait count is an immediate
template <int BASE, int MAXX>
__device__ __forceinline__ void pw_wait_vm_plus(int extra) {
    if constexpr (MAXX == 0) {
        pw_wait_vm<BASE>();
    } else {
        if (extra >= MAXX) pw_wait_vm<BASE + MAXX>();
        else pw_wait_vm_plus<BASE, MAXX - 1>(extra);
    }
}
// s_waitcnt lgkmcnt(n), n a compile-time value after unrolling, tied to the register(s) the wait is for
__device__ __forceinline__ void pw_wait_lgkm(half8_t& r, int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r)); break;
        case 1: asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(r)); break;
        case 2: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(r)); break;
        default: asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(r)); break;
    }
}
__device__ __forceinline__ void pw_wait_lgkm2(half8_t& r, half8_t& r2, int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r), "+v"(r2)); break;
        case 2: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(r), "+v"(r2)); break;
        case 4: asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(r), "+v"(r2)); break;
        default: asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(r), "+v"(r2)); break;
    }
}
__device__ __forceinline__ void pw_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int K, int WM, bool PRO, bool RES, int MODE2, bool RSUB = false, bool PSC = false, int CB = 256, bool REB = false, int OUTM = 0>
__global__ __launch_bounds__(pw::NT) void conv_pw64_kernel(Pw64Args a) {
    using namespace pw;
    using L = Lay<K, WM, CB>;
    static_assert((WM == 4 && K <= 128) || (WM == 8 && K <= 512), "weights must fit the register file as MFMA fragments");
    static_assert(CB == 256 || (CB == 512 && K == 128 && WM == 8 && RES && !PRO && !RSUB) ||
                      (CB == 512 && K == 256 && WM == 8 && !RES && PRO && MODE2 == 1),
                  "512-channel blocks: conv3 of block2, or block2's projection shortcut + conv1 pair");
    static_assert(MODE2 == 0 || (K == 64 && WM == 4) || (MODE2 == 2 && CB == 512) || (MODE2 == 1 && CB == 512 && K == 256),
                  "second outputs: block1 shapes, conv3 + next conv1 of block2, or block2's pair");
    static_assert(!PSC || (MODE2 == 2 && !PRO && !RES), "in-launch projection shortcut: conv3 + next conv1 of block1/unit_1");
    static_assert(!REB || PSC, "rebuilt residual: on top of the in-launch projection shortcut");
    static_assert(OUTM == 0 || (MODE2 == 2 && PSC), "outputs kept on chip: the conv3 + next conv1 launches of block1");
    constexpr int KK = K / 16, WN = L::WN, NI = L::NI, TN = L::TN, XI = L::XI, RI = L::RI;
    constexpr int X_BYTES = L::X_BYTES, X_OFF = L::X_OFF, OUT_OFF = L::OUT_OFF, PAR_OFF = L::PAR_OFF, RES_OFF = L::RES_OFF,
                  RES_BYTES = L::RES_BYTES, SL_BYTES = L::SL_BYTES, OUT_ROW = L::OUT_ROW, CPR = L::CPR, C2 = L::C2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void lds_void_t;
    const unsigned smem_base = (unsigned)(size_t)(lds_void_t*)smem;
    constexpr int XS_OFF = RES_OFF;                          // PSC: two buffers of the unit-input tile where the shortcut rows would be
    constexpr int XB_OFF = XS_OFF + 2 * X_BYTES;             // REB: two buffers of the previous unit's conv2 output tile

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;           // wave tile: couts wm*32*NI.., pixels wn*32..+32
    const int frag_row = lane & 31, frag_half = lane >> 5;
    // CB-channel slabs of wider outputs go to different blocks (the weights are register-resident).  Blocks are
    // dealt round-robin to the 8 XCDs (one L2 each): the `halves` blocks that share an input tile are placed on the
    // SAME XCD (PMC: with consecutive block ids block4's conv3 fetched its input 8 times, 203 MB instead of 84 MB).
    const int halves = a.c_out / CB;
    const int G = gridDim.x / halves;                       // tile streams
    int half, t;
    if ((gridDim.x & 7) == 0 && ((gridDim.x >> 3) % halves) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        half = j % halves;
        t = xcd * ((gridDim.x >> 3) / halves) + j / halves;
    } else {
        half = blockIdx.x % halves;
        t = blockIdx.x / halves;
    }
    if (t >= a.n_tiles) return;
    a.w += (size_t)half * CB * K;
    a.bias += half * CB;
    a.out += half * CB;
    if (RES) a.residual += half * CB;
    const int ldo = a.c_out;
    const half_t* zero = reinterpret_cast<const half_t*>(g_zero_page_pw);

    // ---- launch-resident operands ------------------------------------------------------------
    half8_t wf[NI][KK];
    constexpr bool W_STAGED = K >= 128 && pw::lds_bytes<K, WM, RES, MODE2, PSC, CB, REB>() >= NW * W_STAGE_BYTES;   // through LDS (metro_common.h)
    constexpr int RG = C2 / 16;                              // MODE2 = 2: row groups of W2: 4 (CB 256) or 8 (CB 512)
    constexpr int NTW = TN / 16 / (NW / RG);                 // 16-pixel tiles per wave: 2
    constexpr int KS2 = CB / 32;                             // k steps of 32
    half8_t w2r[MODE2 == 2 ? KS2 : 1];                       // W2 [C2][CB] as 16x16x32 A fragments (see below)
    if constexpr (W_STAGED) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
            load_w_frags_staged<K>(a.w + (size_t)((wm * NI + i) * 32) * K, wf[i], smem + wave * W_STAGE_BYTES, lane);
        // W2's fragments take the same road (256-byte row pieces, not 64-byte ones: 4x fewer L2 requests)
        if constexpr (MODE2 == 2) load_w_frags16_staged<CB>(a.w2 + (size_t)((wave % RG) * 16) * CB, w2r, smem + wave * W_STAGE_BYTES, lane);
        __syncthreads();                                     // the scratch overlays the tile buffers and the parameter block
    } else {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
                wf[i][kk] = *reinterpret_cast<const half8_t*>(a.w + (size_t)((wm * NI + i) * 32 + frag_row) * K + kk * 16 + frag_half * 8);
    }
    half8_t wsf[NI][KK];                                     // PSC: the projection shortcut's weights, same fragment layout
    if constexpr (PSC) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
                wsf[i][kk] = *reinterpret_cast<const half8_t*>(a.w_sc + (size_t)((wm * NI + i) * 32 + frag_row) * K + kk * 16 + frag_half * 8);
    }
    half8_t wbf[NI][KK];                                     // REB: the previous unit's conv3 weights
    if constexpr (REB) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
                wbf[i][kk] = *reinterpret_cast<const half8_t*>(a.w_b + (size_t)((wm * NI + i) * 32 + frag_row) * K + kk * 16 + frag_half * 8);
    }
    // MODE2 = 1: conv1's rows [C2][K] next to the shortcut's; waves 0-3 own one 32-row tile each (K = 64: 2 row tiles x 2 pixel
    // tiles; K = 256, CB = 512: 4 row tiles x the one pixel tile -- waves w and w + 4 share a SIMD, so every SIMD issues 3 + 2 MFMAs
    // per k step)
    // K = 256: 128 (W) + 64 (W2) fragment registers next to 48 accumulators do not fit 256 VGPRs (80 spilled): there W2 lives in LDS,
    // rows padded to 2 K + 16 bytes (conflict-free ds_read_b128 of 32 rows), one more fragment read per k step for waves 0-3
    constexpr bool W2_LDS = MODE2 == 1 && K > 64;
    constexpr bool PREPASS = PRO && W2_LDS;
    constexpr int W2_ROW = 2 * K + 16;
    constexpr int W2_OFF = RES_OFF;                          // (MODE2 = 1 has no shortcut rows: !RES)
    constexpr int O2_ROW = 2 * C2 + 16, O2_OFF = W2_OFF + C2 * W2_ROW;   // ... and conv1's output tile [TN][C2] goes out in full rows too
    half8_t w2f[MODE2 == 1 && !W2_LDS ? KK : 1];
    if constexpr (MODE2 == 1 && !W2_LDS) {
        if (wave < 4) {
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
                w2f[kk] = *reinterpret_cast<const half8_t*>(a.w2 + (size_t)(wm * 32 + frag_row) * K + kk * 16 + frag_half * 8);
        }
    }
    if constexpr (W2_LDS) {
        for (int c = tid; c < C2 * (K / 8); c += NT) {
            const int r = c / (K / 8), q = c % (K / 8);
            *reinterpret_cast<uint4*>(smem + W2_OFF + r * W2_ROW + q * 16) = *reinterpret_cast<const uint4*>(a.w2 + (size_t)r * K + q * 8);
        }
    }
    const char* w2l = smem + W2_OFF + (wm * 32 + frag_row) * W2_ROW + frag_half * 16;
    float* bias_l = reinterpret_cast<float*>(smem + PAR_OFF);
    float* bias2_l = reinterpret_cast<float*>(smem + PAR_OFF + L::BIAS2_OFF);
    half_t* pro_l = reinterpret_cast<half_t*>(smem + PAR_OFF + L::PRO_OFF);
    if (tid < CB) bias_l[tid] = a.bias[tid];
    if (MODE2 != 0 && tid < C2) bias2_l[tid] = a.bias2[tid];
    if ((PRO || PSC) && tid < K) { pro_l[tid] = a.pro_scale[tid]; pro_l[K + tid] = a.pro_shift[tid]; }
    float* bias_sc_l = reinterpret_cast<float*>(smem + PAR_OFF + L::BSC_OFF);
    if (PSC && tid < 256) bias_sc_l[tid] = a.bias_sc[tid];
    float* bias_b_l = reinterpret_cast<float*>(smem + PAR_OFF + L::BB_OFF);
    if (REB && tid < 256) bias_b_l[tid] = a.bias_b[tid];
    // row-wise pass: this thread always owns 16-byte chunk `ch` of a row
    const int ch = tid & (CPR - 1);
    half8_t sc2 = {}, sh2 = {};
    // MODE2 = 2: W2 [C2][CB] (the next unit's conv1) in registers as 16x16x32 A fragments: wave w owns output rows 16 (w % RG) ..
    // + 15 over the whole K = CB, for the pixels of its pixel group w / RG (round 3: the 64 -> 256 kernels ran this GEMM on four
    // of their eight waves from an LDS image of W2)
    if constexpr (MODE2 == 2) {
        sc2 = *reinterpret_cast<const half8_t*>(a.scale2 + ch * 8);
        sh2 = *reinterpret_cast<const half8_t*>(a.shift2 + ch * 8);
        if constexpr (!W_STAGED) {
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks)
                w2r[ks] = *reinterpret_cast<const half8_t*>(a.w2 + (size_t)((wave % RG) * 16 + (lane & 15)) * CB + ks * 32 + (lane >> 4) * 8);
        }
    }

    // ---- per-lane DMA coordinates ------------------------------------------------------------
    // input tile = K/64 slices [TN rows][64 k] (128-byte rows, chunk-swizzled); DMA instruction q = i*8 + wave
    // fills 8 rows of slice q / (TN/8)
    int xrow[XI], xoff[XI];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int q = i * NW + wave;
        xrow[i] = (q % (TN / 8)) * 8 + (lane >> 3);
        xoff[i] = xrow[i] * K + (q / (TN / 8)) * 64 + (((lane & 7) ^ pw_swz(xrow[i])) * 8);
    }
    auto issue_tile = [&](int tile, int buf) {
        const int m0 = tile * TN;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const half_t* xs = (m0 + xrow[i] < a.m_total) ? a.in + (size_t)m0 * K + xoff[i] : zero;
            pw_dma16(xs, __builtin_amdgcn_readfirstlane(smem_base + X_OFF + buf * X_BYTES + (i * NW + wave) * 1024));
            if constexpr (PSC) {
                const half_t* ss = (m0 + xrow[i] < a.m_total) ? a.x_sc + (size_t)m0 * K + xoff[i] : zero;
                pw_dma16(ss, __builtin_amdgcn_readfirstlane(smem_base + XS_OFF + buf * X_BYTES + (i * NW + wave) * 1024));
            }
            if constexpr (REB) {
                const half_t* sb = (m0 + xrow[i] < a.m_total) ? a.in_b + (size_t)m0 * K + xoff[i] : zero;
                pw_dma16(sb, __builtin_amdgcn_readfirstlane(smem_base + XB_OFF + buf * X_BYTES + (i * NW + wave) * 1024));
            }
        }
        if constexpr (RES) {
            // shortcut rows: chunk c = it*512 + tid (row c / CPR, 16-byte column c % CPR) lands at c*16, i.e. every
            // wave later reads back exactly the bytes it requested
#pragma unroll
            for (int it = 0; it < RI; ++it) {
                const int c = it * NT + tid;
                const int m = m0 + c / CPR;
                size_t rrow = m;
                if constexpr (RSUB) {
                    const int hw = a.h_out * a.w_out;
                    const int img = m / hw, rem = m - img * hw;
                    const int ho = rem / a.w_out, wo = rem - ho * a.w_out;
                    rrow = ((size_t)img * a.res_h + a.res_off + a.res_stride * ho) * a.res_w + a.res_off + a.res_stride * wo;
                }
                const half_t* rs = m < a.m_total ? a.residual + rrow * ldo + (c & (CPR - 1)) * 8 : zero;
                pw_dma16(rs, __builtin_amdgcn_readfirstlane(smem_base + RES_OFF + buf * RES_BYTES + it * 8192 + wave * 1024));
            }
        }
    };

    issue_tile(t, 0);
    // stores of one tile per wave (all younger than the next tile's loads): RI row-wise (+4 second-output)
    // second-output stores per tile: MODE2 = 1 four 8-byte stores by waves 0-3; MODE2 = 2 NTW by every wave
    constexpr int S2 = MODE2 == 2 ? NTW : (MODE2 == 1 && K > 64) ? 1 : 4;
    constexpr int RS = OUTM == 0 ? RI : 0;                   // row-wise stores of `out` every tile issues (OUTM = 2: 0 .. RI, per tile)
    const bool two = MODE2 == 2 || (MODE2 == 1 && (wave < 4 || K > 64));
    bool prev_full = false;
    int prev_sub = 0;                                        // OUTM = 2: sub-sampled row-wise stores the previous tile issued (wave-uniform)
    for (int it = 0;; ++it, t += G) {
        const int buf = it & 1;
        const int m0 = t * TN;
        // ---- the tile's loads have landed (for this wave), then for every wave -----------------
        if (it == 0 || !prev_full) pw_wait_vm<0>();
        else if constexpr (OUTM == 2) pw_wait_vm_plus<S2, RI>(prev_sub);
        else if (two) pw_wait_vm<RS + S2>();
        else pw_wait_vm<RS>();
        pw_barrier();
        if (t + G < a.n_tiles) issue_tile(t + G, buf ^ 1);
        prev_full = m0 + TN <= a.m_total;

        // ---- K = 256 pair: the pre-activation ONCE per element, in place in the landed tile (resnet_v2.py:119: fp16 BN + ReLU), not
        //      once per fragment read in each of the 8 waves: per k step a wave is left with one fragment read and its MFMAs
        if constexpr (PREPASS) {
            static_assert(X_BYTES / 16 == 2 * NT, "two 16-byte chunks of the input tile per thread");
            const int c8 = lane & 7, prow = (wave & 3) * 8 + (lane >> 3);
            const half8_t z = {};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int sl = (wave >> 2) + 2 * h;
                half8_t* xp = reinterpret_cast<half8_t*>(smem + X_OFF + buf * X_BYTES + sl * SL_BYTES + prow * 128 + ((c8 ^ pw_swz(prow)) << 4));
                const half8_t sc = *reinterpret_cast<const half8_t*>(pro_l + sl * 64 + c8 * 8);
                const half8_t sh = *reinterpret_cast<const half8_t*>(pro_l + K + sl * 64 + c8 * 8);
                *xp = __builtin_elementwise_max(*xp * sc + sh, z);
            }
            pw_barrier();
        }
        // ---- GEMM 1: [256 x 64] x [64 x 64 pixels] ---------------------------------------------
        floatx16 acc[NI], acc2, accs[NI];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
#pragma unroll
            for (int i = 0; i < NI; ++i) { acc[i][e] = 0.f; accs[i][e] = 0.f; }
            acc2[e] = 0.f;
        }
        const char* xl = smem + X_OFF + buf * X_BYTES;
        const int brow = wn * 32 + frag_row;
        // REB: the residual x_prev = fp16(W3_prev . t2_prev + b3_prev) + fp16(Wsc . pre(x0) + bsc) first, down to packed fp16
        // (16 registers), so that its two accumulator sets are dead before this unit's conv3 accumulates
        half4_t xprev[NI][4];
        if constexpr (REB) {
            // one GEMM at a time (the projection shortcut, then the previous conv3): 32 live accumulator registers, not 64
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                const int chunk = (kk & 3) * 2 + frag_half;
                half8_t bs = *reinterpret_cast<const half8_t*>(smem + XS_OFF + buf * X_BYTES + (kk >> 2) * SL_BYTES + brow * 128 + ((chunk ^ pw_swz(brow)) << 4));
                const half8_t s = *reinterpret_cast<const half8_t*>(pro_l + kk * 16 + frag_half * 8);
                const half8_t b = *reinterpret_cast<const half8_t*>(pro_l + K + kk * 16 + frag_half * 8);
                const half8_t z = {};
                bs = __builtin_elementwise_max(bs * s + b, z);
#pragma unroll
                for (int i = 0; i < NI; ++i) accs[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wsf[i][kk], bs, accs[i], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const floatx4 bs = *reinterpret_cast<const floatx4*>(bias_sc_l + (wm * NI + i) * 32 + 8 * q + 4 * frag_half);
#pragma unroll
                    for (int e = 0; e < 4; ++e) xprev[i][q][e] = (half_t)(accs[i][4 * q + e] + bs[e]);
                }
            floatx16 accb[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) accb[i][e] = 0.f;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                const int chunk = (kk & 3) * 2 + frag_half;
                const half8_t bb = *reinterpret_cast<const half8_t*>(smem + XB_OFF + buf * X_BYTES + (kk >> 2) * SL_BYTES + brow * 128 + ((chunk ^ pw_swz(brow)) << 4));
#pragma unroll
                for (int i = 0; i < NI; ++i) accb[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wbf[i][kk], bb, accb[i], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const floatx4 bv = *reinterpret_cast<const floatx4*>(bias_b_l + (wm * NI + i) * 32 + 8 * q + 4 * frag_half);
                    half4_t hb;
#pragma unroll
                    for (int e = 0; e < 4; ++e) hb[e] = (half_t)(accb[i][4 * q + e] + bv[e]);
                    xprev[i][q] = hb + xprev[i][q];      // the fp16 Add of the previous unit (resnet_v2.py:138), as its own launch computes it
                }
        }
        if constexpr (PREPASS) {
            // The fragments run DEPTH k steps ahead of their MFMAs through a register ring, reads and waits in inline asm with counted
            // lgkmcnt (conv_pws.hip: hipcc turns the C++ form into read - wait - MFMA, one exposed LDS round trip per k step and two in
            // waves 0-3, which also read conv1's A fragment).  The role (with / without conv1 rows) is wave-uniform: two loops.
            constexpr int DEPTH = 3;
            unsigned fbase[4];
#pragma unroll
            for (int c = 0; c < 4; ++c)
                fbase[c] = smem_base + X_OFF + buf * X_BYTES + brow * 128 + (((2 * c + frag_half) ^ pw_swz(brow)) << 4);
            const unsigned f2base = smem_base + W2_OFF + (wm * 32 + frag_row) * W2_ROW + frag_half * 16;
            auto gemm = [&](auto has2_c) {
                constexpr bool HAS2 = decltype(has2_c)::value;
                constexpr int PER = HAS2 ? 2 : 1;               // LDS reads per k step
                half8_t fr[DEPTH], f2[HAS2 ? DEPTH : 1];
#pragma unroll
                for (int kk = 0; kk < DEPTH; ++kk) {
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[kk]) : "v"(fbase[kk & 3]), "n"((kk >> 2) * SL_BYTES));
                    if constexpr (HAS2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f2[kk]) : "v"(f2base), "n"(kk * 32));
                }
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) {
                    constexpr int D1 = DEPTH - 1;
                    const int younger = (KK - 1 - kk < D1 ? KK - 1 - kk : D1) * PER;     // a constant after unrolling
                    if constexpr (HAS2) pw_wait_lgkm2(fr[kk % DEPTH], f2[kk % DEPTH], younger);
                    else pw_wait_lgkm(fr[kk % DEPTH], younger);
#pragma unroll
                    for (int i = 0; i < NI; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i][kk], fr[kk % DEPTH], acc[i], 0, 0, 0);
                    if constexpr (HAS2) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f2[kk % DEPTH], fr[kk % DEPTH], acc2, 0, 0, 0);
                    if (kk + DEPTH < KK) {
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[kk % DEPTH]) : "v"(fbase[(kk + DEPTH) & 3]), "n"(((kk + DEPTH) >> 2) * SL_BYTES));
                        if constexpr (HAS2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f2[kk % DEPTH]) : "v"(f2base), "n"((kk + DEPTH) * 32));
                    }
                }
            };
            if (wave < 4) gemm(std::true_type{});
            else gemm(std::false_type{});
        } else {
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const int chunk = (kk & 3) * 2 + frag_half;
            half8_t bf = *reinterpret_cast<const half8_t*>(xl + (kk >> 2) * SL_BYTES + brow * 128 + ((chunk ^ pw_swz(brow)) << 4));
            if constexpr (PRO && !PREPASS) {
                const half8_t s = *reinterpret_cast<const half8_t*>(pro_l + kk * 16 + frag_half * 8);
                const half8_t b = *reinterpret_cast<const half8_t*>(pro_l + K + kk * 16 + frag_half * 8);
                const half8_t z = {};
                bf = __builtin_elementwise_max(bf * s + b, z);
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i][kk], bf, acc[i], 0, 0, 0);
            if constexpr (MODE2 == 1 && !W2_LDS) {
                if (wave < 4) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2f[kk], bf, acc2, 0, 0, 0);
            }
            if constexpr (W2_LDS) {
                if (wave < 4) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const half8_t*>(w2l + kk * 32), bf, acc2, 0, 0, 0);
            }
            if constexpr (PSC && !REB) {
                // projection shortcut on the pre-activated unit input (same pixels, same k step)
                half8_t bs = *reinterpret_cast<const half8_t*>(smem + XS_OFF + buf * X_BYTES + (kk >> 2) * SL_BYTES + brow * 128 + ((chunk ^ pw_swz(brow)) << 4));
                const half8_t s = *reinterpret_cast<const half8_t*>(pro_l + kk * 16 + frag_half * 8);
                const half8_t b = *reinterpret_cast<const half8_t*>(pro_l + K + kk * 16 + frag_half * 8);
                const half8_t z = {};
                bs = __builtin_elementwise_max(bs * s + b, z);
#pragma unroll
                for (int i = 0; i < NI; ++i) accs[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wsf[i][kk], bs, accs[i], 0, 0, 0);
            }
        }
        }
        if constexpr (MODE2 == 1) {
            // conv1 rows: relu(acc + bias2) -> out2[m][64]   (waves 0..3: couts wm*32.., pixels wn*32..)
            if (wave < 4) {
                const int m = m0 + wn * 32 + frag_row;
                if (m < a.m_total) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int co = wm * 32 + 8 * q + 4 * frag_half;
                        const floatx4 bv = *reinterpret_cast<const floatx4*>(bias2_l + co);
                        half4_t hv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) hv[e] = (half_t)fmaxf(acc2[4 * q + e] + bv[e], 0.f);
                        if constexpr (W2_LDS) *reinterpret_cast<half4_t*>(smem + O2_OFF + (wn * 32 + frag_row) * O2_ROW + co * 2) = hv;   // full rows below
                        else *reinterpret_cast<half4_t*>(a.out2 + (size_t)m * C2 + co) = hv;
                    }
                }
            }
        }
        // ---- accumulators (+bias) -> LDS tile [pixel][cout] fp16 -------------------------------
        char* ol = smem + OUT_OFF;
        // the tile's 4 NI bias reads FIRST, back to back (hipcc's own order is read - wait - convert - write per group of four channels:
        // 4 NI exposed LDS round trips per tile in every wave); the B fragments are dead by now, their registers hold the bias
        floatx4 bvv[NI][4];
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) bvv[i][q] = *reinterpret_cast<const floatx4*>(bias_l + (wm * NI + i) * 32 + 8 * q + 4 * frag_half);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = (wm * NI + i) * 32 + 8 * q + 4 * frag_half;
                const floatx4 bv = bvv[i][q];
                half4_t hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) hv[e] = (half_t)(acc[i][4 * q + e] + bv[e]);
                if constexpr (REB) hv = hv + xprev[i][q];      // identity shortcut of this unit: the rebuilt x_prev (resnet_v2.py:120-121,138)
                if constexpr (PSC && !REB) {
                    // fp16(shortcut conv + bias) + fp16(conv3 + bias): the fp16 Add of the reference graph (resnet_v2.py:138)
                    const floatx4 bs = *reinterpret_cast<const floatx4*>(bias_sc_l + col);
                    half4_t hs;
#pragma unroll
                    for (int e = 0; e < 4; ++e) hs[e] = (half_t)(accs[i][4 * q + e] + bs[e]);
                    hv = hv + hs;
                }
                *reinterpret_cast<half4_t*>(ol + brow * OUT_ROW + col * 2) = hv;
            }
        }
        pw_barrier();
        // ---- row-wise: 16 bytes per lane, + shortcut, full-row stores ---------------------------
        const char* rl = smem + RES_OFF + buf * RES_BYTES;
        // OUTM = 2: which of the RI row-wise instructions of this tile hold pixels the next (strided) unit's shortcut reads.  A
        // wave's instruction r covers pixels 16 r + 2 w, + 1 of the tile: one map row (w_out >= 16, a power of two), both column
        // parities -- so whether it stores is a property of (tile, r), the same for every wave: the store count stays countable.
        int img0 = 0, rem0 = 0, now_sub = 0;
        if constexpr (OUTM == 2) {
            const int hw = a.h_out * a.w_out;
            img0 = m0 / hw;
            rem0 = m0 - img0 * hw;
        }
        // OUTM = 0: the RI chunk (and shortcut) reads of the pass back to back, then sums and stores.  A tile that lies inside the
        // tensor (wave-uniform test; every tile but the last) stores without per-lane guards: a guard is an exec-mask branch per
        // iteration, and hipcc does not move the next iteration's LDS reads across it (read - wait - add - store, RI times)
        uint4 vpre[OUTM == 0 ? RI : 1], rpre[OUTM == 0 && RES ? RI : 1];
        if constexpr (OUTM == 0) {
#pragma unroll
            for (int r = 0; r < RI; ++r) {
                const int idx = tid + r * NT;
                vpre[r] = *reinterpret_cast<const uint4*>(ol + (idx / CPR) * OUT_ROW + ch * 16);
                if constexpr (RES) rpre[r] = *reinterpret_cast<const uint4*>(rl + idx * 16);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const bool inside = m0 + TN <= a.m_total;
#pragma unroll
        for (int r = 0; r < RI; ++r) {
            const int idx = tid + r * NT;
            const int prow = idx / CPR;
            const int m = m0 + prow;
            uint4 v;
            if constexpr (OUTM == 0) v = vpre[r];
            else v = *reinterpret_cast<const uint4*>(ol + prow * OUT_ROW + ch * 16);
            if constexpr (RES) {
                uint4 rv;
                if constexpr (OUTM == 0) rv = rpre[r];
                else rv = *reinterpret_cast<const uint4*>(rl + idx * 16);
                half2_t* x = reinterpret_cast<half2_t*>(&v);
                const half2_t* rr = reinterpret_cast<const half2_t*>(&rv);
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = x[e] + rr[e];       // fp16 Add, like the reference graph
            }
            if constexpr (OUTM == 0) {
                if (inside) store_out16<1>(a.out + (size_t)m * ldo + ch * 8, v);
                else if (m < a.m_total) store_out16<1>(a.out + (size_t)m * ldo + ch * 8, v);
            }
            if constexpr (OUTM == 2) {
                const int hr = ((rem0 + 16 * r) >> a.lw_out) - a.sub_off;         // wave-uniform: map row of this instruction's pixels
                if (hr >= 0 && (hr & 1) == 0 && (hr >> 1) < a.h_sub) {
                    ++now_sub;
                    const int wo = ((rem0 + prow) & (a.w_out - 1)) - a.sub_off;
                    if (wo >= 0 && (wo & 1) == 0 && (wo >> 1) < a.w_sub)
                        store_out16<1>(a.out_sub + (((size_t)img0 * a.h_sub + (hr >> 1)) * a.w_sub + (wo >> 1)) * CB + ch * 8, v);
                }
            }
            if constexpr (MODE2 == 2) {
                // next unit's pre-activation (fp16 BN + ReLU) goes back into the tile for the second GEMM
                const half8_t z = {};
                half8_t p = *reinterpret_cast<const half8_t*>(&v);
                p = __builtin_elementwise_max(p * sc2 + sh2, z);
                *reinterpret_cast<half8_t*>(ol + prow * OUT_ROW + ch * 16) = p;
            }
        }
        if constexpr (W2_LDS) {
            // conv1's rows: TN x C2 / 8 = 512 chunks of 16 bytes, one per thread
            static_assert(TN * (C2 / 8) == NT, "one 16-byte chunk of the second output per thread");
            const int prow2 = tid / (C2 / 8), ch2 = tid % (C2 / 8);
            const uint4 v2 = *reinterpret_cast<const uint4*>(smem + O2_OFF + prow2 * O2_ROW + ch2 * 16);
            if (m0 + prow2 < a.m_total) store_out16<1>(a.out2 + (size_t)(m0 + prow2) * C2 + ch2 * 8, v2);
        }
        if constexpr (MODE2 == 2) {
            pw_barrier();
            // ---- GEMM 2: [C2 x CB] x [CB x TN pixels] from the tile: v_mfma_f32_16x16x32_f16, A[i][k] lane (i = lane & 15, k group =
            //      lane >> 4), B likewise by pixel, D[i][j] lane (j = lane & 15 -> pixel, rows 4 (lane >> 4) .. + 3)
            floatx4 dacc[NTW];
#pragma unroll
            for (int j = 0; j < NTW; ++j) dacc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
            const int px0 = (wave / RG) * (NTW * 16);
            const char* bl = ol + (px0 + (lane & 15)) * OUT_ROW + (lane >> 4) * 16;
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) {
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    const half8_t bj = *reinterpret_cast<const half8_t*>(bl + j * 16 * OUT_ROW + ks * 64);
                    dacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2r[ks], bj, dacc[j], 0, 0, 0);
                }
            }
            const int co = (wave % RG) * 16 + (lane >> 4) * 4;
            const floatx4 bv = *reinterpret_cast<const floatx4*>(bias2_l + co);
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                const int m = m0 + px0 + j * 16 + (lane & 15);
                half4_t hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) hv[e] = (half_t)fmaxf(dacc[j][e] + bv[e], 0.f);
                if (m < a.m_total) *reinterpret_cast<half4_t*>(a.out2 + (size_t)m * C2 + co) = hv;
            }
        }
        prev_sub = now_sub;
        if (t + G >= a.n_tiles) break;
    }
}

static int pw_env_int(const char* name, int dflt) { return tuning_knob(name, dflt); }

static bool pw_enabled() {
    static const int v = pw_env_int("METRO_PW64", 1);
    return v != 0;
}

// mode: 0 plain, 1 shortcut + conv1 pair (desc.c_out = 320 = 256 + 64 concatenated rows), 2 conv3 + next conv1
bool conv_pw64_supported(const MetroConvDesc& d, int mode) {
    if (!pw_enabled()) return false;
    if (!(d.kh == 1 && d.kw == 1 && d.stride == 1 && d.pad_top == 0 && d.pad_left == 0 && d.in_pix_stride == d.c_in &&
          d.h_in == d.h_out && d.w_in == d.w_out && d.relu == 0 && d.out_dtype == METRO_F16 && d.in_dtype == METRO_F16))
        return false;
    const bool res_plain = d.res_stride == 1 && d.res_offset == 0 && d.res_h == d.h_out && d.res_w == d.w_out;
    // sub-sampled shortcut of the stride-2 units: every addressed shortcut pixel must exist
    const bool res_sub = d.res_stride == 2 && d.res_offset >= 0 && d.res_offset + 2 * (d.h_out - 1) < d.res_h &&
                         d.res_offset + 2 * (d.w_out - 1) < d.res_w;
    if (d.has_residual && !res_plain && !(mode == 0 && res_sub && d.c_in <= 128)) return false;
    // built combinations: prologue without shortcut (projection shortcut, pair) / shortcut without prologue (conv3)
    if (mode == 1) {
        // block1's pair (64 -> 256 + 64) and, round 5, block2's (256 -> 512 + 128: all 640 weight rows register-resident in one block)
        static const int pair256 = pw_env_int("METRO_PW_PAIR256", 1);
        // (under the test switch metro_conv_b1_form(1) the 256-channel pair stays on the ring kernel it replaced: same bits, tested)
        return ((d.c_in == 64 && d.c_out == 320) || (pair256 && !classic_forms_forced() && d.c_in == 256 && d.c_out == 640)) &&
               d.has_prologue && !d.has_residual;
    }
    if (mode == 2) {
        static const int next128 = pw_env_int("METRO_PW_NEXT128", 1);
        return ((d.c_in == 64 && d.c_out == 256) || (next128 && d.c_in == 128 && d.c_out == 512 && res_plain)) && !d.has_prologue && d.has_residual;
    }
    if (mode == 3) return d.c_in == 64 && d.c_out == 256 && !d.has_prologue && !d.has_residual;    // + in-launch projection shortcut
    if (mode == 4) {           // + the residual rebuilt from the previous unit's conv2 output (and optionally kept on chip)
        const int hw = d.h_out * d.w_out;
        return d.c_in == 64 && d.c_out == 256 && !d.has_prologue && !d.has_residual && hw % 64 == 0 && d.w_out >= 16 &&
               (d.w_out & (d.w_out - 1)) == 0;
    }
    if (d.c_in == 64 && d.c_out == 256) return (d.has_prologue != 0) != (d.has_residual != 0);
    // conv3 (+ shortcut) of blocks 2-4: c_out = 4 * c_in in 256-channel slabs (METRO_PW_MAXK caps c_in for A/B runs)
    static const int maxk = pw_env_int("METRO_PW_MAXK", 512);
    return (d.c_in == 128 || d.c_in == 256 || d.c_in == 512) && d.c_in <= maxk && d.c_out == 4 * d.c_in &&
           !d.has_prologue && d.has_residual;
}

template <int K, int WM, bool PRO, bool RES, int MODE2, bool RSUB = false, bool PSC = false, int CB = 256, bool REB = false, int OUTM = 0>
static int launch_pw(Pw64Args a, hipStream_t stream) {
    if (note_kernel("conv_pw64<k%d,wm%d%s%s%s%s%s%s%s%s>", K, WM, CB == 512 ? ",cb512" : "", PRO ? ",pro" : "", RES ? ",res" : "",
                    MODE2 == 1 ? ",pair" : MODE2 == 2 ? ",next" : "", RSUB ? ",ressub" : "", PSC ? ",projsc" : "", REB ? ",rebuild" : "",
                    OUTM == 1 ? ",noout" : OUTM == 2 ? ",subout" : ""))
        return METRO_OK;
    auto kern = conv_pw64_kernel<K, WM, PRO, RES, MODE2, RSUB, PSC, CB, REB, OUTM>;
    constexpr int lds = pw::lds_bytes<K, WM, RES, MODE2, PSC, CB, REB>();
    a.n_tiles = (a.m_total + pw::Lay<K, WM, CB>::TN - 1) / pw::Lay<K, WM, CB>::TN;
    static PerDeviceInt cap;
    int grid_cap = 0;
    if (const int st = ensure_dyn_lds_and_grid_cap(reinterpret_cast<const void*>(kern), pw::NT, lds, cap, "conv_pw64",
                                                   pw_env_int("METRO_PW64_BPC", 0) /* blocks per CU, 0 = what fits */, &grid_cap))
        return st;
    const int halves = a.c_out / CB;
    int grid = a.n_tiles * halves < grid_cap ? a.n_tiles * halves : grid_cap;
    grid -= grid % halves;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(pw::NT), lds, stream, a);
    return launch_status("conv_pw64");
}

int launch_conv_pw64(const MetroConvDesc& d, const void* in, const void* w, const float* bias, const void* ps,
                     const void* pb, const void* res, void* out, hipStream_t stream, const ConvSplit* split,
                     const ConvFuse2* f2, const ConvProjSc* psc, const ConvRebuild* rb) {
    const bool proj = psc != nullptr && psc->x != nullptr;
    const bool reb = rb != nullptr && rb->t2_prev != nullptr;
    const int outm = rb != nullptr ? rb->out_mode : 0;
    const int mode = (f2 != nullptr && f2->w2 != nullptr) ? (proj ? (reb ? 4 : 3) : 2) : (split != nullptr && split->split > 0) ? 1 : 0;
    if (proj && mode < 3) { set_error("conv_pw64: the in-launch projection shortcut exists for conv3 + next conv1 only"); return METRO_ERR_INVALID_ARG; }
    if ((reb || outm != 0) && mode < 3) { set_error("conv_pw64: rebuilt residual / on-chip output exist on top of the in-launch projection shortcut only"); return METRO_ERR_INVALID_ARG; }
    if (outm < 0 || outm > 2 || (outm == 2 && !(reb && rb->out_sub != nullptr && rb->sub_off >= 0 && rb->sub_off <= 1 && rb->h_sub > 0 && rb->w_sub > 0)) ||
        (outm == 0 && out == nullptr)) {
        set_error("conv_pw64: bad output mode %d (2 = sub-sampled copy: needs the rebuilt residual, out_sub, sub_off 0|1 and the sub-sampled map size)", outm);
        return METRO_ERR_INVALID_ARG;
    }
    if (!conv_pw64_supported(d, mode) || (mode == 1 && !(split->relu2 == 1 && ((d.c_in == 64 && split->split == 256 && split->c_out2 == 64) ||
                                                                      (d.c_in == 256 && split->split == 512 && split->c_out2 == 128)))) ||
        (mode >= 2 && f2->c2 != (d.c_in == 128 ? 128 : 64))) {
        set_error("conv_pw64: unsupported layer");
        return METRO_ERR_INVALID_ARG;
    }
    // the launches whose sum stays on chip (or is rebuilt): the producer / consumer form, unless the classic one is asked for
    if (mode >= 3 && rb != nullptr && !rb->classic && (reb || outm == 1) && conv_b1_chain_preferred() && conv_pw64_supported(d, 4))
        return launch_conv_b1_chain(d, in, w, bias, out, stream, *f2, *psc, *rb);
    Pw64Args a;
    a.in = static_cast<const half_t*>(in);
    a.w = static_cast<const half_t*>(w);
    a.bias = bias;
    a.pro_scale = static_cast<const half_t*>(ps);
    a.pro_shift = static_cast<const half_t*>(pb);
    a.residual = d.has_residual ? static_cast<const half_t*>(res) : nullptr;
    a.out = static_cast<half_t*>(out);
    a.w2 = nullptr; a.bias2 = nullptr; a.scale2 = nullptr; a.shift2 = nullptr; a.out2 = nullptr;
    a.x_sc = nullptr; a.w_sc = nullptr; a.bias_sc = nullptr;
    a.in_b = nullptr; a.w_b = nullptr; a.bias_b = nullptr; a.out_sub = nullptr; a.sub_off = 0; a.h_sub = 0; a.w_sub = 0; a.lw_out = 0;
    a.m_total = d.n * d.h_out * d.w_out;
    a.n_tiles = 0;
    a.c_out = mode == 1 ? split->split : d.c_out;
    a.res_stride = d.res_stride; a.res_off = d.res_offset; a.res_h = d.res_h; a.res_w = d.res_w;
    a.h_out = d.h_out; a.w_out = d.w_out;
    const bool rsub = d.has_residual && d.res_stride == 2;
    if (mode == 1) {
        a.w2 = a.w + (size_t)split->split * d.c_in; a.bias2 = bias + split->split; a.out2 = static_cast<half_t*>(split->out2);
        if (d.c_in == 256) return launch_pw<256, 8, true, false, 1, false, false, 512>(a, stream);
        return launch_pw<64, 4, true, false, 1>(a, stream);
    }
    if (mode >= 2) {
        a.w2 = static_cast<const half_t*>(f2->w2); a.bias2 = f2->bias2;
        a.scale2 = static_cast<const half_t*>(f2->scale2); a.shift2 = static_cast<const half_t*>(f2->shift2);
        a.out2 = static_cast<half_t*>(f2->out2);
        if (mode >= 3) {
            a.x_sc = static_cast<const half_t*>(psc->x); a.w_sc = static_cast<const half_t*>(psc->w_sc); a.bias_sc = psc->bias_sc;
            a.pro_scale = static_cast<const half_t*>(psc->pro_scale); a.pro_shift = static_cast<const half_t*>(psc->pro_shift);
            if (reb) {
                a.in_b = static_cast<const half_t*>(rb->t2_prev); a.w_b = static_cast<const half_t*>(rb->w3_prev); a.bias_b = rb->bias3_prev;
                if (a.w_b == nullptr || a.bias_b == nullptr) { set_error("conv_pw64: rebuilt residual without the previous unit's conv3 parameters"); return METRO_ERR_INVALID_ARG; }
                if (outm == 2) {
                    a.out_sub = static_cast<half_t*>(rb->out_sub); a.sub_off = rb->sub_off; a.h_sub = rb->h_sub; a.w_sub = rb->w_sub;
                    while ((1 << a.lw_out) < d.w_out) ++a.lw_out;
                    return launch_pw<64, 4, false, false, 2, false, true, 256, true, 2>(a, stream);
                }
                if (outm == 1) { set_error("conv_pw64: a rebuilt residual whose sum is neither stored nor sub-sampled has no consumer"); return METRO_ERR_INVALID_ARG; }
                return launch_pw<64, 4, false, false, 2, false, true, 256, true, 0>(a, stream);
            }
            if (outm == 1) return launch_pw<64, 4, false, false, 2, false, true, 256, false, 1>(a, stream);
            return launch_pw<64, 4, false, false, 2, false, true>(a, stream);
        }
        if (d.c_in == 128) return launch_pw<128, 8, false, true, 2, false, false, 512>(a, stream);
        return launch_pw<64, 4, false, true, 2>(a, stream);
    }
    if (d.c_in == 512) return launch_pw<512, 8, false, true, 0>(a, stream);
    if (d.c_in == 256) return launch_pw<256, 8, false, true, 0>(a, stream);
    if (d.c_in == 128) return rsub ? launch_pw<128, 4, false, true, 0, true>(a, stream) : launch_pw<128, 4, false, true, 0>(a, stream);
    if (d.has_prologue) return launch_pw<64, 4, true, false, 0>(a, stream);
    return rsub ? launch_pw<64, 4, false, true, 0, true>(a, stream) : launch_pw<64, 4, false, true, 0>(a, stream);
}

}  // namespace metro
