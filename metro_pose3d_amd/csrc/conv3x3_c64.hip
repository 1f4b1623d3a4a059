// Persistent, weight-resident 3x3 convolution for the 64 -> 64 channel layers (conv2 of block1's stride-1 units:
// reference resnet_v2.py:130-132 through resnet_utils.conv2d_same, SAME padding, rate 1; folded BN + ReLU).
//
// The tap-reuse kernel (conv3x3_f16_slab.hip) runs these layers as 1024 (batch 64) short-lived blocks: each stages a
// 48 KB activation slab AND all 72 KB of weights for 72 MFMAs per wave, then transposes and stores -- its phases are
// serial and latency bound (38 us where the bytes need 12 and the MFMAs 16; the matrix pipe is ~11 % busy).  Here
//   * ONE block per CU stays resident and walks a CONTIGUOUS range of 128-pixel tiles (two rows of a 64-wide map):
//     the 9 x [64 x 64] weight images (72 KB) are DMA'd into LDS once per block;
//   * the activation slab of tile i+1 (tile + one row of halo above and below, 32 KB) is LDS-DMA'd into the other
//     slab buffer while tile i is computed -- consecutive tiles share their halo rows through L2;
//   * a tile is 36 MFMAs per wave (9 taps x 4 k steps, one 32 x 32 tile per wave) with NO barrier inside; taps that
//     leave the image read a zero row (TF SAME zero padding, resnet_utils.py:120-123);
//   * the epilogue (+bias, ReLU, fp16) goes through its own LDS staging tile and leaves as full 128-byte rows;
//   * ordering: one counted s_waitcnt vmcnt(2) per tile (the only VMEM operations younger than the next slab's DMA
//     are this wave's 2 row stores) + two raw s_barrier.
// Arithmetic = the tap-reuse kernel's: fp16 operands, fp32 accumulate over (tap, channel), fp16(relu(acc + bias)).
//
// PRE1 (round 3): conv1 of the SAME unit in front of it, for block1/unit_1 whose input has only 64 channels (the pooled
// stem output): the slab that is DMA'd is the raw unit input x, and every wave turns its 32 slab rows into
// t1 = fp16(relu(W1 . fp16(relu(x * scale + shift)) + b1)) IN PLACE (pre-activation BN + ReLU, conv1 with its folded BN + ReLU:
// reference resnet_v2.py:119,127-128) -- 8 more MFMAs per wave and tile, W1 in registers as A fragments; the halo rows
// are recomputed by both tiles that share them.  The 3x3 then runs on t1 exactly as before (taps outside the image still
// read the zero area: conv2d_same pads t1, not x).  t1 never exists in HBM: the unit loses a launch (projection
// shortcut + conv1 pair -> nothing; the shortcut moves into the conv3 launch, conv_pw64.hip PSC) and 1152 of its 2304
// bytes per pixel.  Same arithmetic as the separate launches: one fp16 rounding per tensor (oracle/f16emu.py unit_conv1).
#include "metro_common.h"

namespace metro {

typedef _Float16 half_t;
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

__device__ __attribute__((aligned(16))) unsigned int g_zero_page_c64[4];   // zero-initialised

namespace c64 {
constexpr int C = 64, TN = 128, NW = 8, NT = 512;
constexpr int W_BYTES = 9 * C * 128;                     // 72 KiB: 9 taps x [64 cout][64 k] fp16
constexpr int SLAB_ROWS = 256;                           // TN + 2 * halo, halo <= 64
constexpr int SLAB_BYTES = SLAB_ROWS * 128;              // 32 KiB
constexpr int SI = SLAB_BYTES / 1024 / NW;               // slab DMA instructions per wave per tile: 4
constexpr int WI = W_BYTES / 1024 / NW;                  // weight DMA instructions per wave (once): 9
constexpr int W_OFF = 0;
constexpr int SLAB_OFF = W_BYTES;
constexpr int ZERO_OFF = SLAB_OFF + 2 * SLAB_BYTES;      // 256-byte zero area (256-byte aligned)
constexpr int OUT_OFF = ZERO_OFF + 256;
constexpr int OUT_ROW = C * 2 + 16;
constexpr int LDS_BYTES = OUT_OFF + TN * OUT_ROW;        // 157,952 B
constexpr int STORES = TN * (C / 8) / NT;                // row-wise 16-byte stores per thread per tile: 2
static_assert(LDS_BYTES <= 160 * 1024 && ZERO_OFF % 256 == 0, "LDS budget / zero-area alignment");
}  // namespace c64

typedef __attribute__((address_space(3))) void c64_lds_void_t;

__device__ __forceinline__ void c64_dma16(const void* gsrc, unsigned lds_addr) {
    asm volatile(
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, off"
        :
        : "v"(gsrc), "s"(lds_addr));
}

struct C64Args {
    const half_t* in;      // [m_total][64]
    const half_t* w;       // [64][3][3][64]
    const float* bias;     // [64]
    half_t* out;           // [m_total][64]
    int m_total, h, w_map, relu, n_tiles, tiles_per_block;
    // PRE1: conv1 (1x1, 64 -> 64, folded BN + ReLU) on the pre-activated slab in front of the 3x3
    const half_t* w1;      // [64][64]
    const float* bias1;    // [64]
    const half_t* pro_scale;   // [64] pre-activation BN of the unit
    const half_t* pro_shift;
    half_t* t1_dump;       // layer dumps / tests only (NULL in the product path): conv1's output [m_total][64] as the 3x3 reads it
};

template <bool PRE1>
__global__ __launch_bounds__(c64::NT) void conv3x3_c64_kernel(C64Args a) {
    using namespace c64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;              // 32-cout tile, 32-pixel tile
    const int frag_row = lane & 31, frag_half = lane >> 5;
    const int t_begin = blockIdx.x * a.tiles_per_block;
    int t_end = t_begin + a.tiles_per_block;
    if (t_end > a.n_tiles) t_end = a.n_tiles;
    if (t_begin >= t_end) return;
    const half_t* zero = reinterpret_cast<const half_t*>(g_zero_page_c64);
    const unsigned smem_base = (unsigned)(size_t)(c64_lds_void_t*)smem;
    const int halo = a.w_map;                             // one map row above and below
    const int hw = a.h * a.w_map;

    if (tid < 16) reinterpret_cast<uint4*>(smem + ZERO_OFF)[tid] = make_uint4(0, 0, 0, 0);

    // ---- weights: once per block.  Image of tap t: rows = cout, 128-byte rows of 64 k, chunk-swizzled ----
    const int lrow = lane >> 3, lch = lane & 7;
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const int grp = i * NW + wave;                    // 72 groups of 8 rows: tap = grp / 8, cout rows (grp % 8) * 8 ..
        const int tap = grp >> 3, row = (grp & 7) * 8 + lrow;
        const half_t* src = a.w + (size_t)row * (9 * C) + tap * C + ((lch ^ ((row >> 1) & 7)) * 8);
        c64_dma16(src, __builtin_amdgcn_readfirstlane(smem_base + W_OFF + grp * 1024));
    }
    // ---- slab DMA of a tile: flattened pixel rows [m0 - halo, m0 + TN + halo) (zero page outside the tensor) ----
    auto issue_slab = [&](int tile, int buf) {
        const int m0 = tile * TN;
#pragma unroll
        for (int i = 0; i < SI; ++i) {
            const int srow = (i * NW + wave) * 8 + lrow;
            const int g = m0 - halo + srow;
            const bool ok = g >= 0 && g < a.m_total && srow < TN + 2 * halo;
            const half_t* src = ok ? a.in + (size_t)g * C + ((lch ^ ((srow >> 1) & 7)) * 8) : zero;
            c64_dma16(src, __builtin_amdgcn_readfirstlane(smem_base + SLAB_OFF + buf * SLAB_BYTES + (i * NW + wave) * 1024));
        }
    };
    issue_slab(t_begin, 0);

    // per-lane A (weight) fragment base: row * 128 + swizzle bits; tap adds 8192, k step kk is an XOR with kk << 5
    const int arow = wm * 32 + frag_row;
    const int a_base = W_OFF + arow * 128 + ((frag_half ^ ((arow >> 1) & 7)) << 4);
    float bias_v[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const floatx4 bv = *reinterpret_cast<const floatx4*>(a.bias + wm * 32 + 8 * q + 4 * frag_half);
#pragma unroll
        for (int e = 0; e < 4; ++e) bias_v[q][e] = bv[e];
    }
    // PRE1: W1 as A fragments (both 32-cout tiles x 4 k steps), its bias in the accumulator layout, the pre-activation per k step
    half8_t w1f[2][4], ps1[4], pb1[4];
    float bias1_v[2][4][4];
    if constexpr (PRE1) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                w1f[i][kk] = *reinterpret_cast<const half8_t*>(a.w1 + (size_t)(i * 32 + frag_row) * C + kk * 16 + frag_half * 8);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const floatx4 bv = *reinterpret_cast<const floatx4*>(a.bias1 + i * 32 + 8 * q + 4 * frag_half);
#pragma unroll
                for (int e = 0; e < 4; ++e) bias1_v[i][q][e] = bv[e];
            }
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            ps1[kk] = *reinterpret_cast<const half8_t*>(a.pro_scale + kk * 16 + frag_half * 8);
            pb1[kk] = *reinterpret_cast<const half8_t*>(a.pro_shift + kk * 16 + frag_half * 8);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");     // weights + first slab + zero area

    for (int t = t_begin; t < t_end; ++t) {
        const int buf = (t - t_begin) & 1;
        if (t != t_begin) {
            // the next slab was requested before this wave's STORES row stores of the previous tile: everything older has landed
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(STORES) : "memory");
        }
        if (t + 1 < t_end) issue_slab(t + 1, buf ^ 1);     // other buffer: last read before the barrier above

        if constexpr (PRE1) {
            // ---- conv1 in place: wave w owns slab rows [32 w, 32 w + 32): all 64 input channels of a row are read before
            //      its 64 output channels are written, nobody else touches these rows before the barrier below ----
            char* sp = smem + SLAB_OFF + buf * SLAB_BYTES;
            const int srow = wave * 32 + frag_row;
            const int r_base = srow * 128 + ((frag_half ^ ((srow >> 1) & 7)) << 4);
            floatx16 a1[2];
#pragma unroll
            for (int e = 0; e < 16; ++e) { a1[0][e] = 0.f; a1[1][e] = 0.f; }
            const half8_t z = {};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                half8_t bf = *reinterpret_cast<const half8_t*>(sp + (r_base ^ (kk << 5)));
                bf = __builtin_elementwise_max(bf * ps1[kk] + pb1[kk], z);            // fp16 FMA + ReLU, one rounding
                a1[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1f[0][kk], bf, a1[0], 0, 0, 0);
                a1[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1f[1][kk], bf, a1[1], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    half4_t hv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) hv[e] = (half_t)fmaxf(a1[i][4 * q + e] + bias1_v[i][q][e], 0.f);
                    // channels i*32 + 8q + 4 half ..+3 = 16-byte chunk (4i + q), byte 8 * half inside it
                    *reinterpret_cast<half4_t*>(sp + srow * 128 + ((((4 * i + q) ^ ((srow >> 1) & 7)) << 4) | (frag_half << 3))) = hv;
                    if (a.t1_dump != nullptr && srow >= halo && srow < halo + TN)      // the tile's own rows, once
                        *reinterpret_cast<half4_t*>(a.t1_dump + (size_t)(t * TN + srow - halo) * C + i * 32 + 8 * q + 4 * frag_half) = hv;
                }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        // ---- per-lane tap rows of this tile (slab row, or the zero area for taps outside the image) ----
        const int m0 = t * TN;
        const int tl = wn * 32 + frag_row;                 // tile-local pixel
        const int rem = (m0 + tl) % hw;
        const int py = rem / a.w_map, px = rem - py * a.w_map;
        floatx16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        const int sl = SLAB_OFF + buf * SLAB_BYTES;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dr = tap / 3 - 1, ds = tap % 3 - 1;
            const bool ok = (unsigned)(py + dr) < (unsigned)a.h && (unsigned)(px + ds) < (unsigned)a.w_map;
            const int srow = halo + tl + dr * a.w_map + ds;
            const int b_base = ok ? sl + srow * 128 + ((frag_half ^ ((srow >> 1) & 7)) << 4) : ZERO_OFF;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const half8_t af = *reinterpret_cast<const half8_t*>(smem + tap * 8192 + (a_base ^ (kk << 5)));
                const half8_t bf = *reinterpret_cast<const half8_t*>(smem + (b_base ^ (kk << 5)));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc, 0, 0, 0);
            }
        }
        // ---- epilogue: (+bias, ReLU) -> staging [pixel][cout] fp16 -> full 128-byte rows ----
        char* ol = smem + OUT_OFF;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            half4_t hv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[4 * q + e] + bias_v[q][e];
                if (a.relu) v = fmaxf(v, 0.f);
                hv[e] = (half_t)v;
            }
            *reinterpret_cast<half4_t*>(ol + tl * OUT_ROW + (wm * 32 + 8 * q + 4 * frag_half) * 2) = hv;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int r = 0; r < STORES; ++r) {
            const int idx = tid + r * NT;
            const int prow = idx >> 3, ch = idx & 7;
            const uint4 v = *reinterpret_cast<const uint4*>(ol + prow * OUT_ROW + ch * 16);
            store_out16<1>(a.out + (size_t)(m0 + prow) * C + ch * 8, v);
        }
    }
}

// 64 -> 64 channels, 3x3, stride 1, rate 1, SAME padding, dense fp16 NHWC, whole 128-pixel tiles of whole map rows
bool conv3x3_c64_supported(const MetroConvDesc& d) {
    static const int enabled = tuning_knob("METRO_CONV_C64", 1);
    if (!enabled) return false;
    if (!(d.kh == 3 && d.kw == 3 && d.stride == 1 && d.dilation == 1 && d.pad_top == 1 && d.pad_left == 1 &&
          d.h_in == d.h_out && d.w_in == d.w_out && d.c_in == 64 && d.c_out == 64 && d.in_pix_stride == 64 &&
          !d.has_prologue && !d.has_residual && d.in_dtype == METRO_F16 && d.out_dtype == METRO_F16))
        return false;
    const long m = (long)d.n * d.h_out * d.w_out;
    return d.w_out <= 64 && c64::TN % d.w_out == 0 && (d.h_out * d.w_out) % c64::TN == 0 && m >= 4 * c64::TN;
}

int launch_conv3x3_c64(const MetroConvDesc& d, const void* in, const void* w, const float* bias, void* out, hipStream_t stream,
                       const ConvPre1* pre1) {
    if (!conv3x3_c64_supported(d)) { set_error("conv3x3_c64: unsupported layer"); return METRO_ERR_UNSUPPORTED; }
    const bool p1 = pre1 != nullptr && pre1->w1 != nullptr;
    if (note_kernel(p1 ? "conv3x3_c64<pre1>" : "conv3x3_c64")) return METRO_OK;
    C64Args a;
    a.in = static_cast<const half_t*>(in); a.w = static_cast<const half_t*>(w); a.bias = bias; a.out = static_cast<half_t*>(out);
    a.m_total = d.n * d.h_out * d.w_out; a.h = d.h_out; a.w_map = d.w_out; a.relu = d.relu;
    a.n_tiles = a.m_total / c64::TN;
    a.w1 = p1 ? static_cast<const half_t*>(pre1->w1) : nullptr; a.bias1 = p1 ? pre1->bias1 : nullptr;
    a.pro_scale = p1 ? static_cast<const half_t*>(pre1->pro_scale) : nullptr;
    a.pro_shift = p1 ? static_cast<const half_t*>(pre1->pro_shift) : nullptr;
    a.t1_dump = p1 ? static_cast<half_t*>(pre1->t1_dump) : nullptr;
    auto kern = p1 ? conv3x3_c64_kernel<true> : conv3x3_c64_kernel<false>;
    static PerDeviceInt cap[2];
    int grid_cap = 0;
    if (const int st = ensure_dyn_lds_and_grid_cap(reinterpret_cast<const void*>(kern), c64::NT, c64::LDS_BYTES, cap[p1 ? 1 : 0],
                                                   "conv3x3_c64", 1, &grid_cap))
        return st;
    // contiguous tile ranges per block: consecutive tiles share their halo rows through the L2 of one XCD
    a.tiles_per_block = (a.n_tiles + grid_cap - 1) / grid_cap;
    const int grid = (a.n_tiles + a.tiles_per_block - 1) / a.tiles_per_block;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(c64::NT), c64::LDS_BYTES, stream, a);
    return launch_status("conv3x3_c64");
}

}  // namespace metro
