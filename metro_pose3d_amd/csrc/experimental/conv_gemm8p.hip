// 1x1 convolution as a 256 x 256 x 64 GEMM with an 8-PHASE, two-wave-group schedule (gfx950).
//
// conv_igemm_f16_dma.hip runs every wave of a block in lock step: all eight waves read fragments, then all
// eight issue MFMAs, one barrier per K step -- the matrix pipe idles while the LDS is read and vice versa
// (PMC: MFMA pipe 12-30 % busy on the 1x1 layers, SQ_WAIT_ANY 30-60 %).  Here, for the deep-K 1x1 layers that
// have >= 256 tiles of 256 cout x 256 pixels (conv1 / projection shortcut / shortcut+conv1 pair of blocks 3-4):
//   * a K tile (64 channels) is computed in FOUR phases, one 64-cout x 32-pixel quadrant of the wave's 128 x 64
//     output tile per phase: [ds_read the quadrant's fragments | issue one half-tile of LDS-DMA | counted wait]
//     -> s_barrier -> 8 x v_mfma_f32_32x32x16_f16 -> s_barrier.  8 phases = 2 K tiles per loop iteration, so the
//     two LDS buffers have compile-time addresses;
//   * waves 4-7 execute ONE extra s_barrier before the loop (and waves 0-3 one after it), so the two waves that
//     share a SIMD are permanently one barrier apart: while one is in its MFMA interval the other reads LDS and
//     issues DMA, and the matrix pipe alternates between them (s_setprio 1 around the MFMA cluster);
//   * the next K tile is staged one half-tile (128 rows x 128 B = 2 DMA instructions per wave) per phase, four
//     phases ahead of its wait; s_waitcnt vmcnt(6) -- never 0 in the loop: three half-tiles (48 KiB per CU) stay in
//     flight -- retires a half-tile ONE phase before it is read (a barrier more than lock step needs: the reading group may be a barrier behind the
//     issuing one); a slot is re-staged three or four phases after its last ds_read;
//   * operand images, swizzle, zero page, pre-activation prologue, bias/ReLU/shortcut epilogue through LDS and
//     the fused-pair output routing are those of conv_igemm_f16_dma.hip (same arithmetic: fp32 accumulate,
//     fp16(conv + bias), fp16 shortcut add -- reference resnet_v2.py:119-138 under tfu.py:426-440).
#include <type_traits>

#include "../metro_common.h"
#include "metro_experimental.h"

namespace metro {

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

namespace g8 {
constexpr int TM = 256, TN = 256, BK = 64, NT = 512;
constexpr int ROW_BYTES = BK * 2;                  // 128
constexpr int OPER_BYTES = 256 * ROW_BYTES;        // one operand image of a K tile: 32 KiB
// LDS ring: [A buf0 | A buf1 | B buf0 | B buf1]: the buffer index is a 32 KiB IMMEDIATE on every ds_read
// (16-bit offset field), so both K-tile buffers share their per-lane address registers
constexpr int BUF_STRIDE = OPER_BYTES;
constexpr int B_BASE = 2 * OPER_BYTES;
constexpr int RING_BYTES = 4 * OPER_BYTES;         // 128 KiB
constexpr int OUT_ROW_BYTES = TM * 2 + 16;
constexpr int OUT_BYTES = TN * OUT_ROW_BYTES;      // 135168
constexpr int MAIN_BYTES = OUT_BYTES > RING_BYTES ? OUT_BYTES : RING_BYTES;
constexpr int PRO_BYTES = 2 * 2048 * 2;            // scale | shift, c_in <= 2048
}  // namespace g8

__device__ __forceinline__ int g8_swz(int row) { return (row >> 1) & 7; }

typedef __attribute__((address_space(3))) void g8_lds_void_t;

// one LDS-DMA wave-instruction (64 lanes x 16 B, lane l lands at lds_addr + 16*l); inline asm so that hipcc does
// not drain it with vmcnt(0) before every ds_read (see conv_igemm_f16_dma.hip).  Source = wave-uniform base
// (SGPR pair, advanced per K tile by scalar adds) + per-lane 32-bit byte offset: 1 VGPR per instruction stream.
__device__ __forceinline__ void g8_dma16(const void* sbase, unsigned voff, unsigned lds_addr) {
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %1"
        :
        : "v"(voff), "s"(sbase), "s"(lds_addr));
}
template <int N>
__device__ __forceinline__ void g8_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void g8_barrier() { asm volatile("s_barrier" ::: "memory"); }
__device__ __forceinline__ void g8_wait_lgkm_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <bool PROLOGUE>
__global__ __launch_bounds__(g8::NT) void conv_gemm8p_kernel(
    ConvArgs a, const half_t* __restrict__ in, const half_t* __restrict__ w, const float* __restrict__ bias,
    const half_t* __restrict__ pro_scale, const half_t* __restrict__ pro_shift, const half_t* __restrict__ residual,
    half_t* __restrict__ out, half_t* __restrict__ out2, int tiles_m) {
    using namespace g8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2;            // 128-cout half of the tile; ALSO the wave group (0 leads, 1 is a barrier behind)
    const int wc = wave & 3;             // 64-pixel quarter

    // XCD-aware (bijective) block -> tile map: the blocks of one XCD share pixel tiles in their L2
    const int nblk = gridDim.x;
    int lid;
    {
        const int b = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = b & 7, idx = b >> 3;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = lid / tiles_m;
    const int tile_m = lid % tiles_m;
    const int m0 = tile_n * TN;
    const int n0 = tile_m * TM;
    const int K = a.c_in;
    const int nk = K / BK;
    const unsigned smem_base = (unsigned)(size_t)(g8_lds_void_t*)smem;
    half_t* pro_lds = reinterpret_cast<half_t*>(smem + MAIN_BYTES);

    // ---- LDS-DMA sources.  Half-tiles (what ONE phase stages = what every wave reads in one later phase):
    //   HA0 = cout rows of m-tiles {0,1} of both wave rows: [0,64) u [128,192);   HA1 = [64,128) u [192,256)
    //   HB0 = pixel rows of n-tile 0 of the four wave columns: [64c, 64c+32);     HB1 = [64c+32, 64c+64)
    // A half-tile is 16 groups of 8 rows; wave w issues two of them.  LDS row = tile row (natural order).
    // The launcher guarantees whole tiles (c_out % 256 == 0, pixels % 256 == 0): no zero page, uniform strides.
    const int lrow = lane >> 3, lch = lane & 7;
    unsigned voff[8];              // per-lane byte offsets from the tile's operand base: HA0 x2, HA1 x2, HB0 x2, HB1 x2
    unsigned ldsoff[8];            // byte offset of the 8-row group inside the ring (buffer 0)
#pragma unroll
    for (int h = 0; h < 2; ++h) {          // A half-tiles
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int grp = (i * 16) + h * 8 + wave;            // rows [8*grp, 8*grp+8): i = wave row, h = half
            const int row = grp * 8 + lrow;
            const int e = h * 2 + i;
            voff[e] = (unsigned)(row * K + ((lch ^ g8_swz(row)) * 8)) * 2u;
            ldsoff[e] = grp * 8 * ROW_BYTES;
        }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {          // B half-tiles
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = i * 8 + wave;                          // 0..15: (wave column c = q >> 2, 8-row group g = q & 3)
            const int grp = (q >> 2) * 8 + h * 4 + (q & 3);      // rows 64c + 32h + 8g ..
            const int row = grp * 8 + lrow;
            const int e = 4 + h * 2 + i;
            voff[e] = (unsigned)(row * K + ((lch ^ g8_swz(row)) * 8)) * 2u;
            ldsoff[e] = B_BASE + grp * 8 * ROW_BYTES;
        }
    }
    const half_t* wbase = w + (size_t)n0 * K;      // wave-uniform operand bases of this tile
    const half_t* xbase = in + (size_t)m0 * K;
    // stage half-tile `which` (0 HA0, 1 HA1, 2 HB0, 3 HB1) of K tile `kt` into buffer `buf`
    auto stage = [&](int which, int buf, int kt) {
        const half_t* sb = (which < 2 ? wbase : xbase) + kt * BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = which * 2 + i;
            g8_dma16(sb, voff[e], __builtin_amdgcn_readfirstlane(smem_base + buf * BUF_STRIDE + ldsoff[e]));
        }
    };

    floatx16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int frag_row = lane & 31, frag_half = lane >> 5;
    // per-lane fragment base addresses (buffer 0): row * 128 + swizzle bits; k step kk and the buffer are added as
    // XOR / immediate at the read: chunk (kk*2 + half) ^ swz(row) == ((half ^ swz) | (swz & 6)) ^ (kk << 1)
    unsigned a_base[4], b_base[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = wr * 128 + i * 32 + frag_row;
        a_base[i] = row * ROW_BYTES + ((frag_half ^ g8_swz(row)) << 4);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = wc * 64 + j * 32 + frag_row;
        b_base[j] = B_BASE + row * ROW_BYTES + ((frag_half ^ g8_swz(row)) << 4);
    }

    // ---- prologue: K tile 0 (all four half-tiles), pre-activation table, then the stagger ------------------
    stage(0, 0, 0); stage(2, 0, 0); stage(3, 0, 0); stage(1, 0, 0);
    stage(2, 1, nk > 1 ? 1 : 0);          // HB0 of tile 1: the loop stages every half-tile four phases ahead of its wait
    if (PROLOGUE) {
        for (int c = tid * 8; c < K; c += NT * 8) {
            *reinterpret_cast<uint4*>(pro_lds + c) = *reinterpret_cast<const uint4*>(pro_scale + c);
            *reinterpret_cast<uint4*>(pro_lds + 2048 + c) = *reinterpret_cast<const uint4*>(pro_shift + c);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // tile 0 landed, table visible
    if (wr == 1) g8_barrier();                                                  // waves 4-7 run one barrier behind

    half8_t af[2][4], bf0[4], bf1[4];
    auto load_a = [&](const char* buf, int half) {           // m-tiles {2*half, 2*half+1}, 4 k steps
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                af[i][kk] = *reinterpret_cast<const half8_t*>(buf + (a_base[2 * half + i] ^ (kk << 5)));
    };
    auto load_b = [&](const char* buf, int j, half8_t (&bf)[4], int k0) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            bf[kk] = *reinterpret_cast<const half8_t*>(buf + (b_base[j] ^ (kk << 5)));
        if (PROLOGUE) {
            // pre-activation BN + ReLU on the pixel fragment (fp16 FMA, one rounding: resnet_v2.py:119)
            const half8_t z = {};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const half8_t sc = *reinterpret_cast<const half8_t*>(pro_lds + k0 + kk * 16 + frag_half * 8);
                const half8_t sh = *reinterpret_cast<const half8_t*>(pro_lds + 2048 + k0 + kk * 16 + frag_half * 8);
                bf[kk] = __builtin_elementwise_max(bf[kk] * sc + sh, z);
            }
            // keep this VALU work in the load interval: volatile asm statements stay ordered with the barrier asm,
            // so the values must exist before it (otherwise hipcc sinks half of it behind the barrier, in front of
            // the MFMA cluster of the compute interval)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) asm volatile("" : "+v"(bf[kk]));
        }
    };
    auto mma = [&](int half, int j, const half8_t (&bf)[4]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                acc[2 * half + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][kk], bf[kk], acc[2 * half + i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    // One K tile = four phases.  The next tile is staged unconditionally: past the end the LAST tile is staged again
    // (valid memory, 64 KiB of wasted L2 reads per block) so that the loop body, its wait counts and the register
    // allocation are the same for every tile; the launcher guarantees an even number of K tiles.
    auto ktile = [&](auto buf_c, int kt) {
        constexpr int BUF = decltype(buf_c)::value;
        const char* buf = smem + BUF * BUF_STRIDE;
        const int k0 = kt * BK;
        const int kn = kt + 1 < nk ? kt + 1 : nk - 1;
        const int kn2 = kt + 2 < nk ? kt + 2 : nk - 1;
        // ---- phase 1: quadrant (m 0-1, n 0) ----
        load_a(buf, 0);
        load_b(buf, 0, bf0, k0);
        stage(0, BUF ^ 1, kn);            // HA0 of tile t+1 (slot last read in phase 1 of tile t-1)
        g8_wait_vm<6>();                  // retires HB1 of THIS tile (read in phase 2); three half-tiles stay in flight
        g8_wait_lgkm_barrier();
        mma(0, 0, bf0);
        g8_barrier();
        // ---- phase 2: quadrant (m 0-1, n 1) ----
        load_b(buf, 1, bf1, k0);
        stage(3, BUF ^ 1, kn);            // HB1 of tile t+1
        g8_wait_vm<6>();                  // retires HA1 of this tile (read in phase 3)
        g8_wait_lgkm_barrier();
        mma(0, 1, bf1);
        g8_barrier();
        // ---- phase 3: quadrant (m 2-3, n 1) ----
        load_a(buf, 1);
        stage(1, BUF ^ 1, kn);            // HA1 of tile t+1
        g8_wait_lgkm_barrier();
        mma(1, 1, bf1);
        g8_barrier();
        // ---- phase 4: quadrant (m 2-3, n 0): both fragments are in registers ----
        stage(2, BUF, kn2);               // HB0 of tile t+2 into THIS buffer (its B0 was read in phase 1 only)
        g8_wait_vm<6>();                  // retires HB0 + HA0 of tile t+1 (read in its phase 1)
        g8_barrier();
        mma(1, 0, bf0);
        g8_barrier();
    };
    for (int t = 0; t < nk; t += 2) {     // two tiles per iteration: compile-time buffer addresses
        ktile(std::integral_constant<int, 0>{}, t);
        ktile(std::integral_constant<int, 1>{}, t + 1);
    }
    g8_wait_vm<0>();                      // this wave's re-staged last tile has landed ...
    if (wr == 0) g8_barrier();            // re-align the two groups: every wave is done reading the ring
    g8_barrier();                         // ... and so has EVERY wave's (group 1 leaves its last loop barrier with up to six
                                          // LDS-DMAs in flight; vmcnt orders only a wave's own): nobody writes the epilogue
                                          // tile over the ring before all of them drained

    // ---- epilogue: accumulators (+bias, ReLU) -> LDS [pixel][cout] fp16 -> full-line stores (+ shortcut) ----
    const bool second = a.split > 0 && n0 >= a.split;     // fused pair: this cout tile belongs to one of the two outputs
    const int o_c = a.split > 0 ? (second ? a.c_out2 : a.split) : a.c_out;
    const int o_n0 = second ? n0 - a.split : n0;
    const int o_relu = second ? a.relu2 : a.relu;
    half_t* o_ptr = second ? out2 : out;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = wr * 128 + i * 32 + 8 * q + 4 * frag_half;
            const int co = n0 + col;
            floatx4 bv = {0.f, 0.f, 0.f, 0.f};
            bv = *reinterpret_cast<const floatx4*>(bias + co);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int prow = wc * 64 + j * 32 + frag_row;
                half4_t hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[i][j][4 * q + e] + bv[e];
                    if (o_relu) v = fmaxf(v, 0.f);
                    hv[e] = (half_t)v;
                }
                *reinterpret_cast<half4_t*>(smem + prow * OUT_ROW_BYTES + col * 2) = hv;
            }
        }
    }
    __syncthreads();
    constexpr int CPRO = TM / 8;                       // 16-byte chunks per tile row
    constexpr int EPI_ITERS = TN * CPRO / NT;          // 16
    const bool res_same = a.res_stride == 1 && a.res_offset == 0 && a.res_h == a.h_out && a.res_w == a.w_out;
    const int hw_out = a.h_out * a.w_out;
#pragma unroll 4
    for (int it = 0; it < EPI_ITERS; ++it) {
        const int idx = tid + it * NT;
        const int prow = idx / CPRO;
        const int ch = idx - prow * CPRO;
        const int m = m0 + prow;
        const int co = o_n0 + ch * 8;
        if (co + 8 > o_c) continue;                    // narrow second output of a fused pair (c_out2 < 256)
        uint4 v = *reinterpret_cast<const uint4*>(smem + prow * OUT_ROW_BYTES + ch * 16);
        if (residual != nullptr) {
            size_t rp = m;
            if (!res_same) {
                const int img = m / hw_out;
                const int rem = m - img * hw_out;
                const int ho = rem / a.w_out;
                const int wo = rem - ho * a.w_out;
                rp = (size_t)(img * a.res_h + ho * a.res_stride + a.res_offset) * a.res_w + (wo * a.res_stride + a.res_offset);
            }
            const uint4 rv = *reinterpret_cast<const uint4*>(residual + rp * a.c_out + co);
            half2_t* x = reinterpret_cast<half2_t*>(&v);
            const half2_t* r = reinterpret_cast<const half2_t*>(&rv);
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = x[e] + r[e];    // fp16 Add, like the reference graph
        }
        store_out16<2>(o_ptr + (size_t)m * o_c + co, v);
    }
}

int launch_conv_gemm8p(const MetroConvDesc& d, const void* in, const void* w, const float* bias, const void* ps,
                       const void* pb, const void* res, void* out, hipStream_t stream, const ConvSplit* split) {
    if (!conv_gemm4w_shape_ok(d, split)) {
        set_error("conv_gemm8p: needs a 1x1 stride-1 fp16 layer with c_in %% 128 == 0 (<= 2048), c_out %% 256 == 0 and pixels %% 256 == 0 "
                  "(got c_in %d, c_out %d, %d x %d x %d pixels)", d.c_in, d.c_out, d.n, d.h_out, d.w_out);
        return METRO_ERR_UNSUPPORTED;
    }
    ConvArgs a = make_conv_args(d);
    void* out2 = nullptr;
    if (split != nullptr && split->split > 0) {
        a.split = split->split; a.c_out2 = split->c_out2; a.relu2 = split->relu2;
        out2 = split->out2;
    }
    if (note_kernel("conv_gemm8p<256x256%s>%s%s", d.has_prologue ? ",pro" : "", d.has_residual ? "+res" : "", a.split > 0 ? "+pair" : ""))
        return METRO_OK;
    const int tiles_m = (d.c_out + g8::TM - 1) / g8::TM;
    const int tiles_n = (a.m_total + g8::TN - 1) / g8::TN;
    const half_t* r = d.has_residual ? static_cast<const half_t*>(res) : nullptr;
    if (d.has_prologue) {
        auto kern = conv_gemm8p_kernel<true>;
        constexpr int lds = g8::MAIN_BYTES + g8::PRO_BYTES;
        static PerDeviceInt done;
        if (const int st = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds, done, "conv_gemm8p<pro>")) return st;
        hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(g8::NT), lds, stream, a, static_cast<const half_t*>(in),
                           static_cast<const half_t*>(w), bias, static_cast<const half_t*>(ps), static_cast<const half_t*>(pb), r,
                           static_cast<half_t*>(out), static_cast<half_t*>(out2), tiles_m);
    } else {
        auto kern = conv_gemm8p_kernel<false>;
        constexpr int lds = g8::MAIN_BYTES;
        static PerDeviceInt done;
        if (const int st = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds, done, "conv_gemm8p")) return st;
        hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(g8::NT), lds, stream, a, static_cast<const half_t*>(in),
                           static_cast<const half_t*>(w), bias, nullptr, nullptr, r, static_cast<half_t*>(out),
                           static_cast<half_t*>(out2), tiles_m);
    }
    return launch_status("conv_gemm8p");
}

}  // namespace metro
