// Stem 7x7/2 convolution + 3x3/2 zero-padded max-pool in one persistent kernel (fp16 mode, base width 64).
//
// reference: resnet_v2.py:219-224 (conv2d_same 7x7 stride 2 with biases, no activation) followed by
// resnet_utils.max_pool2d_same (resnet_utils.py:138-185: explicit ZERO padding, then VALID 3x3/2 pooling).
// Run separately these are 76 + 36 us at batch 64: the 134 MB conv output is written and read back just to be
// reduced 4:1.  Here a block owns an 8x8 patch of POOLED pixels:
//   * the 39x40-pixel window of the bordered 4-channel fp16 image (prep_input_f16) it needs is LDS-DMA'd,
//     the next patch's window while the current one is computed;
//   * the 17x17 conv pixels under the patch (one halo row/column: 13 % recompute) are an implicit GEMM
//     [64 couts x 224] x [224 x 289 pixels] whose B fragments are read straight from the window (tap row r =
//     8 pixels x 4 channels = 32 contiguous fp16, the 8th pixel and 4th channel meet zero weights);
//   * the weights stay in registers as MFMA A-fragments for the whole launch (112 VGPRs per lane);
//   * conv + bias goes to LDS as fp16 (the value the separate kernels would have stored), the pool takes the
//     max of the 3x3 window with out-of-image conv positions contributing 0, and only pooled rows are stored.
#include <cstdlib>
#include <type_traits>

#include "metro_common.h"

namespace metro {

typedef _Float16 half_t;
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

__device__ __attribute__((aligned(16))) unsigned int g_zero_page_sp[4];   // zero-initialised

namespace sp {
constexpr int MG = 4;                         // wave groups over the pixel tiles (tiles mg, mg+4, mg+8)
constexpr int PP = 8;                         // pooled patch side
constexpr int CP = 2 * PP + 1;                // conv pixels per patch side (17)
constexpr int CPIX = CP * CP;                 // 289
constexpr int MT = (CPIX + 31) / 32;          // 10 MFMA pixel tiles
constexpr int WIN_R = 2 * (CP - 1) + 7;       // 39 window rows
constexpr int WIN_C = 2 * (CP - 1) + 8;       // 40 window columns (pixels of 4 channels = 8 bytes)
constexpr int WIN_ROW_BYTES = WIN_C * 8;      // 320
constexpr int WIN_CHUNKS = WIN_R * WIN_C / 2; // 780 16-byte chunks
constexpr int WIN_INSTR = (WIN_CHUNKS + 63) / 64;     // 13 DMA wave-instructions
constexpr int WIN_BYTES = WIN_INSTR * 1024;
constexpr int CONV_ROW = 64 * 2 + 8;          // rows of the [conv pixel][cout] tile: 34 banks apart, so the 8-byte
                                              // epilogue writes of 16 consecutive pixels cover the 32 banks once
constexpr int CONV_BYTES = CPIX * CONV_ROW;
constexpr int NBUF = 3;                       // windows in flight: the current one + 2 ahead (one ahead left the
                                              // DMA latency exposed: 6 us per patch where the arithmetic needs 1.5)
constexpr int WIN_OFF = 0;
constexpr int CONV_OFF = WIN_OFF + NBUF * WIN_BYTES;
constexpr int BIAS_OFF = CONV_OFF + ((CONV_BYTES + 15) / 16) * 16;
constexpr int LDS_BYTES = BIAS_OFF + 256;
constexpr int KK = 14;                        // 7 tap rows x 2 k-steps of 16
}  // namespace sp

struct StemPoolArgs {
    const float* img_f32;  // RAW: [n][side][side][3] fp32, cast + bordered on the way into LDS (architectures.py:29)
    const half_t* img;     // [n][side+6][side+8][4] fp16 (prep_input_f16)
    const half_t* w;       // [64][7][8][4] fp16
    const float* bias;     // [64]
    half_t* out;           // [n][side/4][side/4][64]
    int n, side;           // side % 32 == 0
    int n_patches;
};

__device__ __forceinline__ void sp_dma16(const void* gsrc, unsigned lds_addr) {
    asm volatile(
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, off"
        :
        : "v"(gsrc), "s"(lds_addr));
}
template <int N>
__device__ __forceinline__ void sp_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wave-uniform n in [0, 8]
__device__ __forceinline__ void sp_wait_vm_dyn(int n) {
    switch (n) {
        case 0: sp_wait_vm<0>(); break;
        case 1: sp_wait_vm<1>(); break;
        case 2: sp_wait_vm<2>(); break;
        case 3: sp_wait_vm<3>(); break;
        case 4: sp_wait_vm<4>(); break;
        case 5: sp_wait_vm<5>(); break;
        case 6: sp_wait_vm<6>(); break;
        case 7: sp_wait_vm<7>(); break;
        default: sp_wait_vm<8>(); break;
    }
}
__device__ __forceinline__ void sp_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// NSPLIT = 1: 4 waves, each holds both 32-cout weight tiles (112 VGPRs) and reuses every pixel fragment twice;
// NSPLIT = 2: 8 waves, a wave holds one cout tile (56 VGPRs): twice the waves per CU to overlap the phases
// RAW: the fp32 NHWC3 crops are read directly (prep_input_f16 fused away): every thread fetches ~3 window pixels
// of the NEXT patch into registers at the top of an iteration and writes them to the other window buffer as
// zero-bordered 4-channel fp16 at the end of it (ordinary loads: the compiler places their waits).
template <int NSPLIT, bool RAW>
__global__ __launch_bounds__(64 * sp::MG * NSPLIT, 2 * NSPLIT) void stem_pool_f16_kernel(StemPoolArgs a) {
    using namespace sp;
    constexpr int NW = MG * NSPLIT, NT = 64 * NW, CT = 2 / NSPLIT;
    constexpr int PS = 512 / NT;                  // pooled stores per wave per patch
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void lds_void_t;
    const unsigned smem_base = (unsigned)(size_t)(lds_void_t*)smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frag_row = lane & 31, frag_half = lane >> 5;
    const int G = gridDim.x;
    int p = blockIdx.x;
    if (p >= a.n_patches) return;
    const half_t* zero = reinterpret_cast<const half_t*>(g_zero_page_sp);
    const int hp = a.side + 6, wp = a.side + 8;          // bordered image
    const int ps = a.side / 4;                           // pooled side
    const int ppr = ps / PP;                             // patches per row
    const int cs = a.side / 2;                           // conv side

    // ---- launch-resident weights: both 32-cout tiles, 14 k-steps ------------------------------
    const int ct0 = NSPLIT == 2 ? (wave & 1) : 0;          // first cout tile of this wave
    const int mg = NSPLIT == 2 ? (wave >> 1) : wave;       // pixel-tile group
    half8_t wf[CT][KK];
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
            wf[i][kk] = *reinterpret_cast<const half8_t*>(a.w + (size_t)((ct0 + i) * 32 + frag_row) * 224 + kk * 16 + frag_half * 8);
    float* bias_l = reinterpret_cast<float*>(smem + BIAS_OFF);
    if (tid < 64) bias_l[tid] = a.bias[tid];

    // window chunk c = q*64 + lane (q = wave + 4*i): window row c / 20, pixel pair c % 20
    auto issue_window = [&](int patch, int buf) {
        const int img = patch / (ppr * ppr);
        const int rem = patch - img * ppr * ppr;
        const int r0 = 4 * PP * (rem / ppr) - 2, c0 = 4 * PP * (rem % ppr) - 2;   // window origin in the bordered image
        const half_t* base = a.img + (size_t)img * hp * wp * 4;
#pragma unroll
        for (int i = 0; i < (WIN_INSTR + NW - 1) / NW; ++i) {
            const int q = wave + NW * i;
            if (q < WIN_INSTR) {
                const int c = q * 64 + lane;
                const int wr = c / (WIN_C / 2), wc = (c - wr * (WIN_C / 2)) * 2;
                const int y = r0 + wr, x = c0 + wc;
                const bool ok = c < WIN_CHUNKS && (unsigned)y < (unsigned)hp && (unsigned)x < (unsigned)wp;
                sp_dma16(ok ? base + ((size_t)y * wp + x) * 4 : zero,
                         __builtin_amdgcn_readfirstlane(smem_base + WIN_OFF + buf * WIN_BYTES + q * 1024));
            }
        }
    };

    // RAW: window pixel idx = k*NT + tid (row idx / 40, column idx % 40 of the bordered window)
    constexpr int RK = (WIN_R * WIN_C + NT - 1) / NT;
    float rawpx[RAW ? RK : 1][3];
    auto raw_fetch = [&](int patch) {
        const int img = patch / (ppr * ppr);
        const int rem = patch - img * ppr * ppr;
        const int r0 = 4 * PP * (rem / ppr) - 2 - 3, c0 = 4 * PP * (rem % ppr) - 2 - 3;   // window origin in the IMAGE
        const float* base = a.img_f32 + (size_t)img * a.side * a.side * 3;
#pragma unroll
        for (int k = 0; k < RK; ++k) {
            const int idx = k * NT + tid;
            const int wr = idx / WIN_C, wc = idx - wr * WIN_C;
            const int y = r0 + wr, x = c0 + wc;
            const bool ok = idx < WIN_R * WIN_C && (unsigned)y < (unsigned)a.side && (unsigned)x < (unsigned)a.side;
            const float* s3 = base + ((size_t)(ok ? y : 0) * a.side + (ok ? x : 0)) * 3;
            rawpx[k][0] = ok ? s3[0] : 0.f; rawpx[k][1] = ok ? s3[1] : 0.f; rawpx[k][2] = ok ? s3[2] : 0.f;
        }
    };
    auto raw_commit = [&](int buf) {
#pragma unroll
        for (int k = 0; k < RK; ++k) {
            const int idx = k * NT + tid;
            if (idx < WIN_R * WIN_C) {
                half4_t v = {(half_t)rawpx[k][0], (half_t)rawpx[k][1], (half_t)rawpx[k][2], (half_t)0};
                *reinterpret_cast<half4_t*>(smem + WIN_OFF + buf * WIN_BYTES + idx * 8) = v;
            }
        }
    };

    // Window k is requested at the top of iteration k-2 (windows 0 and 1 up front).  VMEM operations of this
    // wave younger than window `it` when iteration `it` starts: the PS pooled stores of each iteration since the
    // request, and the DMA instructions of window it+1 (nw per wave) if that window exists.
    const int nw = (WIN_INSTR - wave + NW - 1) / NW;
    if constexpr (RAW) {
        raw_fetch(p);
        raw_commit(0);
    } else {
        issue_window(p, 0);
        if (p + G < a.n_patches) issue_window(p + G, 1);
    }
    int buf = 0;
    for (int it = 0;; ++it, p += G) {
        if constexpr (!RAW) {
            const int next_dma = p + G < a.n_patches ? nw : 0;
            sp_wait_vm_dyn((it == 0 ? 0 : it == 1 ? PS : 2 * PS) + next_dma);
        }
        sp_barrier();
        if constexpr (RAW) {
            if (p + G < a.n_patches) raw_fetch(p + G);
        } else {
            if (p + 2 * G < a.n_patches) issue_window(p + 2 * G, buf + 2 >= NBUF ? buf + 2 - NBUF : buf + 2);
        }

        const int img = p / (ppr * ppr);
        const int rem = p - img * ppr * ppr;
        const int py0 = PP * (rem / ppr), px0 = PP * (rem % ppr);
        const char* wl = smem + WIN_OFF + buf * WIN_BYTES;
        char* cl = smem + CONV_OFF;
        // ---- conv: pixel tiles mt = wave, wave+4, wave+8.  The epilogue of a tile (bias, fp16, LDS) is issued in
        // the MFMA shadow of the NEXT tile (two accumulator sets); run back to back it cost 16 of 64 us.
        auto tile_base = [&](int mt, int& m) -> const char* {
            m = mt * 32 + frag_row;
            const int mc = m < CPIX ? m : CPIX - 1;
            const int cyl = mc / CP, cxl = mc - cyl * CP;
            return wl + (2 * cyl) * WIN_ROW_BYTES + (2 * cxl) * 8 + frag_half * 16;
        };
        auto epi_part = [&](const floatx16 (&acc)[CT], int m, int c) {     // c in [0, 4*CT): (cout tile, quad)
            const int i = c >> 2, q = c & 3;
            const int co = (ct0 + i) * 32 + 8 * q + 4 * frag_half;
            const floatx4 bv = *reinterpret_cast<const floatx4*>(bias_l + co);
            half4_t hv;
#pragma unroll
            for (int e = 0; e < 4; ++e) hv[e] = (half_t)(acc[i][4 * q + e] + bv[e]);
            if (m < CPIX)
                *reinterpret_cast<half4_t*>(cl + m * CONV_ROW + co * 2) = hv;
        };
        auto conv_tile = [&](int mt, floatx16 (&acc)[CT], int& m, auto with_prev, const floatx16 (&pacc)[CT], int pm) {
            const char* bp = tile_base(mt, m);
#pragma unroll
            for (int e = 0; e < 16; ++e)
#pragma unroll
                for (int i = 0; i < CT; ++i) acc[i][e] = 0.f;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                const half8_t bf = *reinterpret_cast<const half8_t*>(bp + (kk >> 1) * WIN_ROW_BYTES + (kk & 1) * 32);
#pragma unroll
                for (int i = 0; i < CT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i][kk], bf, acc[i], 0, 0, 0);
                if constexpr (decltype(with_prev)::value) {
                    if (kk >= 2 && kk < 2 + 4 * CT) epi_part(pacc, pm, kk - 2);
                }
            }
        };
        {
            using Yes = std::integral_constant<bool, true>;
            using No = std::integral_constant<bool, false>;
            floatx16 accA[CT];
            int mA;
            if constexpr (NSPLIT == 2) {
                // one accumulator set (128-VGPR budget of 4 waves per SIMD): tiles one after the other
#pragma unroll 1
                for (int mt = mg; mt < MT; mt += MG) {
                    conv_tile(mt, accA, mA, No{}, accA, 0);
#pragma unroll
                    for (int c = 0; c < 4 * CT; ++c) epi_part(accA, mA, c);
                }
            } else {
            floatx16 accB[CT];
            int mB;
            conv_tile(mg, accA, mA, No{}, accA, 0);
            conv_tile(mg + MG, accB, mB, Yes{}, accA, mA);
            if (mg + 2 * MG < MT) {
                conv_tile(mg + 2 * MG, accA, mA, Yes{}, accB, mB);
#pragma unroll
                for (int c = 0; c < 4 * CT; ++c) epi_part(accA, mA, c);
            } else {
#pragma unroll
                for (int c = 0; c < 4 * CT; ++c) epi_part(accB, mB, c);
            }
            }
        }
        sp_barrier();
        // ---- pool: (pooled pixel, 8-channel chunk) items, zero where the conv position is outside -------
#pragma unroll
        for (int r = 0; r < PS; ++r) {
            const int item = tid + r * NT;
            const int c8 = item & 7, pp = item >> 3;
            const int ppy = pp >> 3, ppx = pp & 7;
            // out-of-image conv positions contribute 0 (mask, no branches: all 9 reads are issued back to back)
            half8_t best = {};
            {
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {           // one window row at a time (3 reads in flight: registers)
                    uint4 v[3];                            // rows are 8-byte aligned: two 8-byte reads per element
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const uint2* src = reinterpret_cast<const uint2*>(cl + ((2 * ppy + dy) * CP + 2 * ppx + dx) * CONV_ROW + c8 * 16);
                        const uint2 lo = src[0], hi = src[1];
                        v[dx] = make_uint4(lo.x, lo.y, hi.x, hi.y);
                    }
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const int cy = 2 * (py0 + ppy) - 1 + dy, cx = 2 * (px0 + ppx) - 1 + dx;
                        const unsigned keep = ((unsigned)cy < (unsigned)cs && (unsigned)cx < (unsigned)cs) ? 0xffffffffu : 0u;
                        v[dx].x &= keep; v[dx].y &= keep; v[dx].z &= keep; v[dx].w &= keep;
                        const half8_t h = *reinterpret_cast<const half8_t*>(&v[dx]);
                        best = (dy == 0 && dx == 0) ? h : __builtin_elementwise_max(best, h);
                    }
                }
            }
            store_out16<1>(a.out + (((size_t)img * ps + py0 + ppy) * ps + px0 + ppx) * 64 + c8 * 8, *reinterpret_cast<const uint4*>(&best));
        }
        if (p + G >= a.n_patches) break;
        buf = buf + 1 == NBUF ? 0 : buf + 1;
        if constexpr (RAW) raw_commit(buf);      // last read by the conv of iteration it-2: two barriers ago
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 4: the stem for 256 x 256 crops with the max-pool IN REGISTERS (stem_pool_rows_kernel).
// The kernel above spends 19 us moving crops in and pooled rows out, 31 us in its conv phase (7 of them MFMAs, 10 the conv -> LDS
// writes) and 9 us pooling out of LDS: 400 KB of LDS traffic per 8 x 8 pooled patch, three barrier-separated phases per patch.  Here
// the conv pixels are the ROWS of the MFMA (A = pixel fragments from the LDS window, B = the launch-resident weight fragments), so a
// lane holds 16 conv pixels of ONE channel: with row m of a 32-pixel tile = conv column 32 w + m, lane half h' holds the column
// quads 8 q + 4 h' + {0..3} = (E[2g], O[2g], E[2g+1], O[2g+1]), g = 2 q + h' (E / O: even / odd conv columns) and
//     pooled X = 2g + 1 = max(O[2g], E[2g+1], O[2g+1])                           in the lane,
//     pooled X = 2g     = max(O[2g-1], E[2g], O[2g]),  O[2g-1] from the other lane half (one 32-lane exchange of a packed pair per quad);
// the vertical 3-max runs over consecutive conv rows of the same lane (a rolling carry), zero rows / columns of the pool's padding
// (resnet_utils.py:138-185) are the initial carry / exchange value.  Only column 16 w of x-tile w needs a value of ANOTHER wave
// (O of the left neighbour's last column): each wave also pools that column of its own (the "edge") and the store phase folds it in.
//   * block = 4 waves = the 4 x-tiles of a band of 8 pooled rows x the full width (no horizontal halo; one conv row of vertical halo:
//     6 % recompute against 13 % + 10 % tile padding above: 32.0 GFLOP issued at batch 64 instead of 37.6); every wave all 64
//     channels (112 VGPRs of weights; a pixel fragment is read once for both channel tiles);
//   * the band is walked one POOLED row (two conv rows) per barrier; window row 4Y + j is tap row j of conv row 2Y and tap row j - 2
//     of conv row 2Y + 1, so each pixel fragment is read once for both rows (18 LDS reads per 56 MFMAs, three ahead of their use);
//   * the fp32 crop rows come by LDS-DMA into a staging ring two pooled rows ahead and are cast into the zero-bordered 4-channel
//     fp16 window ring (architectures.py:29) one pooled row ahead (read at the top of an iteration, written at its end);
//   * pooled rows leave through a 10 KiB LDS tile ([X][channel], 4-byte writes after a lane-pair swap, 16-byte reads and stores):
//     20 KB of LDS writes + reads per pooled row against 111 KB of conv tile traffic before.
// What it is bound by (tools/stem_knockouts.sh, tools/stem_clock.py, tools/stem_pmc.sh at batch 64): a wave's iteration is MFMAs
// (1 800 cycles at full rate) + pooling (1 300) + tile / staging / window traffic and requests (1 400), one after the other -- 330
// VALU instructions per 56 MFMAs do not fit the five issue slots an MFMA leaves, and the pooling of a row needs its accumulators
// finished; two co-resident blocks overlap by a third (26 us each against 20 alone).  MFMA pipe busy 36 %.
// Bit-identical to the kernel above (same k order per output, same fp16 rounding before the max).
namespace sp2 {
constexpr int SIDE = 256, PS = 64, PY = 8, NW = 4, NT = 256, KK = 14;      // crop side, pooled side, pooled rows per band
constexpr int WROW = (SIDE + 8) * 8;          // bordered fp16 4-channel row: 2112 bytes
constexpr int NWR = 16;                       // window ring rows (a pooled row's two conv rows use 9, the next one's 4 new rows are cast meanwhile)
constexpr int SROW = SIDE * 12;               // fp32 crop row: 3072 bytes = 3 LDS-DMA instructions
constexpr int NSR = 12;                       // staging ring rows: three groups of four (one being cast, two in flight)
constexpr int OUT_PITCH = 160;                // pooled row tile [64 X][64 channels] fp16, rows 40 banks apart: the 4-byte writes of
                                              // the even lanes (column X) and the odd lanes (column X + 1) of a wave half never collide
constexpr int OUT_BYTES = PS * OUT_PITCH;
constexpr int EDGE_BYTES = (NW + 1) * 128;    // the edge column of every x-tile [64 channels] fp16, and a row of -inf (no edge to fold)
constexpr int WIN_OFF = 0;
constexpr int STG_OFF = WIN_OFF + NWR * WROW;
constexpr int OUT_OFF = STG_OFF + NSR * SROW;
constexpr int EDGE_OFF = OUT_OFF + OUT_BYTES;
constexpr int LDS_BYTES = EDGE_OFF + EDGE_BYTES;          // 81 536: two blocks per CU, to the byte
static_assert(2 * LDS_BYTES <= 160 * 1024, "two blocks per CU");
}  // namespace sp2

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

// packed fp16 max as the instruction (the builtin canonicalises both operands first: a third of the pooling's VALU work)
__device__ __forceinline__ half2_t sp2_max(half2_t a, half2_t b) {
    half2_t r;
    asm("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ half2_t sp2_max_pair_hi(half2_t p) {      // (max(lo, hi), hi)
    half2_t r;
    asm("v_pk_max_f16 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(p));
    return r;
}
__device__ __forceinline__ half2_t sp2_max_pair_both(half2_t p) {    // (max(lo, hi), max(lo, hi))
    half2_t r;
    asm("v_pk_max_f16 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1]" : "=v"(r) : "v"(p));
    return r;
}
__device__ __forceinline__ half2_t sp2_hi_lo(half2_t a, half2_t b) {        // (a.hi, b.lo)
    return __builtin_bit_cast(half2_t, __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, a), 0x05040302u));
}
__device__ __forceinline__ half2_t sp2_xchg32(half2_t v) {          // the value of lane ^ 32
    const int i = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(half2_t, __shfl_xor(i, 32, 64));
}

__global__ __launch_bounds__(sp2::NT, 2) void stem_pool_rows_kernel(StemPoolArgs a) {
    using namespace sp2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void lds_void_t;
    const unsigned smem_base = (unsigned)(size_t)(lds_void_t*)smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // x-tile: conv columns 32 w ..., pooled columns 16 w ...
    const int n = lane & 31, hh = lane >> 5;
    const int img = blockIdx.x / (PS / PY), band = blockIdx.x % (PS / PY);
    const int Y0 = band * PY;

    const float* crop = a.img_f32 + (size_t)img * SIDE * SIDE * 3;
    // crop row i (clamped: rows outside the crop are cast to zeros whatever arrived) -> a staging slot: waves 0-2 a KiB each
    auto issue_row = [&](int i, int slot) {
        if (wave < 3) {
            const int ic = i < 0 ? 0 : i > SIDE - 1 ? SIDE - 1 : i;
            const float* src = crop + (size_t)ic * SIDE * 3 + wave * 256 + lane * 4;
            sp_dma16(src, __builtin_amdgcn_readfirstlane(smem_base + STG_OFF + slot * SROW + wave * 1024));
        }
    };
    // window row p (bordered: crop row p - 3) from a staging slot: thread = pixel.  Read and write are separate steps: the steady
    // state reads at the top of an iteration and writes at its end (an LDS latency and a half otherwise sit on the critical path);
    // no branch around a read (a branch is a wait per read)
    auto cast_read = [&](int p, int slot, half4_t& v) {
        const float* s3 = reinterpret_cast<const float*>(smem + STG_OFF + slot * SROW) + tid * 3;
        const bool ok = (unsigned)(p - 3) < (unsigned)SIDE;
        const float r = s3[0], g = s3[1], b = s3[2];
        v = half4_t{(half_t)(ok ? r : 0.f), (half_t)(ok ? g : 0.f), (half_t)(ok ? b : 0.f), (half_t)0};
    };
    auto cast_write = [&](int p, const half4_t& v) {
        *reinterpret_cast<half4_t*>(smem + WIN_OFF + (p & (NWR - 1)) * WROW + (tid + 3) * 8) = v;
    };
    // group t = the four window rows 4 (Y0 + t) + 9 ... + 12 that pooled row Y0 + t + 1 adds to pooled row Y0 + t's; staging slots 4 (t % 3) ...
    auto issue_group = [&](int t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) issue_row(4 * (Y0 + t) + 6 + r, (t % 3) * 4 + r);
    };
    auto cast_group_read = [&](int t, half4_t (&v)[4]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) cast_read(4 * (Y0 + t) + 9 + r, (t % 3) * 4 + r, v[r]);
    };
    auto cast_group_write = [&](int t, const half4_t (&v)[4]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) cast_write(4 * (Y0 + t) + 9 + r, v[r]);
    };

    // ---- pipeline fill, part 1: the eleven window rows 4 Y0 - 2 ... 4 Y0 + 8 (the conv row above the band and the first pooled row's
    //      two) are requested; the weights meanwhile ----
    const int plo = 4 * Y0 - 2;
#pragma unroll
    for (int r = 0; r < 11; ++r) issue_row(plo + r - 3, r);
    // launch-resident weights as B fragments: both 32-channel tiles, 14 k-steps; the per-lane bias of its two channels
    // Round 5: the 28 KiB weight matrix is fetched ONCE per block in whole 448-byte rows into the (not yet zeroed) window area and
    // every wave reads its 28 fragments from there -- fetched in fragment order straight from global memory it is 32 rows x 32
    // bytes per wave instruction: 3 584 L2 requests per block, 7.3 M per launch at batch 256 against 2.6 M for the crops and the
    // pooled output together (the L2s answer a near-constant request rate: DESIGN.md section 5).  Rows padded to 464 bytes:
    // conflict-free 16-byte reads of 32 rows.
    constexpr int WST_ROW = 224 * 2 + 16;
    static_assert(64 * WST_ROW <= NWR * WROW, "the staged weights must fit the window area");
    for (int c = tid; c < 64 * 28; c += NT)
        *reinterpret_cast<uint4*>(smem + WIN_OFF + (c / 28) * WST_ROW + (c % 28) * 16) = *reinterpret_cast<const uint4*>(a.w + (size_t)(c / 28) * 224 + (c % 28) * 8);
    float bias_l[2] = {a.bias[n], a.bias[32 + n]};
    __syncthreads();
    half8_t wf[2][KK];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
            wf[i][kk] = *reinterpret_cast<const half8_t*>(smem + WIN_OFF + (i * 32 + n) * WST_ROW + kk * 32 + hh * 16);
    __syncthreads();                               // every wave has its fragments: the window area may be zeroed
    // zero window (the borders stay zero: the cast only writes the 256 interior pixels of a row); the -inf row of the edge table
    for (int i = tid; i < NWR * WROW / 16; i += NT) reinterpret_cast<uint4*>(smem + WIN_OFF)[i] = make_uint4(0, 0, 0, 0);
    if (tid < 32) reinterpret_cast<unsigned*>(smem + EDGE_OFF + NW * 128)[tid] = 0xfc00fc00u;
    sp_wait_vm<0>();
    // every ordinary load is waited for HERE: a compiler-placed wait at the first use inside the row loop would be a vmcnt(0) per row
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) asm volatile("" : "+v"(wf[i][kk]));
    asm volatile("" : "+v"(bias_l[0]), "+v"(bias_l[1]));
    sp_barrier();
#pragma unroll
    for (int r = 0; r < 11; ++r) {
        half4_t v;
        cast_read(plo + r, r, v);
        cast_write(plo + r, v);
    }
    sp_barrier();            // the staging slots are free again, the window rows visible
    issue_group(0);
    issue_group(1);

    // pooling state per channel tile: packed pairs (X = 2g, 2g + 1) per quad q; the edge column (its last O) likewise
    half2_t carry[2][4], ecarry[2];
    const half2_t zero2 = {(half_t)0, (half_t)0};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) carry[i][q] = zero2;
        ecarry[i] = zero2;
    }
    const unsigned lane_win = (unsigned)(512 * wave + 16 * n + 16 * hh);      // byte of the lane's pixel pair inside a window row
    // pooled tile: quad q's columns X = 16 w + 4 q + 2 h' (even lanes), X + 1 (odd lanes), channels (n & ~1, + 1) of tile i at + 64 i
    const unsigned out_lane = (unsigned)(OUT_OFF + (16 * wave + 2 * hh + (n & 1)) * OUT_PITCH + (n & ~1) * 2);
    const unsigned out_sel = (n & 1) ? 0x03020706u : 0x05040100u;             // (other.hi, mine.hi) / (mine.lo, other.lo)
    // quad -1 is the left x-tile's (its O is folded in by the store phase: -inf here) or, for x-tile 0, the pool's zero column
    half2_t left;
    left[0] = (half_t)0;
    left[1] = wave == 0 ? (half_t)0 : (half_t)-INFINITY;

    // conv row y: 14 k-steps, pixel fragments straight from the window (tap row kk >> 1, pixels 4 (kk & 1) + 2 h' ...)
    auto conv_row = [&](floatx16 (&acc)[2], int y, auto pin_c) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        // fragments are read THREE k-steps ahead of their MFMAs (four buffers): an LDS read takes ~150 cycles, a k-step's two
        // MFMAs 64 -- left to itself the compiler reads one step ahead and every MFMA pair waits
        auto frag = [&](int kk) {
            const int slot = (2 * y + (kk >> 1)) & (NWR - 1);
            return *reinterpret_cast<const half8_t*>(smem + WIN_OFF + slot * WROW + lane_win + (kk & 1) * 32);
        };
#ifndef SP2_PF
#define SP2_PF 4
#endif
        half8_t pf[SP2_PF];
#pragma unroll
        for (int kk = 0; kk < SP2_PF - 1; ++kk) pf[kk] = frag(kk);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            if (kk + SP2_PF - 1 < KK) pf[(kk + SP2_PF - 1) % SP2_PF] = frag(kk + SP2_PF - 1);
            asm volatile("" ::: "memory");          // the read stays ahead of this k-step's MFMAs
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pf[kk % SP2_PF], wf[0][kk], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pf[kk % SP2_PF], wf[1][kk], acc[1], 0, 0, 0);
        }
        if constexpr (decltype(pin_c)::value) {      // on its own: the emitted order (with the pooling: pinned by the caller)
            __builtin_amdgcn_sched_group_barrier(0x100, SP2_PF - 1, 0);
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // Conv rows 2Y and 2Y + 1 together: window row 4Y + j is tap row j of the first and tap row j - 2 of the second, so each of its two
    // fragments is read ONCE for both rows (18 reads instead of 28: the fragment reads are 70 % of the kernel's LDS traffic, and two
    // co-resident blocks were bound by it); four independent accumulator chains.  The k order of every output is unchanged.
    auto conv_pair = [&](floatx16 (&accA)[2], floatx16 (&accB)[2], int Y) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) { accA[i][e] = 0.f; accB[i][e] = 0.f; }
        auto frag = [&](int f) {          // fragment f = 2 j + half of window row 4Y + j
            const int slot = (4 * Y + (f >> 1)) & (NWR - 1);
            return *reinterpret_cast<const half8_t*>(smem + WIN_OFF + slot * WROW + lane_win + (f & 1) * 32);
        };
        half8_t pf[SP2_PF];
#pragma unroll
        for (int f = 0; f < SP2_PF - 1; ++f) pf[f] = frag(f);
#pragma unroll
        for (int f = 0; f < 18; ++f) {
            if (f + SP2_PF - 1 < 18) pf[(f + SP2_PF - 1) % SP2_PF] = frag(f + SP2_PF - 1);
            if (f < KK) {
                accA[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pf[f % SP2_PF], wf[0][f < KK ? f : 0], accA[0], 0, 0, 0);
                accA[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pf[f % SP2_PF], wf[1][f < KK ? f : 0], accA[1], 0, 0, 0);
            }
            if (f >= 4) {
                accB[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pf[f % SP2_PF], wf[0][f >= 4 ? f - 4 : 0], accB[0], 0, 0, 0);
                accB[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pf[f % SP2_PF], wf[1][f >= 4 ? f - 4 : 0], accB[1], 0, 0, 0);
            }
        }
        // the emitted order: fragment reads SP2_PF - 1 ahead of their MFMAs
        __builtin_amdgcn_sched_group_barrier(0x100, SP2_PF - 1, 0);
#pragma unroll
        for (int f = 0; f < 4; ++f) {          // window rows 0, 1: the first conv row only
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#pragma unroll
        for (int f = 4; f < KK; ++f) {         // rows 2 ... 6: both
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#pragma unroll
        for (int f = KK; f < 18; ++f) {        // rows 7, 8: the second only
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // bias, fp16 (the value the reference's conv stores) and the horizontal 3-max of a conv row: branch-free, so that it can sit in
    // the shadow of the next row's MFMAs.  hp[i][q] = pooled columns (2g, 2g + 1) of channel tile i; eh[i].hi = the tile's last O
    auto hmax_row = [&](const floatx16 (&acc)[2], half2_t (&hp)[2][4], half2_t (&eh)[2]) {
        half2_t P0[2][4], P1[2][4], R[2][4];             // (E[2g], O[2g]), (E[2g+1], O[2g+1]) per quad; the other half's P1
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                P0[i][q][0] = (half_t)(acc[i][4 * q + 0] + bias_l[i]);
                P0[i][q][1] = (half_t)(acc[i][4 * q + 1] + bias_l[i]);
                P1[i][q][0] = (half_t)(acc[i][4 * q + 2] + bias_l[i]);
                P1[i][q][1] = (half_t)(acc[i][4 * q + 3] + bias_l[i]);
            }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) R[i][q] = sp2_xchg32(P1[i][q]);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // O[2g - 1] (the high half of): lane half 1 <- half 0's quad q; half 0 <- half 1's quad q - 1
                const half2_t tsrc = hh ? R[i][q] : (q == 0 ? left : R[i][q > 0 ? q - 1 : 0]);
                const half2_t A = sp2_max_pair_hi(P0[i][q]);         // (max(E0, O0), O0)
                const half2_t C = sp2_max_pair_both(P1[i][q]);       // (max(E1, O1), same)
                hp[i][q] = sp2_max(A, sp2_hi_lo(tsrc, C));           // (max(T, E0, O0), max(O0, E1, O1))
            }
            eh[i] = R[i][3];                              // lane half 0 holds half 1's quad 3 after the exchange
        }
    };
    // the pooled row out of its LDS tile: 16 bytes per lane, the left x-tile's edge folded into column 16 w.  Read at the top of
    // an iteration (the tile is rewritten at the end of it), stored after the conv rows
    auto store_read = [&](half8_t (&v)[2]) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int c = tid + r * NT;
            const int X = c >> 3, c8 = c & 7;
            const bool fold = (X & 15) == 0 && X > 0;       // O of the left x-tile's last column, pooled over the same three rows
            typedef half2_t half2x4_t[4];
            half2x4_t t, e;
            *reinterpret_cast<uint4*>(&t) = *reinterpret_cast<const uint4*>(smem + OUT_OFF + X * OUT_PITCH + c8 * 16);
            *reinterpret_cast<uint4*>(&e) = *reinterpret_cast<const uint4*>(smem + EDGE_OFF + (fold ? (X >> 4) - 1 : NW) * 128 + c8 * 16);
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] = sp2_max(t[k], e[k]);
            v[r] = *reinterpret_cast<const half8_t*>(&t);
        }
    };
    auto store_write = [&](int Y, const half8_t (&v)[2]) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int c = tid + r * NT;
            const int X = c >> 3, c8 = c & 7;
            store_out16<1>(a.out + (((size_t)img * PS + Y) * PS + X) * 64 + c8 * 8, *reinterpret_cast<const uint4*>(&v[r]));
        }
    };

#define SP2_CLK(i) do { } while (0)
    floatx16 accA[2], accB[2];
    // ---- the conv row above the band: only feeds the carry (band 0: the pool's zero row is the initial carry) ----
    if (Y0 > 0) {
        half2_t hp[2][4], eh[2];
        conv_row(accA, 2 * Y0 - 1, std::true_type{});
        hmax_row(accA, hp, eh);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) carry[i][q] = hp[i][q];
            ecarry[i] = eh[i];
        }
    }
    // ---- one iteration per pooled row Y = Y0 + t: conv rows 2Y (accA) and 2Y + 1 (accB, with row 2Y's pooling in the shadow of its
    //      MFMAs); request of the crop rows that iteration t + 2 casts; cast of the window rows 4Y + 9 ... 4Y + 12 ----
    for (int t = 0; t < PY; ++t) {
        const int Y = Y0 + t;
        SP2_CLK(0);
        // the crop rows cast in this iteration were requested two iterations ago: one iteration of requests (4) is younger.  The
        // pooled-row stores in between are not counted: requests land in order among themselves, so "at most 4 operations
        // outstanding" implies this group has landed whatever the stores do
        if (wave < 3) sp_wait_vm<4>();
        sp_barrier();            // ... and everybody's share has landed; the last pooled row's tile is complete
        SP2_CLK(1);
        half8_t sv[2];
        half4_t cv[4];
        store_read(sv);          // (whatever the tile holds in the first iteration: a branch would be a wait per read)
        cast_group_read(t, cv);
        issue_group(t + 2);
        SP2_CLK(2);
        __builtin_amdgcn_sched_barrier(0);
        conv_pair(accA, accB, Y);
        half2_t hpA[2][4], ehA[2], hpB[2][4], ehB[2];
        hmax_row(accA, hpA, ehA);
        SP2_CLK(3);
        hmax_row(accB, hpB, ehB);
        if (t > 0) store_write(Y - 1, sv);
        cast_group_write(t, cv);
        SP2_CLK(4);
        // vertical 3-max; lane n holds (X, X + 1) of channel n: the lane pair (n, n ^ 1) swaps one column so that the even lane
        // writes column X of channels (n, n + 1) and the odd lane column X + 1 of (n - 1, n): one 4-byte write per quad
        sp_barrier();            // every wave has read the last pooled row out of the tile (at the top of this iteration)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const half2_t pooled = sp2_max(sp2_max(carry[i][q], hpA[i][q]), hpB[i][q]);
                const unsigned mine = __builtin_bit_cast(unsigned, pooled);
                const unsigned other = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine, 0xB1, 0xF, 0xF, true);      // quad_perm [1,0,3,2]
                *reinterpret_cast<unsigned*>(smem + out_lane + q * 4 * OUT_PITCH + i * 64) = __builtin_amdgcn_perm(other, mine, out_sel);
                carry[i][q] = hpB[i][q];
            }
            const half2_t ep = sp2_max(sp2_max(ecarry[i], ehA[i]), ehB[i]);
            if (hh == 0) *reinterpret_cast<half_t*>(smem + EDGE_OFF + wave * 128 + (i * 32 + n) * 2) = ep[1];
            ecarry[i] = ehB[i];
        }
        SP2_CLK(5);
    }
    sp_barrier();
    {
        half8_t sv[2];
        store_read(sv);
        store_write(Y0 + PY - 1, sv);       // the band's last pooled row
    }
#undef SP2_CLK
}

static int sp_env_int(const char* name, int dflt) { return tuning_knob(name, dflt); }

bool stem_pool_f16_supported(int side, int base_width) {
    static const int enabled = sp_env_int("METRO_STEM_POOL", 1);
    return enabled && base_width == 64 && side % 32 == 0 && side >= 32;
}

template <int NSPLIT, bool RAW>
static int launch_sp(const StemPoolArgs& a, hipStream_t stream);

int launch_stem_pool_f16(const void* prepped, const void* w, const float* bias, void* out, int n, int side,
                         hipStream_t stream) {
    if (!stem_pool_f16_supported(side, 64)) { set_error("stem_pool_f16: unsupported shape (side %d)", side); return METRO_ERR_INVALID_ARG; }
    StemPoolArgs a;
    a.img = static_cast<const half_t*>(prepped);
    a.w = static_cast<const half_t*>(w);
    a.bias = bias;
    a.out = static_cast<half_t*>(out);
    a.n = n; a.side = side;
    const int ppr = side / 4 / sp::PP;
    a.n_patches = n * ppr * ppr;
    a.img_f32 = nullptr;
    static const int split = sp_env_int("METRO_STEM_SPLIT", 2);
    if (split == 2) return launch_sp<2, false>(a, stream);
    return launch_sp<1, false>(a, stream);
}

bool stem_pool_f32in_supported(int side, int base_width) {
    static const int enabled = sp_env_int("METRO_STEM_RAW", 1);
    return enabled && stem_pool_f16_supported(side, base_width);
}

int launch_stem_pool_f32in(const float* images, const void* w, const float* bias, void* out, int n, int side,
                           hipStream_t stream) {
    if (!stem_pool_f32in_supported(side, 64)) { set_error("stem_pool_f32in: unsupported shape (side %d)", side); return METRO_ERR_INVALID_ARG; }
    StemPoolArgs a;
    a.img_f32 = images;
    a.img = nullptr;
    a.w = static_cast<const half_t*>(w);
    a.bias = bias;
    a.out = static_cast<half_t*>(out);
    a.n = n; a.side = side;
    const int ppr = side / 4 / sp::PP;
    a.n_patches = n * ppr * ppr;
    static const int rows = sp_env_int("METRO_STEM_ROWS", 1);
    if (rows && side == sp2::SIDE) {
        if (note_kernel("stem_pool_f16<rows,f32in>")) return METRO_OK;
        static PerDeviceInt cap;
        int grid_cap = 0;
        if (const int st = ensure_dyn_lds_and_grid_cap(reinterpret_cast<const void*>(stem_pool_rows_kernel), sp2::NT, sp2::LDS_BYTES, cap, "stem_pool_f16<rows>", 0, &grid_cap)) return st;
        hipLaunchKernelGGL(stem_pool_rows_kernel, dim3(n * (sp2::PS / sp2::PY)), dim3(sp2::NT), sp2::LDS_BYTES, stream, a);
        return launch_status("stem_pool_f16<rows>");
    }
    static const int split = sp_env_int("METRO_STEM_RAW_SPLIT", 2);
    if (split == 1) return launch_sp<1, true>(a, stream);
    return launch_sp<2, true>(a, stream);
}

template <int NSPLIT, bool RAW>
static int launch_sp(const StemPoolArgs& a, hipStream_t stream) {
    if (note_kernel("stem_pool_f16<split%d%s>", NSPLIT, RAW ? ",f32in" : "")) return METRO_OK;
    auto kern = stem_pool_f16_kernel<NSPLIT, RAW>;
    constexpr int NT = 64 * sp::MG * NSPLIT;
    static PerDeviceInt cap;
    int grid_cap = 0;
    if (const int st = ensure_dyn_lds_and_grid_cap(reinterpret_cast<const void*>(kern), NT, sp::LDS_BYTES, cap, "stem_pool_f16", 0, &grid_cap))
        return st;
    const int grid = a.n_patches < grid_cap ? a.n_patches : grid_cap;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), sp::LDS_BYTES, stream, a);
    return launch_status("stem_pool_f16");
}

}  // namespace metro
