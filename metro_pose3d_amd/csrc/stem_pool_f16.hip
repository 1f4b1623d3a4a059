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
#if defined(METRO_DBG_SP_SKIP_CONV) || defined(METRO_DBG_SP_SKIP_CONVWRITE)
    for (int i = tid; i < CONV_BYTES / 16; i += NT) reinterpret_cast<uint4*>(smem + CONV_OFF)[i] = make_uint4(0, 0, 0, 0);
#endif

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
#ifdef METRO_DBG_SP_LINEAR_WINDOW   // timing experiment: same bytes, contiguous source
                sp_dma16(ok ? a.img + ((size_t)(patch % (a.n * 8 * 8)) * WIN_BYTES / 2 + c * 8) : zero,
#else
                sp_dma16(ok ? base + ((size_t)y * wp + x) * 4 : zero,
#endif
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
#ifdef METRO_DBG_SP_SKIP_CONVWRITE
            if (m < CPIX && a.n < 0)
#else
            if (m < CPIX)
#endif
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
#ifdef METRO_DBG_SP_SKIP_MFMA      // timing experiments only (tools/build_dbg_variants.sh)
#pragma unroll
                for (int i = 0; i < CT; ++i) acc[i][0] += (float)bf[i] * (float)wf[i][kk][0];
#else
#pragma unroll
                for (int i = 0; i < CT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i][kk], bf, acc[i], 0, 0, 0);
#endif
                if constexpr (decltype(with_prev)::value) {
                    if (kk >= 2 && kk < 2 + 4 * CT) epi_part(pacc, pm, kk - 2);
                }
            }
        };
#ifdef METRO_DBG_SP_SKIP_CONV
        if (a.n < 0)
#endif
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
#ifdef METRO_DBG_SP_SKIP_POOL
            if (a.n < 0)
#endif
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
