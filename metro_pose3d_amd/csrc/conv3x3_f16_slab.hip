// 3x3 stride-1 (dilated) convolution with TAP REUSE: the activation slab of a pixel tile is staged
// in LDS once per channel chunk and all 9 taps are served from it.
//
// The generic implicit GEMM (conv_igemm_f16_dma.hip) re-fetches the 256-pixel activation tile for
// every tap, so a 3x3 layer moves 9x its activations from L2 into LDS; PMC showed those layers
// bound by L2->LDS delivery, not by MFMA.  Here:
//   * a tile is 256 or 512 CONSECUTIVE pixels in (n,h,w) order (whole rows / whole images: consecutive
//     in NHWC memory too).  Its slab is the flattened pixel range [m0 - halo, m0 + TN + halo),
//     halo = dil*W: one contiguous range -> plain row-by-row LDS-DMA, 1/9 of the traffic;
//   * tap (dr,ds) of tile pixel t reads slab row halo + t + (dr*W + ds)*dil, or a ZERO row when the
//     tap leaves the image (TF SAME zero padding, reference resnet_utils.py:120-123); the per-lane
//     row offsets (256-pixel tiles) or centre offset + validity mask (512) are computed once;
//   * weights stream per step (one tap, or one kernel row of three taps, of one chunk) through a 3- or
//     4-deep LDS-DMA ring; the next chunk's slab is issued during the chunk's first step into the other
//     slab buffer.  One raw s_barrier per step; waits are counted (s_waitcnt vmcnt(N)) over the merged
//     in-order DMA stream.  SlabCfg holds the tile shape, chunk width, ring depth and taps per step.
// Covers the reference's conv2 call sites with stride 1: resnet_v2.py:130-132 via
// resnet_utils.conv2d_same (SAME padding, rate r).  Post-conv BN+ReLU folded (bias + ReLU).
// Below the tile: an EXPERIMENTAL launch that chains conv3 + shortcut behind it for block3 (four
// workgroups per image, in-launch hand-off), measured slower than two launches and not dispatched.
#include <cstdlib>
#include <type_traits>

#include "metro_common.h"

namespace metro {

typedef _Float16 half_t;
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

// (no relocatable device code: every translation unit keeps its own zero page)
__device__ __attribute__((aligned(16))) unsigned int g_zero_page_slab[4];

typedef __attribute__((address_space(3))) void lds_void3_t;

__device__ __forceinline__ void slab_dma16(const void* gsrc, unsigned lds_addr) {
    asm volatile(
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, off"
        :
        : "v"(gsrc), "s"(lds_addr));
}

template <int N>
__device__ __forceinline__ void slab_wait_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

// WM: 32-row cout tiles per wave (TM = WAVES_M*WM*32); pixels: TN = WAVES_N * WN * 32 (256 or 512)
// TPS: taps per K step.  1 = one (chunk, tap) per barrier; 3 = one kernel ROW (3 taps) per barrier for the
// 64-cout tiles, whose 8 MFMAs per wave per tap are too few to amortise the barrier + fragment-read latency.
// KC: input channels per chunk (slab and weight rows are KC*2 bytes).  64 for the 256-pixel tiles; the 512-pixel
// tiles take 32 so that two slabs (2 x 40 KiB) and a deeper weight ring fit next to each other.
// WS: weight ring depth (W(q + WS - 1) is issued during step q).
// Why 512-pixel tiles: a tile streams its cout-slice of ALL weights (TM x 9 C_in) once, whatever its pixel count, so
// the bytes a layer pulls from L2 into LDS are  tiles x (TM x 9 C_in x 2  +  slab);  block4's conv2 at batch 256 moved
// 1.58 GB that way in 260 us (6.1 TB/s, the rate every DMA-fed kernel here settles at) -- twice the pixels per tile
// halve the weight stream.
template <int WAVES_M_, int WAVES_N_, int WM_, int WN_, int SLAB_ROWS_, int SLAB_BUFS_ = 2, int TPS_ = 1, int KC_ = 64, int WS_ = 3>
struct SlabCfg {
    static constexpr int TPS = TPS_;
    static constexpr int KC = KC_;
    static constexpr int NKK = KC / 16;                      // MFMA k-steps per tap
    static constexpr int ROW_BYTES = KC * 2;
    static constexpr int RPI = 1024 / ROW_BYTES;             // LDS rows per DMA instruction (64 lanes x 16 B)
    static constexpr int SLAB_BUFS = SLAB_BUFS_;             // 1: single-chunk layers (c_in == 64): half the LDS, 2 blocks/CU
    static constexpr int WAVES_M = WAVES_M_, WAVES_N = WAVES_N_, WM = WM_, WN = WN_;
    static constexpr int NW = WAVES_M * WAVES_N, NT = 64 * NW;
    static constexpr int TM = WAVES_M * WM * 32, TN = WAVES_N * WN * 32;
    static constexpr int SLAB_ROWS = SLAB_ROWS_;             // padded to a multiple of RPI*NW
    static constexpr int SI = SLAB_ROWS / (RPI * NW);        // slab DMA instructions per wave per chunk
    static constexpr int WI = TM / (RPI * NW);               // weight DMA instructions per wave per tap
    static constexpr int SLAB_BYTES = SLAB_ROWS * ROW_BYTES;
    static constexpr int ZERO_OFF = SLAB_BUFS * SLAB_BYTES;  // 256-byte zero area behind the slab(s)
    static constexpr int W_OFF = ZERO_OFF + 256;
    static constexpr int W_STAGE_BYTES = TPS * TM * ROW_BYTES;
    static constexpr int W_STAGES = WS_;
    static constexpr int NSTEPS = 9 / TPS;                   // steps per chunk
    static constexpr int RAW_BYTES = W_OFF + W_STAGES * W_STAGE_BYTES;
    static constexpr int OUT_ROW_BYTES = TM * 2 + 16;
    static constexpr int OUT_BYTES = TN * OUT_ROW_BYTES;
    static constexpr int LDS_BYTES = RAW_BYTES > OUT_BYTES ? RAW_BYTES : OUT_BYTES;
    static_assert(TN == 256 || TN == 512, "slab kernel tiles 256 or 512 pixels");
    static_assert(KC == 64 || KC == 32, "64 or 32 channels per chunk");
    static_assert(TPS == 1 || TPS == 3, "one tap or one kernel row per step");
    static_assert(SLAB_ROWS % (RPI * NW) == 0 && TM % (RPI * NW) == 0, "loader mismatch");
    static_assert(W_STAGES >= 3 && W_STAGES - 2 <= NSTEPS - 1, "ring depth");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    // swizzle of the 16-byte chunk index inside a row (conflict-free ds_read_b128 for 32 consecutive rows at any
    // alignment: the 16 rows one LDS cycle serves cover the 256-byte bank span exactly once)
    __device__ static __forceinline__ int swz(int row) { return KC == 64 ? (row >> 1) & 7 : (row >> 2) & 3; }
    // the epilogue tile overlays slabs + zero area + weight ring (all idle by then)
};

template <int I, int N, class F>
__device__ __forceinline__ void slab_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        slab_static_for<I + 1, N>(f);
    }
}

// One tile: cout [n0, n0 + TM) of pixels [m0, m0 + TN).
template <class Cfg>
__device__ __forceinline__ void slab_tile(const ConvArgs& a, const half_t* __restrict__ in, const half_t* __restrict__ w,
                                          const float* __restrict__ bias, half_t* __restrict__ out, int halo, int m0, int n0,
                                          char* smem, int tid, int sgd) {
    constexpr int NW = Cfg::NW;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave / Cfg::WAVES_N;
    const int wave_n = wave % Cfg::WAVES_N;

    const int c_in = a.c_in;
    const int k_total = 9 * c_in;
    const int kc = c_in / Cfg::KC;                 // chunks (c_in % 64 == 0: launcher)
    // SUB-GRID mode (sgd = the layer's rate d > 1, round 6).  A rate-d 3x3 convolution with SAME padding on an H x W map is d*d
    // independent rate-1 convolutions on the (H/d) x (W/d) sub-images of the pixels with equal (y mod d, x mod d): a tap of
    // pixel (y, x) lands on (y +- d, x +- d), the same residue class, and leaves the image exactly when it leaves the sub-image.
    // The tile walks the pixels in SUB-IMAGE order -- virtual image n*d*d + (y mod d)*d + (x mod d), row y / d, column x / d -- so
    // its slab needs W/d rows of halo instead of d*W (rate 8 on a 64-wide map: 8 instead of 512: such layers used to fall to the
    // generic ring kernel, which re-fetches the activations for every tap); only the two places that turn a pixel index
    // into an address (the slab's DMA sources, the output rows) know about the permutation.  The k order is this kernel's
    // (chunk-major: every tap of a channel chunk, then the next chunk), as for every other layer it serves.
    const int gw = sgd ? a.w_out / sgd : a.w_out, gh = sgd ? a.h_out / sgd : a.h_out, gdil = sgd ? 1 : a.dil;
    const int hw = gh * gw;
    auto pixel_of = [&](int g) -> size_t {         // virtual (tile-order) pixel index -> NHWC pixel index
        if (!sgd) return (size_t)g;
        const int vimg = g / hw, rem = g - vimg * hw;
        const int vy = rem / gw, vx = rem - vy * gw;
        const int d2 = sgd * sgd;
        const int img = vimg / d2, sub = vimg - img * d2;
        const int sy = sub / sgd, sx = sub - sy * sgd;
        return ((size_t)img * a.h_out + (vy * sgd + sy)) * a.w_out + (vx * sgd + sx);
    };
    const half_t* zero = reinterpret_cast<const half_t*>(g_zero_page_slab);
    const unsigned smem_base = (unsigned)(size_t)(lds_void3_t*)smem;

    // zero area (read by taps that fall outside the image)
    if (tid < 16) reinterpret_cast<uint4*>(smem + Cfg::ZERO_OFF)[tid] = make_uint4(0, 0, 0, 0);

    // ---- DMA source coordinates -------------------------------------------------------------
    constexpr int RPI = Cfg::RPI, CPR = Cfg::ROW_BYTES / 16;    // rows per DMA instruction, 16-byte chunks per row
    const int lrow = lane / CPR, lch = lane % CPR;
    // slab rows: instruction i of this wave fills slab rows (i*NW + wave)*RPI + lrow
    const half_t* ssrc[Cfg::SI];
    int skoff[Cfg::SI];
    bool svalid[Cfg::SI];
#pragma unroll
    for (int i = 0; i < Cfg::SI; ++i) {
        const int srow = (i * NW + wave) * RPI + lrow;
        const int g = m0 - halo + srow;              // flattened pixel index
        // rows past the tile's own halo (a configuration's slab is sized for the largest halo it serves) come from the zero page
        svalid[i] = g >= 0 && g < a.m_total && srow < Cfg::TN + 2 * halo;
        ssrc[i] = in + pixel_of(svalid[i] ? g : 0) * c_in;
        skoff[i] = (lch ^ Cfg::swz(srow)) * 8;
    }
    const half_t* wsrc[Cfg::WI];
    int wkoff[Cfg::WI];
    bool wvalid[Cfg::WI];
#pragma unroll
    for (int i = 0; i < Cfg::WI; ++i) {
        const int row = (i * NW + wave) * RPI + lrow;
        const int co = n0 + row;
        wvalid[i] = co < a.c_out;
        wsrc[i] = w + (size_t)(wvalid[i] ? co : 0) * k_total;
        wkoff[i] = (lch ^ Cfg::swz(row)) * 8;
    }

    // running DMA pointers (c_in % 64 == 0 is required by the launcher): the slab advances KC
    // channels per chunk; a weight row advances c_in per tap and wraps to the next chunk after 9 taps
    const half_t* sptr[Cfg::SI];
#pragma unroll
    for (int i = 0; i < Cfg::SI; ++i) sptr[i] = svalid[i] ? ssrc[i] + skoff[i] : zero;
    const half_t* wptr[Cfg::WI];
#pragma unroll
    for (int i = 0; i < Cfg::WI; ++i) wptr[i] = wvalid[i] ? wsrc[i] + wkoff[i] : zero;
    const int w_tap_inc = c_in;                 // elements, tap t -> t+1
    const int w_chunk_inc = Cfg::KC - 8 * c_in; // elements, (c, 8) -> (c+1, 0)

    auto issue_slab_part = [&](int buf, int part, bool last_part) {     // next chunk -> slab buffer `buf`
        const unsigned base = __builtin_amdgcn_readfirstlane(smem_base + buf * Cfg::SLAB_BYTES + wave * 1024);
#pragma unroll
        for (int i = 0; i < Cfg::SI; ++i) {
            if (i % Cfg::NKK != part) continue;
            slab_dma16(sptr[i], base + i * NW * 1024);
            sptr[i] += svalid[i] ? Cfg::KC : 0;
        }
        (void)last_part;
    };
    // issues W of the step whose first tap is `t_issue` (TPS taps: t_issue .. t_issue+TPS-1), into ring slot `slot`;
    // `tt` = which of those taps, `part` = which of its NKK pieces; pointers advance after the last piece
    auto issue_w_part = [&](int slot, int t_issue, int part, int tt = 0) {
        const unsigned base = __builtin_amdgcn_readfirstlane(smem_base + Cfg::W_OFF + slot * Cfg::W_STAGE_BYTES +
                                                             tt * Cfg::TM * Cfg::ROW_BYTES + wave * 1024);
#pragma unroll
        for (int i = 0; i < Cfg::WI; ++i) {
            if (i % Cfg::NKK != part) continue;
            slab_dma16(wptr[i] + (wvalid[i] ? tt * w_tap_inc : 0), base + i * NW * 1024);
            if (tt == Cfg::TPS - 1)
                wptr[i] += wvalid[i] ? (t_issue + Cfg::TPS == 9 ? w_chunk_inc + (Cfg::TPS - 1) * w_tap_inc : Cfg::TPS * w_tap_inc) : 0;
        }
    };

    // ---- per-lane tap offsets into the slab (byte offset of the row, or the zero area) ---------
    const int frag_row = lane & 31, frag_half = lane >> 5;
    // TN == 256: the byte offset of every (pixel tile, tap) row in registers (-1 = outside the image).
    // TN == 512: four pixel tiles per wave would need 36 such registers next to 128 accumulators; there each pixel
    // tile keeps the offset of its centre row and a 9-bit validity mask, the tap displacement is a scalar.
    constexpr bool TABLE = Cfg::TN == 256;
    int boff[TABLE ? Cfg::WN : 1][9];
    int bbase[Cfg::WN];
    unsigned bmask[Cfg::WN];
#pragma unroll
    for (int j = 0; j < Cfg::WN; ++j) {
        const int t = (wave_n * Cfg::WN + j) * 32 + frag_row;        // tile-local pixel
        const int m = m0 + t;
        const int rem = m % hw;
        const int h = rem / gw, x = rem - h * gw;
        bbase[j] = (halo + t) * Cfg::ROW_BYTES;
        bmask[j] = 0;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dr = (tap / 3 - 1) * gdil, ds = (tap % 3 - 1) * gdil;
            const bool ok = m < a.m_total && (unsigned)(h + dr) < (unsigned)gh &&
                            (unsigned)(x + ds) < (unsigned)gw;
            if constexpr (TABLE) boff[j][tap] = ok ? (halo + t + dr * gw + ds) * Cfg::ROW_BYTES : -1;
            bmask[j] |= ok ? 1u << tap : 0u;
        }
    }

    unsigned arow[Cfg::WM];                                     // weight fragment rows (lane constants)
#pragma unroll
    for (int i = 0; i < Cfg::WM; ++i) {
        const int row = (wave_m * Cfg::WM + i) * 32 + frag_row;
        arow[i] = row * Cfg::ROW_BYTES + ((frag_half ^ Cfg::swz(row)) << 4);
    }

    floatx16 acc[Cfg::WM][Cfg::WN];
#pragma unroll
    for (int i = 0; i < Cfg::WM; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::WN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    constexpr int WS = Cfg::W_STAGES, NSTEPS = Cfg::NSTEPS, NKK = Cfg::NKK;
    constexpr int WSI = Cfg::WI * Cfg::TPS;        // weight DMA instructions per wave per step
    int ring = 0;                                   // slot of the chunk's first step (deep rings)
    auto slot_of = [&](int j) {
        if constexpr (WS == 3) return j % 3;        // 9 taps (3 rows) = whole turns: every chunk starts on slot 0
        else return (ring + j) % WS;
    };
    auto advance_ring = [&]() {
        if constexpr (WS != 3) ring = (ring + NSTEPS) % WS;
    };
    // LDS byte addresses of k-chunk `frag_half` of the fragment rows of tap TAP (weights in ring slot `wslot`, tap
    // `tt` of its step; pixels in slab `slab_buf`); k-step kk is the same address with bit 5.. flipped (chunk =
    // 2 kk ^ frag_half and the swizzle is an XOR): one VALU op per read
    auto frag_addr = [&](int TAP, int tt, int slab_buf, int wslot, unsigned (&pa)[Cfg::WM], unsigned (&pb)[Cfg::WN]) {
        const unsigned wl = Cfg::W_OFF + wslot * Cfg::W_STAGE_BYTES;
        const unsigned sl = slab_buf * Cfg::SLAB_BYTES;
#pragma unroll
        for (int i = 0; i < Cfg::WM; ++i) pa[i] = wl + tt * Cfg::TM * Cfg::ROW_BYTES + arow[i];
#pragma unroll
        for (int j = 0; j < Cfg::WN; ++j) {
            int bo;
            bool ok;
            if constexpr (TABLE) {
                bo = boff[j][TAP];
                ok = bo >= 0;
            } else {
                const int toff = __builtin_amdgcn_readfirstlane((((TAP / 3 - 1) * gw + (TAP % 3 - 1)) * gdil) * Cfg::ROW_BYTES);
                bo = bbase[j] + toff;
                ok = (bmask[j] >> TAP) & 1u;
            }
            pb[j] = ok ? sl + bo + ((frag_half ^ Cfg::swz(bo / Cfg::ROW_BYTES)) << 4) : (unsigned)Cfg::ZERO_OFF;
        }
    };

    // ---- prologue: slab(0), W(0) .. W(WS-2) --------------------------------------------------------
#pragma unroll
    for (int p = 0; p < NKK; ++p) issue_slab_part(0, p, false);
#pragma unroll
    for (int st = 0; st < WS - 1; ++st)
#pragma unroll
        for (int tt = 0; tt < Cfg::TPS; ++tt)
#pragma unroll
            for (int p = 0; p < NKK; ++p) issue_w_part(st, st * Cfg::TPS, p, tt);
    __syncthreads();   // zero area visible (plain ds_write above); DMA unaffected (asm, uncounted)

    // One step = TPS taps of one chunk.  The taps, what gets issued and the wait count are compile-time, so each
    // step is straight-line code: ds_read_b128 + MFMA + its share of the DMA issue.  The ring slot is a constant
    // for the 3-deep ring (9 taps = 3 turns), a rotating scalar for deeper ones.
    auto step = [&](auto tap_c, auto issue_slab_c, auto issue_w_c, auto wait_c, int slab_buf, int wslot) {
        constexpr int TAP0 = decltype(tap_c)::value;             // first tap of the step
        slab_wait_barrier<decltype(wait_c)::value>();
        const int islot = wslot == 0 ? WS - 1 : wslot - 1;     // slot of step q+WS-1 == slot of step q-1
#pragma unroll
        for (int tt = 0; tt < Cfg::TPS; ++tt) {
            unsigned pa[Cfg::WM], pb[Cfg::WN];
            frag_addr(TAP0 + tt, tt, slab_buf, wslot, pa, pb);
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                half8_t af[Cfg::WM], bf[Cfg::WN];
#pragma unroll
                for (int i = 0; i < Cfg::WM; ++i) af[i] = *reinterpret_cast<const half8_t*>(smem + (pa[i] ^ (kk << 5)));
#pragma unroll
                for (int j = 0; j < Cfg::WN; ++j) bf[j] = *reinterpret_cast<const half8_t*>(smem + (pb[j] ^ (kk << 5)));
#pragma unroll
                for (int i = 0; i < Cfg::WM; ++i)
#pragma unroll
                    for (int j = 0; j < Cfg::WN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
                if constexpr (decltype(issue_slab_c)::value) {
                    if (tt == 0) issue_slab_part(slab_buf ^ 1, kk, kk == NKK - 1);
                }
                if constexpr (decltype(issue_w_c)::value) issue_w_part(islot, (TAP0 + (WS - 1) * Cfg::TPS) % 9, kk, tt);
            }
        }
    };
    auto chunk_main = [&](int slab_buf) {       // not the last chunk: slab(c+1) during the first step, W always
        slab_static_for<0, NSTEPS>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            constexpr int wait = (WS - 2) * WSI + ((j >= 1 && j <= WS - 2) ? Cfg::SI : 0);
            step(std::integral_constant<int, j * Cfg::TPS>{}, std::integral_constant<bool, j == 0>{}, std::true_type{},
                 std::integral_constant<int, wait>{}, slab_buf, slot_of(j));
        });
        advance_ring();
    };
    auto chunk_last = [&](int slab_buf) {       // last chunk: no slab, W stops WS-1 steps before the end
        slab_static_for<0, NSTEPS>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            constexpr int left = NSTEPS - 1 - j;
            constexpr int wait = (left < WS - 2 ? left : WS - 2) * WSI;
            step(std::integral_constant<int, j * Cfg::TPS>{}, std::false_type{}, std::integral_constant<bool, (j + WS - 1 < NSTEPS)>{},
                 std::integral_constant<int, wait>{}, slab_buf, slot_of(j));
        });
    };
    for (int c = 0; c + 1 < kc; ++c) chunk_main(c & 1);      // (kc > 1 requires SLAB_BUFS == 2: launcher)
    chunk_last(Cfg::SLAB_BUFS == 2 ? (kc - 1) & 1 : 0);

    // ---- epilogue: (+bias, ReLU) -> LDS [pixel][cout] -> full-line stores ------------------------
    __syncthreads();
#pragma unroll
    for (int i = 0; i < Cfg::WM; ++i) {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const int col = (wave_m * Cfg::WM + i) * 32 + 8 * qd + 4 * frag_half;
            const int co = n0 + col;
            floatx4 bv = {0.f, 0.f, 0.f, 0.f};
            if (co < a.c_out) bv = *reinterpret_cast<const floatx4*>(bias + co);
#pragma unroll
            for (int j = 0; j < Cfg::WN; ++j) {
                const int prow = (wave_n * Cfg::WN + j) * 32 + frag_row;
                half4_t hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[i][j][4 * qd + e] + bv[e];
                    if (a.relu) v = fmaxf(v, 0.f);
                    hv[e] = (half_t)v;
                }
                *reinterpret_cast<half4_t*>(smem + prow * Cfg::OUT_ROW_BYTES + col * 2) = hv;
            }
        }
    }
    __syncthreads();
    constexpr int CPRO = Cfg::TM / 8;
    for (int idx = tid; idx < Cfg::TN * CPRO; idx += Cfg::NT) {
        const int prow = idx / CPRO;
        const int ch = idx - prow * CPRO;
        const int m = m0 + prow;
        const int co = n0 + ch * 8;
        if (m >= a.m_total || co >= a.c_out) continue;
        const uint4 v = *reinterpret_cast<const uint4*>(smem + prow * Cfg::OUT_ROW_BYTES + ch * 16);
        const size_t mo = pixel_of(m);
        if (co + 8 <= a.c_out) {
            store_out16<2>(out + mo * a.c_out + co, v);
        } else {
            const half8_t x = *reinterpret_cast<const half8_t*>(&v);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (co + e < a.c_out) out[mo * a.c_out + co + e] = x[e];
        }
    }
}

template <class Cfg>
__global__ __launch_bounds__(Cfg::NT) void conv3x3_f16_slab_kernel(
    ConvArgs a, const half_t* __restrict__ in, const half_t* __restrict__ w,
    const float* __restrict__ bias, half_t* __restrict__ out, int tiles_m, int halo, int sgd) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
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
    slab_tile<Cfg>(a, in, w, bias, out, halo, tile_n * Cfg::TN, tile_m * Cfg::TM, smem, threadIdx.x, sgd);
}

template <class Cfg>
static int launch_slab_cfg(const ConvArgs& a, const half_t* in, const half_t* w, const float* bias,
                           half_t* out, int halo, hipStream_t stream, int sgd = 0) {
    if (note_kernel("conv3x3_f16_slab<%dx%d,rows%d,bufs%d,tps%d,kc%d,ws%d>%s", Cfg::TM, Cfg::TN, Cfg::SLAB_ROWS, Cfg::SLAB_BUFS, Cfg::TPS,
                    Cfg::KC, Cfg::W_STAGES, sgd ? "+subgrid" : ""))
        return METRO_OK;
    auto kern = conv3x3_f16_slab_kernel<Cfg>;
    static PerDeviceInt attr_done;
    if (const int st = ensure_dyn_lds(reinterpret_cast<const void*>(kern), Cfg::LDS_BYTES, attr_done, "conv3x3_f16_slab")) return st;
    const int tiles_m = (a.c_out + Cfg::TM - 1) / Cfg::TM;
    const int tiles_n = (a.m_total + Cfg::TN - 1) / Cfg::TN;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(Cfg::NT), Cfg::LDS_BYTES, stream, a, in, w, bias,
                       out, tiles_m, halo, sgd);
    return launch_status("conv3x3_f16_slab");
}

//                    WAVES_M WAVES_N WM WN slab rows        tile, LDS
#ifndef METRO_SLAB_WS128
#define METRO_SLAB_WS128 3
#endif
#ifndef METRO_SLAB_WS512
#define METRO_SLAB_WS512 4
#endif
using Slab128r320 = SlabCfg<2, 4, 2, 2, 320, 2, 1, 64, METRO_SLAB_WS128>;   // 128 cout x 256 px, halo <= 32:  80 + 48 KiB
using Slab128r384 = SlabCfg<2, 4, 2, 2, 384>;   // halo <= 64:  96 + 48 KiB
#ifndef METRO_SLAB_WS64
#define METRO_SLAB_WS64 3
#endif
using Slab64r320 = SlabCfg<1, 8, 2, 1, 320, 2, 1, 64, METRO_SLAB_WS64>;    //  64 cout x 256 px
using Slab64r384 = SlabCfg<1, 8, 2, 1, 384>;
using Slab64r512 = SlabCfg<1, 8, 2, 1, 512>;    // halo <= 128: 128 + 24 KiB
using Slab64r320t3 = SlabCfg<1, 8, 2, 1, 320, 2, 3>;   // one kernel row per step: 80 + 72 KiB
using Slab64r384b1 = SlabCfg<1, 8, 2, 1, 384, 1>;  // single chunk (c_in == 64), halo <= 64: 48 + 24 KiB -> 2 blocks / CU
using Slab64r320b1 = SlabCfg<1, 8, 2, 1, 320, 1>;
using Slab128p512 = SlabCfg<2, 4, 2, 4, 640, 2, 1, 32, METRO_SLAB_WS512>;   // 128 cout x 512 px, 32-channel chunks, halo <= 64: 80 + 32 KiB (epilogue tile 136 KiB)

static bool slab_plain_ok(const MetroConvDesc& d) {
    // tiles are 256 consecutive pixels starting at column 0 (256 % W == 0): a tap (dr, ds<0) of a pixel
    // at x < dil is outside the image, so the slab needs dil*W rows of halo, not dil*W + dil
    if (256 % d.w_out != 0) return false;
    const int halo = d.dilation * d.w_out;
    if (d.c_out <= 64) return halo <= 128;
    return halo <= 64;   // (the 64-cout tiles used for few-tile layers allow more, kept equal for simplicity)
}

// The rate d of a layer that runs in sub-grid order (slab_tile), 0 for the plain tile order.  METRO_SLAB_SUBGRID: 0 = never,
// 1 = where the plain order's halo (d * W rows) exceeds the slab (stride 4: rate 4 / 8 on 64-wide maps; stride 8: rate 4 on 32-wide
// maps -- until round 6 these fell to the generic ring kernel), 2 = every dilated layer (A/B runs).
static int slab_subgrid_rate(const MetroConvDesc& d) {
    static const int mode = tuning_knob("METRO_SLAB_SUBGRID", 1);
    const int r = d.dilation;
    // (under the test switch metro_conv_b1_form(1) such layers run where they ran before round 6: the ring kernel, tap-major)
    if (mode == 0 || classic_forms_forced() || r <= 1 || d.h_out % r != 0 || d.w_out % r != 0 || 256 % (d.w_out / r) != 0) return 0;
    if (mode == 1 && slab_plain_ok(d)) return 0;
    return r;
}

bool conv3x3_slab_supported(const MetroConvDesc& d) {
    static const int enabled = tuning_knob("METRO_CONV_SLAB", 1);
    if (!enabled) return false;
    if (!(d.kh == 3 && d.kw == 3 && d.stride == 1 && d.h_in == d.h_out && d.w_in == d.w_out &&
          d.pad_top == d.dilation && d.pad_left == d.dilation && !d.has_prologue && !d.has_residual &&
          d.out_dtype == METRO_F16 && d.in_dtype == METRO_F16 && d.in_pix_stride == d.c_in &&
          d.c_in % 64 == 0 && d.c_out % 8 == 0))
        return false;
    const long m = (long)d.n * d.h_out * d.w_out;
    if (m < 256) return false;
    return slab_subgrid_rate(d) > 0 || slab_plain_ok(d);
}

int launch_conv3x3_slab(const MetroConvDesc& d, const void* in_, const void* w_, const float* bias,
                        void* out_, hipStream_t stream) {
    const ConvArgs a = make_conv_args(d);
    const half_t* in = static_cast<const half_t*>(in_);
    const half_t* w = static_cast<const half_t*>(w_);
    half_t* out = static_cast<half_t*>(out_);
    const int sgd = slab_subgrid_rate(d);
    const int halo = sgd ? d.w_out / sgd : d.dilation * d.w_out;       // sub-grid order: a rate-1 layer on (W / d)-wide sub-images
    // 64-cout tiles when 128-cout tiles would leave CUs without a block (256 CUs)
    const long blocks128 = (long)((d.c_out + 127) / 128) * ((a.m_total + 255) / 256);
    if (d.c_out <= 64 || blocks128 < 256) {
        if (d.c_in == 64 && halo <= 32) return launch_slab_cfg<Slab64r320b1>(a, in, w, bias, out, halo, stream, sgd);
        if (d.c_in == 64 && halo <= 64) return launch_slab_cfg<Slab64r384b1>(a, in, w, bias, out, halo, stream, sgd);
        static const int t3 = tuning_knob("METRO_SLAB_T3", 1);
        if (t3 && halo <= 32) return launch_slab_cfg<Slab64r320t3>(a, in, w, bias, out, halo, stream, sgd);
        if (halo <= 32) return launch_slab_cfg<Slab64r320>(a, in, w, bias, out, halo, stream, sgd);
        if (halo <= 64) return launch_slab_cfg<Slab64r384>(a, in, w, bias, out, halo, stream, sgd);
        return launch_slab_cfg<Slab64r512>(a, in, w, bias, out, halo, stream, sgd);
    }
    // 512-pixel tiles (half the weight stream per pixel) once they still give every CU a tile (A/B: 256 beats 512 as the
    // threshold at batch 128 and 256)
    static const int min512 = tuning_knob("METRO_SLAB512_MIN_TILES", 256);
    const long blocks512 = (long)((d.c_out + 127) / 128) * ((a.m_total + 511) / 512);
    // (sub-grid layers -- strides 4 and 8 -- reach 256 such tiles at 16 / 32 crops, the per-GPU shards of BASELINE.json configs[4] /
    // [3]; same-box A/B, profiles/r06_ab_subgrid_shards.txt: with 256-pixel tiles the sub-grid order is +1.0 % / +0.2 % SLOWER
    // than the ring kernel it replaces, with 512-pixel tiles -1.3 % / -0.6 % faster.  METRO_SLAB_SG512_MIN_N raises the batch from
    // which they may take them: the 32-channel chunks are another fp32 summation order)
    static const int sg512_min_n = tuning_knob("METRO_SLAB_SG512_MIN_N", 1);
    if (blocks512 >= min512 && (sgd == 0 || d.n >= sg512_min_n)) return launch_slab_cfg<Slab128p512>(a, in, w, bias, out, halo, stream, sgd);
    if (halo <= 32) return launch_slab_cfg<Slab128r320>(a, in, w, bias, out, halo, stream, sgd);
    return launch_slab_cfg<Slab128r384>(a, in, w, bias, out, halo, stream, sgd);
}

}  // namespace metro
