// The volumetric head in ONE launch: postnorm BN + ReLU (prologue) -> 1x1 logits convolution + bias -> per-joint
// softmax statistics over the tile's voxels, for heads of up to 160 channels (17- and 19-joint heads at depth 8).
//
// Reference: resnet_v2.py:229-236 (postnorm, logits), architectures.py:34 (cast to fp32), volumetric.py:227-235
// (reshape [D, J] -> softmax over (H, W, D) per joint -> decode_heatmap), tfu.py:466-499.
//
// Before, the logits (8.9 MB fp32 at RN50-s16 batch 64, 35.7 MB at stride 4) went to HBM and came back into a
// separate soft-argmax launch.  Here a block owns 64 pixels of ONE image and ALL head channels:
//   * GEMM [C <= 160 channels] x [64 pixels] x K (2048): LDS-DMA ring (5 stages of 28 KiB, counted vmcnt waits, one
//     s_barrier per 64-channel K step; four steps = 112 KiB in flight per CU: these blocks stream 0.8 MB each and
//     are latency bound), 8 waves = 4 K-quarters x 2 pixel tiles, v_mfma_f32_32x32x16_f16, fp32 accumulators; the
//     pre-activation is applied to the pixel fragments in fp16 like every other consumer of the residual stream; the
//     K-quarters are added through LDS in a fixed order;
//   * the fp32 logits tile stays in LDS ([pixel][channel], odd row pitch: conflict-free column reads).  Channel
//     c = d * J + j (volumetric.py:231).  Wave w takes joints w, w + 8, ...: a lane owns 1 pixel x D depths of the
//     joint, the tile maximum and the four sums (S, Sx, Sy, Sz) are folded across the 64 lanes with wave shuffles
//     (__shfl_xor -> DPP / ds_bpermute, no LDS round trip, no serial loop) -- max first, then exp(l - max): exact
//     two-pass softmax on the tile, one (m, S, Sx, Sy, Sz) record per (image, 64-pixel slab, joint);
//   * softargmax_finalize (pool_softargmax.hip) folds the slabs, decodes to millimetres, subtracts the root joint
//     and gathers the exported order, as for the two-launch path.
// The logits never exist in HBM.  Arithmetic: fp16 operands, fp32 accumulate, fp32 bias add, fp32 soft-argmax.
#include "metro_common.h"

namespace metro {

typedef _Float16 half_t;
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

__device__ __attribute__((aligned(16))) unsigned int g_zero_page_head[4];   // zero-initialised

namespace hd {
constexpr int TM = 160, MT = 5, TN = 64, BK = 64, NW = 8, NT = 512, STAGES = 5;
constexpr int KH = 4;                                  // K-quarters of every step (one 16-channel MFMA k step each)
constexpr int ROW_BYTES = BK * 2;
constexpr int STAGE_BYTES = (TM + TN) * ROW_BYTES;     // 28 KiB
constexpr int RING_BYTES = STAGES * STAGE_BYTES;       // 140 KiB: four K steps (112 KiB) in flight per CU
constexpr int LROW = 161;                              // fp32 words per logits-tile row (odd: conflict-free columns)
constexpr int LOGITS_BYTES = TN * LROW * 4;            // 41 KiB, overlays the ring after the K loop
constexpr int PRO_BYTES = 2 * 2048 * 2;
constexpr int LDS_BYTES = RING_BYTES + PRO_BYTES;
constexpr int LPS_A = 5, LPS_B = 2;                    // DMA instructions per K step: waves 0-3 (weights) / 4-7 (pixels)
static_assert(LOGITS_BYTES <= RING_BYTES, "logits tile must fit the ring");
}  // namespace hd

__device__ __forceinline__ int hd_swz(int row) { return (row >> 1) & 7; }
typedef __attribute__((address_space(3))) void hd_lds_void_t;

__device__ __forceinline__ void hd_dma16(const void* gsrc, unsigned lds_addr) {
    asm volatile(
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, off"
        :
        : "v"(gsrc), "s"(lds_addr));
}
template <int N>
__device__ __forceinline__ void hd_wait_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

struct HeadArgs {
    const half_t* x;          // [n * pixels][K] fp16, raw residual stream
    const half_t* w;          // [C][K] fp16
    const float* bias;        // [C]
    const half_t* pro_scale;  // [K] postnorm scale / shift (fp16)
    const half_t* pro_shift;
    float* partials;          // [n][slabs][J][5]: m, S, Sx, Sy, Sz
    float* logits_out;        // optional fp32 NHWC logits [n * pixels][C] (tests / layer dumps); NULL in the product path
    int K, C, J, D, side, pixels, slabs;
    // Heads wider than 160 channels (the 53-joint `merged` export: 424, reference data/datasets.py:142-154, main.py:119-127) run the
    // ring kernel once per GROUP of JG joints (blockIdx.z): a group's sub-head is the D * jg rows d * J + j0 + j' of the weight
    // matrix (all depths of its joints: the softmax of a joint needs nothing else), gathered row by row by the LDS-DMA sources,
    // its logits tile is [32][D * jg] in the same (d, j') order and its records go to joints j0 .. j0 + jg - 1.  JG = 0: one group.
    int JG;
};

__global__ __launch_bounds__(hd::NT) void head_f16_kernel(HeadArgs a) {
    using namespace hd;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = wave >> 1, wn = wave & 1;      // K-quarter of every step, 32-pixel tile
    const int slab = blockIdx.x, img = blockIdx.y;
    const int m0 = img * a.pixels + slab * TN;    // first pixel row of the tile
    const int K = a.K, nk = K / BK;
    const half_t* zero = reinterpret_cast<const half_t*>(g_zero_page_head);
    const unsigned smem_base = (unsigned)(size_t)(hd_lds_void_t*)smem;
    half_t* pro_lds = reinterpret_cast<half_t*>(smem + RING_BYTES);

    // ---- DMA sources: waves 0-3 bring the 160 weight rows (20 groups of 8 rows, 5 per wave), waves 4-7 the 64 pixel rows
    const int lrow = lane >> 3, lch = lane & 7;
    const half_t* src[LPS_A];
    int inc[LPS_A];
    unsigned ldsoff[LPS_A];
    const int nld = wave < 4 ? LPS_A : LPS_B;
#pragma unroll
    for (int i = 0; i < LPS_A; ++i) {
        if (wave < 4) {
            const int row = (i * 4 + wave) * 8 + lrow;
            const bool ok = row < a.C;
            src[i] = ok ? a.w + (size_t)row * K + ((lch ^ hd_swz(row)) * 8) : zero;
            inc[i] = ok ? BK : 0;
            ldsoff[i] = (i * 4 + wave) * 8 * ROW_BYTES;
        } else {
            const int g = (i * 4 + (wave - 4)) & 7;              // 8 groups of 8 pixel rows, 2 per wave
            const int row = g * 8 + lrow;
            src[i] = a.x + (size_t)(m0 + row) * K + ((lch ^ hd_swz(row)) * 8);
            inc[i] = BK;
            ldsoff[i] = TM * ROW_BYTES + g * 8 * ROW_BYTES;
        }
    }
    auto issue_step = [&](int slot) {
        const unsigned base = __builtin_amdgcn_readfirstlane(smem_base + slot * STAGE_BYTES);
#pragma unroll
        for (int i = 0; i < LPS_A; ++i) {
            if (i < nld) {                        // wave-uniform: the pixel waves issue 2, the weight waves 5
                hd_dma16(src[i], base + ldsoff[i]);
                src[i] += inc[i];
            }
        }
    };

    floatx16 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const int frag_row = lane & 31, frag_half = lane >> 5;

    // ---- ring prologue, postnorm table ----
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nk) issue_step(s);
    for (int c = tid * 8; c < K; c += NT * 8) {
        *reinterpret_cast<uint4*>(pro_lds + c) = *reinterpret_cast<const uint4*>(a.pro_scale + c);
        *reinterpret_cast<uint4*>(pro_lds + 2048 + c) = *reinterpret_cast<const uint4*>(a.pro_shift + c);
    }
    __syncthreads();      // table visible (drains the DMAs issued so far as well: once per block)

    const int brow = wn * 32 + frag_row;
    const int b_off = TM * ROW_BYTES + brow * ROW_BYTES;
    const int b_sw = hd_swz(brow);
    auto compute_step = [&](int slot, int k0) {
        const char* wl = smem + slot * STAGE_BYTES;
        {
            const int kk = kh;
            const int chunk = kk * 2 + frag_half;
            half8_t bf = *reinterpret_cast<const half8_t*>(wl + b_off + ((chunk ^ b_sw) << 4));
            const half8_t sc = *reinterpret_cast<const half8_t*>(pro_lds + k0 + chunk * 8);
            const half8_t sh = *reinterpret_cast<const half8_t*>(pro_lds + 2048 + k0 + chunk * 8);
            const half8_t z = {};
            bf = __builtin_elementwise_max(bf * sc + sh, z);      // postnorm BN + ReLU, fp16 FMA (resnet_v2.py:229)
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int row = i * 32 + frag_row;
                const half8_t af = *reinterpret_cast<const half8_t*>(wl + row * ROW_BYTES + ((chunk ^ hd_swz(row)) << 4));
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc[i], 0, 0, 0);
            }
        }
    };
    // steady state: STAGES-2 younger steps stay in flight; per-wave DMA counts differ between the two loader groups
    const int n_main = nk - (STAGES - 1);
    int slot = 0, islot = STAGES - 1;
    for (int k = 0; k < n_main; ++k) {
        if (wave < 4) hd_wait_barrier<(STAGES - 2) * LPS_A>(); else hd_wait_barrier<(STAGES - 2) * LPS_B>();
        issue_step(islot);
        compute_step(slot, k * BK);
        slot = slot + 1 == STAGES ? 0 : slot + 1;
        islot = islot + 1 == STAGES ? 0 : islot + 1;
    }
    for (int k = n_main < 0 ? 0 : n_main; k < nk; ++k) {      // drain
        const int ahead = nk - 1 - k;
        if (ahead >= 3) { if (wave < 4) hd_wait_barrier<3 * LPS_A>(); else hd_wait_barrier<3 * LPS_B>(); }
        else if (ahead == 2) { if (wave < 4) hd_wait_barrier<2 * LPS_A>(); else hd_wait_barrier<2 * LPS_B>(); }
        else if (ahead == 1) { if (wave < 4) hd_wait_barrier<LPS_A>(); else hd_wait_barrier<LPS_B>(); }
        else hd_wait_barrier<0>();
        compute_step(slot, k * BK);
        slot = slot + 1 == STAGES ? 0 : slot + 1;
    }
    __syncthreads();      // every wave is done with the ring: the logits tile overlays it

    // ---- the four K-quarters -> fp32 logits tile [pixel][channel] (+bias), added in a FIXED order (bit-reproducible) ----
    float* lt = reinterpret_cast<float*>(smem);
    const int prow = wn * 32 + frag_row;
#pragma unroll
    for (int r = KH - 1; r >= 0; --r) {
        if (kh == r) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int c = i * 32 + 8 * q + 4 * frag_half + e;
                        float v = acc[i][4 * q + e];
                        if (r != KH - 1) v += lt[prow * LROW + c];
                        if (r == 0) v += c < a.C ? a.bias[c] : 0.f;
                        lt[prow * LROW + c] = v;
                    }
        }
        __syncthreads();
    }
    if (a.logits_out != nullptr) {                // layer dump for the tests: coalesced rows of C floats
        for (int idx = tid; idx < TN * a.C; idx += NT) {
            const int p = idx / a.C, c = idx - p * a.C;
            a.logits_out[(size_t)(m0 + p) * a.C + c] = lt[p * LROW + c];
        }
    }

    // ---- per-joint softmax statistics of the tile: wave shuffles ----
    // lane <-> pixels (lane, lane + 64); coordinates in fp32 like tf.linspace (tfu.py:481)
    const float step_s = 1.0f / (float)(a.side - 1);
    const float step_d = 1.0f / (float)(a.D - 1);
    const int pim = slab * TN + lane;                     // pixel index inside the image
    const int py = pim / a.side, px = pim - py * a.side;
    const float cx = (float)px * step_s, cy = (float)py * step_s;
    for (int j = wave; j < a.J; j += NW) {
        float m = -INFINITY;
        for (int d = 0; d < a.D; ++d) m = fmaxf(m, lt[lane * LROW + d * a.J + j]);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        float s = 0.f, sz = 0.f;
        for (int d = 0; d < a.D; ++d) {
            const float e = __expf(lt[lane * LROW + d * a.J + j] - m);
            s += e;
            sz += e * ((float)d * step_d);
        }
        float sx = s * cx, sy = s * cy;                   // the pixel's coordinates are the lane's
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            s += __shfl_xor(s, o, 64);
            sx += __shfl_xor(sx, o, 64);
            sy += __shfl_xor(sy, o, 64);
            sz += __shfl_xor(sz, o, 64);
        }
        if (lane == 0) {
            float* o5 = a.partials + (((size_t)img * a.slabs + slab) * a.J + j) * 5;
            o5[0] = m; o5[1] = s; o5[2] = sx; o5[3] = sy; o5[4] = sz;
        }
    }
}

// Per-joint softmax statistics of a [32 pixels][LROW] fp32 logits tile, for the 32 lanes of a wave half (lane = pixel; `row` = the
// lane's tile row): joints j0, j0 + js, ... of the tile.  Exact two-pass softmax over the tile's 32 x D voxels; the 32 lanes fold
// with xor butterflies 16 .. 1 (ds_swizzle: no address arithmetic, never crossing halves).  Depth 8 (every BASELINE config): the
// D logits of FIVE joints are read at once and their five reductions run side by side -- the serial form (a dependent LDS read per
// depth, a dependent shuffle per fold step, one joint at a time) cost 17 of the 69 us of the 256-pixel head.
template <int O>
__device__ __forceinline__ float hd_xor32(float v) {
    return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), (O << 10) | 0x1F));
}
template <typename F>
__device__ __forceinline__ void hd_tile_stats(const float* row, int J, int D, int j0, int js, float cx, float cy, float step_d,
                                              bool writer, F&& record) {
    constexpr int NJ = 5;
    if (D == 8) {
        for (int jb = j0; jb < J; jb += NJ * js) {
            float v[NJ][8], m[NJ], s[NJ], sx[NJ], sy[NJ], sz[NJ];
#pragma unroll
            for (int q = 0; q < NJ; ++q) {
                const int j = jb + q * js < J ? jb + q * js : jb;         // past the last joint: joint jb again (not recorded)
#pragma unroll
                for (int d = 0; d < 8; ++d) v[q][d] = row[d * J + j];
            }
#pragma unroll
            for (int q = 0; q < NJ; ++q) {
                m[q] = -INFINITY;
#pragma unroll
                for (int d = 0; d < 8; ++d) m[q] = fmaxf(m[q], v[q][d]);
            }
#define HD_FOLD_MAX(O)                                                        \
    _Pragma("unroll") for (int q = 0; q < NJ; ++q) m[q] = fmaxf(m[q], hd_xor32<O>(m[q]))
            HD_FOLD_MAX(16); HD_FOLD_MAX(8); HD_FOLD_MAX(4); HD_FOLD_MAX(2); HD_FOLD_MAX(1);
#undef HD_FOLD_MAX
#pragma unroll
            for (int q = 0; q < NJ; ++q) {
                s[q] = 0.f; sz[q] = 0.f;
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    const float e = __expf(v[q][d] - m[q]);
                    s[q] += e;
                    sz[q] += e * ((float)d * step_d);
                }
                sx[q] = s[q] * cx; sy[q] = s[q] * cy;                     // the pixel's coordinates are the lane's
            }
#define HD_FOLD_SUM(O)                                                        \
    _Pragma("unroll") for (int q = 0; q < NJ; ++q) {                          \
        s[q] += hd_xor32<O>(s[q]); sx[q] += hd_xor32<O>(sx[q]);               \
        sy[q] += hd_xor32<O>(sy[q]); sz[q] += hd_xor32<O>(sz[q]);             \
    }
            HD_FOLD_SUM(16); HD_FOLD_SUM(8); HD_FOLD_SUM(4); HD_FOLD_SUM(2); HD_FOLD_SUM(1);
#undef HD_FOLD_SUM
            if (writer) {
#pragma unroll
                for (int q = 0; q < NJ; ++q)
                    if (jb + q * js < J) record(jb + q * js, m[q], s[q], sx[q], sy[q], sz[q]);
            }
        }
        return;
    }
    for (int j = j0; j < J; j += js) {
        float m = -INFINITY;
        for (int d = 0; d < D; ++d) m = fmaxf(m, row[d * J + j]);
        m = fmaxf(m, hd_xor32<16>(m)); m = fmaxf(m, hd_xor32<8>(m)); m = fmaxf(m, hd_xor32<4>(m));
        m = fmaxf(m, hd_xor32<2>(m)); m = fmaxf(m, hd_xor32<1>(m));
        float s = 0.f, sz = 0.f;
        for (int d = 0; d < D; ++d) {
            const float e = __expf(row[d * J + j] - m);
            s += e;
            sz += e * ((float)d * step_d);
        }
        float sx = s * cx, sy = s * cy;
#define HD_FOLD1(O) s += hd_xor32<O>(s); sx += hd_xor32<O>(sx); sy += hd_xor32<O>(sy); sz += hd_xor32<O>(sz)
        HD_FOLD1(16); HD_FOLD1(8); HD_FOLD1(4); HD_FOLD1(2); HD_FOLD1(1);
#undef HD_FOLD1
        if (writer) record(j, m, s, sx, sy, sz);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same head with 256-PIXEL tiles, for heads with at least one such tile per CU (RN50-s4 at 16 crops, the stride-16 nets
// from 256 crops on).  A 64-pixel tile re-streams the head's 557 KB of weights for 0.14 GFLOP (44 FLOP/B through the L2 -> LDS
// path: the C5 head took 142 us where its HBM bytes need 13); with 256 pixels the weights stream once per FOUR times the
// pixels.  Geometry: 8 waves = 8 x 32 pixels, every wave all <= 160 channels (5 accumulator tiles), the WHOLE K per wave:
// no K-quarters to add up.  Five 26 KiB stages (160 weight rows + 256 pixel rows of 32 channels: four K steps = 104 KiB in
// flight per CU -- with two 52 KiB stages every step paid a whole memory latency: 112 us); then the block walks its
// four 64-pixel chunks: the two owning waves put their fp32 logits (+ bias) into the LDS tile, all eight waves take the
// per-joint softmax statistics exactly as above -- ONE record per (image, 64-pixel slab, joint), the same partials layout
// and slab count as the 64-pixel kernel, so softargmax_finalize does not care which one ran.
namespace hd2 {
constexpr int TM = 160, MT = 5, TN = 256, BK = 32, NW = 8, NT = 512, STAGES = 5;
constexpr int ROW_BYTES = BK * 2;                      // 64: four 16-byte chunks, swizzled by (row >> 2) & 3
constexpr int STAGE_BYTES = (TM + TN) * ROW_BYTES;     // 26 KiB
constexpr int RING_BYTES = STAGES * STAGE_BYTES;       // 130 KiB: four K steps (104 KiB) in flight per CU
constexpr int LROW = 161;
constexpr int LOGITS_BYTES = 64 * LROW * 4;            // fp32 logits of one 64-pixel chunk: overlays the ring after the K loop
constexpr int PRO_OFF = RING_BYTES;
constexpr int LDS_BYTES = PRO_OFF + 2 * 2048 * 2;
constexpr int GA = TM / 16, GB = TN / 16;              // DMA instructions (16 rows each) per K step: 10 weight + 16 pixel groups
constexpr int WTILE_BYTES = 4 * 32 * LROW * 4;           // four wave-private [32][161] fp32 tiles per round
static_assert(LDS_BYTES <= 160 * 1024 && LOGITS_BYTES <= RING_BYTES && WTILE_BYTES <= RING_BYTES && GA <= 2 * NW && GB == 2 * NW, "LDS / loader split");
}  // namespace hd2

__device__ __forceinline__ int hd2_swz(int row) { return (row >> 2) & 3; }

__global__ __launch_bounds__(hd2::NT) void head_f16_kernel256(HeadArgs a) {
    using namespace hd2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x, img = blockIdx.y;
    const int m0 = img * a.pixels + tile * TN;
    const int K = a.K, nk = K / BK;
    const half_t* zero = reinterpret_cast<const half_t*>(g_zero_page_head);
    const unsigned smem_base = (unsigned)(size_t)(hd_lds_void_t*)smem;
    half_t* pro_lds = reinterpret_cast<half_t*>(smem + PRO_OFF);

    // ---- DMA sources per K step (one instruction = 16 rows x 64 B): weight groups g = wave, wave + 8 (< 10), pixel groups
    //      wave, wave + 8: waves 0-1 issue four per step, waves 2-7 three
    const int lrow = lane >> 2, lch = lane & 3;
    const int na = wave < GA - NW ? 2 : 1;
    const half_t* srcw[2];
    const half_t* srcx[2];
    int incw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave + NW * i) * 16 + lrow;
        const bool ok = row < a.C;
        srcw[i] = ok ? a.w + (size_t)row * K + ((lch ^ hd2_swz(row)) * 8) : zero;
        incw[i] = ok ? BK : 0;
        const int prow = (wave + NW * i) * 16 + lrow;
        srcx[i] = a.x + (size_t)(m0 + prow) * K + ((lch ^ hd2_swz(prow)) * 8);
    }
    auto issue_step = [&](int slot) {
        const unsigned base = __builtin_amdgcn_readfirstlane(smem_base + slot * STAGE_BYTES);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i < na) {
                hd_dma16(srcw[i], base + (wave + NW * i) * 16 * ROW_BYTES);
                srcw[i] += incw[i];
            }
            hd_dma16(srcx[i], base + TM * ROW_BYTES + (wave + NW * i) * 16 * ROW_BYTES);
            srcx[i] += BK;
        }
    };

    floatx16 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const int frag_row = lane & 31, frag_half = lane >> 5;

#pragma unroll
    for (int st = 0; st < STAGES - 1; ++st)
        if (st < nk) issue_step(st);
    for (int c = tid * 8; c < K; c += NT * 8) {
        *reinterpret_cast<uint4*>(pro_lds + c) = *reinterpret_cast<const uint4*>(a.pro_scale + c);
        *reinterpret_cast<uint4*>(pro_lds + 2048 + c) = *reinterpret_cast<const uint4*>(a.pro_shift + c);
    }
    __syncthreads();      // table visible (drains the DMAs issued so far as well: once per block)

    const int brow = wave * 32 + frag_row;
    const int b_off = TM * ROW_BYTES + brow * ROW_BYTES;
    const int b_sw = hd2_swz(brow);
    auto compute_step = [&](int slot, int k0) {
        const char* wl = smem + slot * STAGE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int chunk = kk * 2 + frag_half;
            half8_t bf = *reinterpret_cast<const half8_t*>(wl + b_off + ((chunk ^ b_sw) << 4));
            const half8_t sc = *reinterpret_cast<const half8_t*>(pro_lds + k0 + chunk * 8);
            const half8_t sh = *reinterpret_cast<const half8_t*>(pro_lds + 2048 + k0 + chunk * 8);
            const half8_t z = {};
            bf = __builtin_elementwise_max(bf * sc + sh, z);      // postnorm BN + ReLU, fp16 FMA (resnet_v2.py:229)
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int row = i * 32 + frag_row;
                const half8_t af = *reinterpret_cast<const half8_t*>(wl + row * ROW_BYTES + ((chunk ^ hd2_swz(row)) << 4));
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc[i], 0, 0, 0);
            }
        }
    };
    // steady state: the three younger steps stay in flight (counted per wave: 4 or 3 DMA instructions per step)
    const int n_main = nk - (STAGES - 1);
    int slot = 0, islot = STAGES - 1;
    for (int k = 0; k < n_main; ++k) {
        if (na == 2) hd_wait_barrier<3 * 4>(); else hd_wait_barrier<3 * 3>();
        issue_step(islot);
        compute_step(slot, k * BK);
        slot = slot + 1 == STAGES ? 0 : slot + 1;
        islot = islot + 1 == STAGES ? 0 : islot + 1;
    }
    for (int k = n_main < 0 ? 0 : n_main; k < nk; ++k) {      // drain
        const int ahead = nk - 1 - k;
        if (ahead >= 3) { if (na == 2) hd_wait_barrier<12>(); else hd_wait_barrier<9>(); }
        else if (ahead == 2) { if (na == 2) hd_wait_barrier<8>(); else hd_wait_barrier<6>(); }
        else if (ahead == 1) { if (na == 2) hd_wait_barrier<4>(); else hd_wait_barrier<3>(); }
        else hd_wait_barrier<0>();
        compute_step(slot, k * BK);
        slot = slot + 1 == STAGES ? 0 : slot + 1;
    }
    __syncthreads();      // every wave is done with the ring: the logits tile overlays it

    // ---- per-joint softmax statistics, one record per (image, 32-pixel slab, joint): every wave works on ITS OWN 32 pixels.
    //      (First version: four 64-pixel chunks through one shared logits tile, joints dealt to waves, lanes = pixels: 40 of the
    //      launch's 98 us -- two barriers per chunk, six of eight waves idle while two write, serial 64-lane folds.)
    //      The wave's fp32 logits (+ bias) go to a wave-private LDS tile [32 pixels][161]; lane (pixel = lane & 31, half) then takes
    //      joints half, half + 2, ...: exact two-pass softmax over the group's 32 x D voxels, folds across the 32 lanes of its half
    //      with xor shuffles 16 .. 1 (never crossing halves).  Two rounds (waves 0-3, then 4-7) share four tile regions.
    float* lt = reinterpret_cast<float*>(smem) + (wave & 3) * (32 * LROW);
    const float step_s = 1.0f / (float)(a.side - 1);
    const float step_d = 1.0f / (float)(a.D - 1);
    for (int round = 0; round < 2; ++round) {
        if ((wave >> 2) == round) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c0 = i * 32 + 8 * q + 4 * frag_half;
                    floatx4 bv = {0.f, 0.f, 0.f, 0.f};
                    if (c0 < a.C) bv = *reinterpret_cast<const floatx4*>(a.bias + c0);      // C % 4 == 0 (plan)
#pragma unroll
                    for (int e = 0; e < 4; ++e) lt[frag_row * LROW + c0 + e] = acc[i][4 * q + e] + bv[e];
                }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the wave's own LDS writes are complete
            if (a.logits_out != nullptr) {
                for (int idx = lane; idx < 32 * a.C; idx += 64) {
                    const int p = idx / a.C, c = idx - p * a.C;
                    a.logits_out[(size_t)(m0 + wave * 32 + p) * a.C + c] = lt[p * LROW + c];
                }
            }
            const int pim = tile * TN + wave * 32 + frag_row;     // pixel index inside the image
            const int py = pim / a.side, px = pim - py * a.side;
            const float cx = (float)px * step_s, cy = (float)py * step_s;
            const int slab = tile * (TN / 32) + wave;
            float* rec = a.partials + ((size_t)img * a.slabs + slab) * a.J * 5;
            hd_tile_stats(lt + frag_row * LROW, a.J, a.D, frag_half, 2, cx, cy, step_d, frag_row == 0,
                          [&](int j, float m, float s_, float sx, float sy, float sz) {
                              float* o5 = rec + j * 5;
                              o5[0] = m; o5[1] = s_; o5[2] = sx; o5[3] = sy; o5[4] = sz;
                          });
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 4: the head in 128-byte row pieces with K-split wave tiles, for every tile width (head_f16_ring_kernel<TN, KSPLIT, WROWS>).
// What the kernels above are bound by (NOTES_dead_ends.md, rounds 3-4): (1) 32-channel K steps are 64-byte row pieces -- the L2s answer a
// roughly constant REQUEST rate, half-line requests halve the bytes (8.7 against 13.8 TB/s); (2) a wave tile of 160 channels x 32
// pixels over the whole K reads 7 KB of fragments per 5 MFMAs: more LDS time than MFMA time; (3) a barrier per step with dependent
// fragment reads behind it; (4) a serial statistics phase (a dependent LDS read per depth, a dependent shuffle per fold step).  Here:
//   * K steps of 64 channels in 128-byte row pieces, as many stages as fit 150 KiB (WROWS weight rows -- accumulator rows past
//     them read into the pixel rows: channels nobody looks at -- + TN pixel rows: 3 x 50 KiB at 256 pixels, 4 x 34-36 at 128,
//     5 x 26-28 at 64), all but one in flight; the postnorm table and the bias come by LDS-DMA too, ahead of stage 0;
//   * 8 waves = (8 / KSPLIT) pixel groups x KSPLIT K-parts of EVERY step (part h: k-steps h (4 / KSPLIT) ...): wave tiles of
//     160 x 64 over two K-halves (256 pixels: 7 KB of fragment reads per TEN MFMAs), 160 x 64 over four K-quarters (128), 160 x 32
//     over four K-quarters (64);
//   * weight fragments are read two MFMA groups ahead of their use, pixel fragments and table rows a whole phase ahead; the barrier
//     of a step sits after the third group of its last phase (every read of the stage is out by then) and the reads of the next
//     stage follow it: no MFMA waits for an LDS read of its own group;
//   * the K-parts are added through LDS tiles of the pixel group in a fixed order (((h0 + h1) + h2) + h3) + bias, 32 pixels at a
//     time; the group's waves share the per-joint statistics of the tile (joints 2 h + lane half, step 2 KSPLIT), five joints at
//     a time (hd_tile_stats).
// One record per (image, 32-pixel slab, joint) whatever the tile width.
template <int TN_, int KSPLIT_, int WROWS_>
struct HeadRing {
    static constexpr int TN = TN_, KSPLIT = KSPLIT_, WROWS = WROWS_;
    static constexpr int MT = 5, BK = 64, NW = 8, NT = 512;
    static constexpr int PG = NW / KSPLIT;                 // pixel groups
    static constexpr int PT = TN / 32 / PG;                // 32-pixel tiles per wave
    static constexpr int NP = 4 / KSPLIT;                  // phases (16-channel k-steps) per wave and K step
    static constexpr int ROW_BYTES = BK * 2;               // 128
    static constexpr int STAGE_BYTES = (WROWS + TN) * ROW_BYTES;
    static constexpr int STAGES = (150 * 1024) / STAGE_BYTES > 5 ? 5 : (150 * 1024) / STAGE_BYTES;
    static constexpr int RING_BYTES = STAGES * STAGE_BYTES;
    static constexpr int LROW = 161;
    static constexpr int WTILE_BYTES = PG * (KSPLIT - 1) * 32 * LROW * 4;    // a group's KSPLIT - 1 [32][161] fp32 tiles
    static constexpr int PRO_OFF = RING_BYTES;
    static constexpr int BIAS_OFF = PRO_OFF + 2 * 2048 * 2;                  // 256 fp32 bias words (one LDS-DMA instruction)
    static constexpr int LDS_BYTES = BIAS_OFF + 1024;
    static constexpr int GA = WROWS / 8, GB = TN / 8;      // DMA instructions (8 rows x 128 B) per K step
    static constexpr int NA_HI = (GA + NW - 1) / NW, NA_LO = GA / NW, NA_REM = GA % NW;   // waves < NA_REM issue NA_HI weight groups
    static constexpr int NB = GB / NW;                     // pixel groups per wave
    static_assert(PT >= 1 && PT <= 2 && NP >= 1 && STAGES >= 3 && LDS_BYTES <= 160 * 1024 && WTILE_BYTES <= RING_BYTES &&
                  NA_HI <= 3 && NB >= 1 && NB <= 4 && GB % NW == 0 && WROWS % 8 == 0, "head ring geometry");
};

// one LDS-DMA wave-instruction, source = wave-uniform base + per-lane byte offset, destination (lds_base + LDS_IMM) + 16 l
template <int LDS_IMM>
__device__ __forceinline__ void hd_dma16s(const void* sbase, unsigned voff, unsigned lds_base) {
    asm volatile(
        "s_add_u32 m0, %2, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %1"
        :
        : "v"(voff), "s"(sbase), "s"(lds_base), "n"(LDS_IMM)
        : "scc");
}
// the emitted order of a segment: its LDS reads, then its MFMAs with the VALU work between them
template <int NREAD, int NMFMA, int NVALU>
__device__ __forceinline__ void hd_pin() {
    __builtin_amdgcn_sched_group_barrier(0x100, NREAD, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, NVALU, 0);
    if constexpr (NMFMA == 2) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, NVALU, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int TN_, int KSPLIT_, int WROWS_>
__global__ __launch_bounds__(512, 2) void head_f16_ring_kernel(HeadArgs a) {
    using G = HeadRing<TN_, KSPLIT_, WROWS_>;
    constexpr int TN = G::TN, KSPLIT = G::KSPLIT, WROWS = G::WROWS, MT = G::MT, BK = G::BK, NW = G::NW, PT = G::PT, NP = G::NP;
    constexpr int ROW_BYTES = G::ROW_BYTES, STAGE_BYTES = G::STAGE_BYTES, STAGES = G::STAGES, RING_BYTES = G::RING_BYTES, LROW = G::LROW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pg = wave / KSPLIT, h = wave % KSPLIT;      // pixel group, K-part of every step (the group shares LDS logits tiles)
    const int tile = blockIdx.x, img = blockIdx.y;
    const int m0 = img * a.pixels + tile * TN;
    const int K = a.K, nk = K / BK;
    const unsigned smem_base = (unsigned)(size_t)(hd_lds_void_t*)smem;
    half_t* pro_lds = reinterpret_cast<half_t*>(smem + G::PRO_OFF);
    // joint group of this block: joints j0 .. j0 + jg - 1, sub-head rows c' = d * jg + j' <-> weight / bias row d * J + j0 + j'
    const int j0 = a.JG > 0 ? (int)blockIdx.z * a.JG : 0;
    const int jg = a.JG > 0 ? (a.J - j0 < a.JG ? a.J - j0 : a.JG) : a.J;
    const int cg = jg * a.D;                               // channels of the sub-head
    auto src_row = [&](int c) { return a.JG > 0 ? (c / jg) * a.J + j0 + c % jg : c; };
    if (a.JG > 0) {
        // the bias of a sub-head is a gather: ordinary loads, FIRST (the compiler waits for them with vmcnt(0): nothing of the ring is
        // in flight yet), visible behind the prologue's barrier
        float* bl = reinterpret_cast<float*>(smem + G::BIAS_OFF);
        if (tid < 256) bl[tid] = tid < cg ? a.bias[src_row(tid)] : 0.f;
    }

    // ---- DMA sources per K step (one instruction = 8 rows x 128 B, lane: row l >> 3, physical chunk l & 7): pixel groups
    //      wave + 8 i (NB each), weight groups wave + 8 i < GA (the first GA % 8 waves issue one more).  Rows past the head's
    //      channels repeat its last row (channels nobody looks at).
    const int lrow = lane >> 3, lch = lane & 7;
    const bool na_hi = G::NA_REM != 0 && wave < G::NA_REM;
    const int na = na_hi ? G::NA_HI : G::NA_LO;
    unsigned voffw[3], voffx[4];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int row = (wave + NW * i) * 8 + lrow;
        const int srow = src_row(row < cg ? row : cg - 1);
        voffw[i] = (unsigned)(srow * K + ((lch ^ hd_swz(row)) * 8)) * 2u;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int prow = ((wave + NW * i) * 8 + lrow) & (TN - 1);
        voffx[i] = (unsigned)(prow * K + ((lch ^ hd_swz(prow)) * 8)) * 2u;
    }
    const half_t* xbase = a.x + (size_t)m0 * K;
    const unsigned lds_wave = __builtin_amdgcn_readfirstlane(smem_base + wave * 8 * ROW_BYTES);
    auto issue_step = [&](int slot_off, int kt) {       // K step kt into the stage at byte offset slot_off
        const unsigned base = lds_wave + slot_off;
        const half_t* xs = xbase + kt * BK;
        const half_t* ws = a.w + kt * BK;
        hd_dma16s<WROWS * ROW_BYTES + 0 * 8192>(xs, voffx[0], base);      // the pixel rows first: first touch from HBM
        if constexpr (G::NB > 1) hd_dma16s<WROWS * ROW_BYTES + 1 * 8192>(xs, voffx[1], base);
        if constexpr (G::NB > 2) hd_dma16s<WROWS * ROW_BYTES + 2 * 8192>(xs, voffx[2], base);
        if constexpr (G::NB > 3) hd_dma16s<WROWS * ROW_BYTES + 3 * 8192>(xs, voffx[3], base);
        hd_dma16s<0 * 8192>(ws, voffw[0], base);
        hd_dma16s<1 * 8192>(ws, voffw[1], base);
        if (na == 3) hd_dma16s<2 * 8192>(ws, voffw[2], base);
    };
    static_assert(G::NA_LO == 2, "two weight groups per wave at least, a third for the first GA % 8 waves");
    constexpr int NW_LO = G::NA_LO + G::NB, NW_HI = G::NA_HI + G::NB;     // DMA instructions per wave and K step

    floatx16 acc[PT][MT];
#pragma unroll
    for (int t = 0; t < PT; ++t)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][i][e] = 0.f;
    const int frag_row = lane & 31, frag_half = lane >> 5;

    // ---- prologue: the postnorm table by LDS-DMA FIRST (the oldest requests: every counted wait below covers them; waves 0-3 a
    //      KiB of the scale each, waves 4-7 of the shift) and the bias (a global load in the statistics phase costs a memory latency
    //      per use), then every stage; stage 0 landed ----
    {
        const int idx = (wave & 3) * 512 + lane * 8;
        const unsigned dst = __builtin_amdgcn_readfirstlane(smem_base + G::PRO_OFF + (wave >> 2) * 4096 + (wave & 3) * 1024);
        hd_dma16s<0>(wave < 4 ? a.pro_scale : a.pro_shift, (unsigned)((idx < K ? idx : 0) * 2), dst);
        if (a.JG == 0) hd_dma16s<0>(a.bias, (unsigned)((lane * 4 < a.C ? lane * 4 : 0) * 4), smem_base + G::BIAS_OFF);
    }
#pragma unroll
    for (int st = 0; st < STAGES; ++st) issue_step(st * STAGE_BYTES, st < nk ? st : nk - 1);
    if (na_hi) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((STAGES - 1) * NW_HI) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((STAGES - 1) * NW_LO) : "memory");
    const float* bias_lds = reinterpret_cast<const float*>(smem + G::BIAS_OFF);

    // fragment addresses inside a stage, per phase of the wave's K-part (ph: the 16-byte chunk (h NP + ph) 2 + lane half)
    unsigned a_addr[NP][MT], b_addr[NP][PT];
#pragma unroll
    for (int ph = 0; ph < NP; ++ph) {
        const int chunk = (h * NP + ph) * 2 + frag_half;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int row = i * 32 + frag_row;
            a_addr[ph][i] = row * ROW_BYTES + ((chunk ^ hd_swz(row)) << 4);
        }
#pragma unroll
        for (int t = 0; t < PT; ++t) {
            const int brow = (pg * PT + t) * 32 + frag_row;
            b_addr[ph][t] = WROWS * ROW_BYTES + brow * ROW_BYTES + ((chunk ^ hd_swz(brow)) << 4);
        }
    }
    // A wave's work is a chain of PHASES (K step k, ph): 5 PT MFMAs = five weight fragments against its PT pixel fragments.  The
    // weight fragments are read TWO MFMA groups ahead of their use (af[parity of the phase][tile]: the last two groups of a phase
    // read the first two fragments of the next), the pixel fragments and table rows of the next phase during this one
    // (pre-activated before the phase ends).
    half8_t af[2][MT], bq[2][PT], braw[PT], sc = {}, sh = {};
    auto rd_a = [&](auto par_c, auto i_c, int soff, auto ph_c) {
        af[decltype(par_c)::value][decltype(i_c)::value] =
            *reinterpret_cast<const half8_t*>(smem + soff + a_addr[decltype(ph_c)::value][decltype(i_c)::value]);
    };
    auto rd_b = [&](int soff, int k0, auto ph_c) {       // the raw pixel fragments and the table rows of their 8 channels
        constexpr int PH = decltype(ph_c)::value;
#pragma unroll
        for (int t = 0; t < PT; ++t) braw[t] = *reinterpret_cast<const half8_t*>(smem + soff + b_addr[PH][t]);
        const int chunk = (h * NP + PH) * 2 + frag_half;
        sc = *reinterpret_cast<const half8_t*>(pro_lds + k0 + chunk * 8);
        sh = *reinterpret_cast<const half8_t*>(pro_lds + 2048 + k0 + chunk * 8);
    };
    auto act = [&](auto par_c, int t) {                  // postnorm BN + ReLU, fp16 FMA, one rounding (resnet_v2.py:229)
        const half8_t z = {};
        bq[decltype(par_c)::value][t] = __builtin_elementwise_max(braw[t] * sc + sh, z);
    };
    auto mma = [&](auto par_c, auto i_c) {
        constexpr int P = decltype(par_c)::value, I = decltype(i_c)::value;
#pragma unroll
        for (int t = 0; t < PT; ++t) acc[t][I] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[P][I], bq[P][t], acc[t][I], 0, 0, 0);
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    using I4 = std::integral_constant<int, 4>;

    // One phase: fragment set PAR, phase PH of K step k (stage at byte offset cur; the next stage at nxt).  The LAST phase of a
    // step holds the step's barrier: every read of the stage is complete in every wave and step k + 1 has landed (this wave's
    // share; the barrier makes it everybody's), steps k + 2 ... stay in flight and step k + STAGES goes into this stage (past the
    // end: the last step again -- valid memory, a stage nobody reads).
    auto phase = [&](auto par_c, auto ph_c, int cur, int nxt, int k) {
        constexpr int P = decltype(par_c)::value, PH = decltype(ph_c)::value;
        constexpr bool LAST = PH == NP - 1;
        using PQ = std::integral_constant<int, P ^ 1>;
        using NPH = std::integral_constant<int, LAST ? 0 : PH + 1>;
        const int noff = LAST ? nxt : cur;
        const int k1 = k + 1 < nk ? k + 1 : nk - 1;
        const int nk0 = (LAST ? k1 : k) * BK;
        rd_a(par_c, I2{}, cur, ph_c);
        if constexpr (!LAST) rd_b(noff, nk0, NPH{});
        mma(par_c, I0{});
        hd_pin<LAST ? 1 : 3 + PT, PT, 2>();
        rd_a(par_c, I3{}, cur, ph_c); mma(par_c, I1{}); hd_pin<1, PT, 2>();
        rd_a(par_c, I4{}, cur, ph_c);
        if constexpr (!LAST) act(PQ{}, 0);
        mma(par_c, I2{});
        hd_pin<1, PT, 4>();
        if constexpr (LAST) {
            if (na_hi) hd_wait_barrier<(STAGES - 2) * NW_HI>(); else hd_wait_barrier<(STAGES - 2) * NW_LO>();
            issue_step(cur, k + STAGES < nk ? k + STAGES : nk - 1);
        }
        rd_a(PQ{}, I0{}, noff, NPH{});
        if constexpr (LAST) rd_b(noff, nk0, NPH{});
        if constexpr (!LAST && PT == 2) act(PQ{}, 1);
        mma(par_c, I3{});
        hd_pin<LAST ? 3 + PT : 1, PT, 4>();
        rd_a(PQ{}, I1{}, noff, NPH{}); mma(par_c, I4{}); hd_pin<1, PT, 2>();
        if constexpr (LAST) {
#pragma unroll
            for (int t = 0; t < PT; ++t) act(PQ{}, t);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    rd_b(0, 0, I0{});
    rd_a(I0{}, I0{}, 0, I0{});
    rd_a(I0{}, I1{}, 0, I0{});
#pragma unroll
    for (int t = 0; t < PT; ++t) act(I0{}, t);
    __builtin_amdgcn_sched_barrier(0);

    int cur = 0;                                   // byte offset of the stage of K step k
    auto next_of = [&](int off) { return off + STAGE_BYTES == RING_BYTES ? 0 : off + STAGE_BYTES; };
    if constexpr (NP == 2) {
        for (int k = 0; k < nk; ++k) {
            const int nxt = next_of(cur);
            phase(I0{}, I0{}, cur, nxt, k);
            phase(I1{}, I1{}, cur, nxt, k);
            cur = nxt;
        }
    } else {
        for (int k = 0; k < nk; k += 2) {          // one phase per step: the fragment sets alternate by step (nk is even)
            const int nxt = next_of(cur), nxt2 = next_of(nxt);
            phase(I0{}, I0{}, cur, nxt, k);
            phase(I1{}, I0{}, nxt, nxt2, k + 1);
            cur = nxt2;
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the re-requested tail has landed everywhere: the
                                                                                 // logits tiles overlay the ring

    // ---- K-parts -> fp32 logits (+ bias) of the group's 32 pixels, then the per-joint statistics, 32 pixels at a time ----
    float* lt0 = reinterpret_cast<float*>(smem) + pg * (KSPLIT - 1) * (32 * LROW);     // the group's tiles: part h writes tile h - 1
    const float step_s = 1.0f / (float)(a.side - 1);
    const float step_d = 1.0f / (float)(a.D - 1);
#pragma unroll
    for (int t = 0; t < PT; ++t) {
        if (h > 0) {
            float* lt = lt0 + (h - 1) * (32 * LROW);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c0 = i * 32 + 8 * q + 4 * frag_half;
#pragma unroll
                    for (int e = 0; e < 4; ++e) lt[frag_row * LROW + c0 + e] = acc[t][i][4 * q + e];
                }
        }
        __syncthreads();
        if (h == 0) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c0 = i * 32 + 8 * q + 4 * frag_half;
                    const floatx4 bv = *reinterpret_cast<const floatx4*>(bias_lds + c0);     // words past the head's channels: unused
#pragma unroll
                    for (int e = 0; e < 4; ++e) {      // ((part 0 + part 1) + ...) + bias: a fixed order
                        float v = acc[t][i][4 * q + e];
#pragma unroll
                        for (int r = 0; r < KSPLIT - 1; ++r) v += lt0[r * (32 * LROW) + frag_row * LROW + c0 + e];
                        lt0[frag_row * LROW + c0 + e] = v + bv[e];
                    }
                }
        }
        __syncthreads();
        const int p0 = (pg * PT + t) * 32;                        // first pixel of the tile inside the block
        if (a.logits_out != nullptr) {
            for (int idx = h * 64 + lane; idx < 32 * cg; idx += KSPLIT * 64) {
                const int p = idx / cg, c = idx - p * cg;
                a.logits_out[(size_t)(m0 + p0 + p) * a.C + src_row(c)] = lt0[p * LROW + c];
            }
        }
        const int pim = tile * TN + p0 + frag_row;                // pixel index inside the image
        const int py = pim / a.side, px = pim - py * a.side;
        const float cx = (float)px * step_s, cy = (float)py * step_s;
        const int slab = tile * (TN / 32) + pg * PT + t;
        float* rec = a.partials + (((size_t)img * a.slabs + slab) * a.J + j0) * 5;
        hd_tile_stats(lt0 + frag_row * LROW, jg, a.D, h * 2 + frag_half, 2 * KSPLIT, cx, cy, step_d, frag_row == 0,
                      [&](int j, float m, float s_, float sx, float sy, float sz) {
                          float* o5 = rec + j * 5;
                          o5[0] = m; o5[1] = s_; o5[2] = sx; o5[3] = sy; o5[4] = sz;
                      });
        if (t + 1 < PT) __syncthreads();                          // the tiles are rewritten in the next round
    }
}

// heads this launch is built for: fp16 input, whole 64-pixel slabs per image, at most 160 channels, K in whole steps
bool head_f16_supported(int c_in, int c_head, int n_joints, int depth, int side) {
    static const int enabled = tuning_knob("METRO_HEAD_FUSED", 1);
    const int pixels = side * side;
    if (!(enabled && c_head == n_joints * depth && c_in % hd::BK == 0 && c_in <= 2048 && c_in / hd::BK >= hd::STAGES - 1 &&
          pixels % hd::TN == 0 && side >= 2 && depth >= 2))
        return false;
    if (c_head <= hd::TM) return true;
    // wider heads: joint groups of at most 160 channels on the ring kernel (its K loop: an even number of 64-channel steps, >= 6)
    static const int grouped = tuning_knob("METRO_HEAD_GROUPS", 1);
    return grouped && depth <= hd::TM && c_in % 128 == 0 && c_in / 64 >= 6 && tuning_knob("METRO_HEAD_RING", 1);
}
// joints per group of a head wider than 160 channels (0 = the head is one group)
static int head_f16_group_joints(int c_head, int depth) { return c_head <= hd::TM ? 0 : hd::TM / depth; }
// 256-pixel tiles once they still give every CU a tile
static bool head_f16_big(int n, int side) {
    static const int t256 = tuning_knob("METRO_HEAD_256", 1);
    return t256 && (side * side) % hd2::TN == 0 && (long)n * (side * side / hd2::TN) >= 256;
}
// Which kernel runs a head at batch n: 0 = 64-pixel tiles (round 2), 1 = 256-pixel tiles over the whole K (round 3), 2 / 3 / 4 = the
// ring kernel with 256- / 128- / 64-pixel tiles (round 4).  The ring kernel's phases alternate two fragment sets by step when a wave
// has one phase per step: an even number of 64-channel K steps, at least as many as it has stages.
static int head_f16_variant(int n, int c_in, int c_head, int side) {
    static const int ring = tuning_knob("METRO_HEAD_RING", 1);
    const int pixels = side * side;
    const bool ring_ok = ring && c_in % 128 == 0 && c_in / 64 >= 6;
    if (c_head > hd::TM) return pixels % 128 == 0 && (long)n * (pixels / 128) >= 256 ? 3 : 4;      // joint groups: ring kernel only
    if (head_f16_big(n, side)) return ring_ok && c_head <= 144 ? 2 : 1;
    if (!ring_ok) return 0;
    if (pixels % 128 == 0 && (long)n * (pixels / 128) >= 256) return 3;
    return 4;
}
// records per image: one per 32 pixels (every kernel but the 64-pixel one of round 2); the partials slot is sized for the larger
int head_f16_slabs(int side) { return side * side / 32; }
// ... and what a launch at batch n writes (= what softargmax_finalize must fold)
int head_f16_records(int n, int c_in, int c_head, int side) {
    return head_f16_variant(n, c_in, c_head, side) == 0 ? side * side / hd::TN : side * side / 32;
}

template <int TN, int KSPLIT, int WROWS>
static int launch_head_ring(const HeadArgs& a, int n, int side, hipStream_t stream) {
    using G = HeadRing<TN, KSPLIT, WROWS>;
    auto kern = head_f16_ring_kernel<TN, KSPLIT, WROWS>;
    static PerDeviceInt done;
    if (const int st = ensure_dyn_lds(reinterpret_cast<const void*>(kern), G::LDS_BYTES, done, "head_f16<ring>")) return st;
    const int groups = a.JG > 0 ? (a.J + a.JG - 1) / a.JG : 1;
    hipLaunchKernelGGL(kern, dim3(side * side / TN, n, groups), dim3(G::NT), G::LDS_BYTES, stream, a);
    return launch_status("head_f16<ring>");
}

int launch_head_f16(const void* x, const void* w, const float* bias, const void* pro_scale, const void* pro_shift,
                    int n, int c_in, int c_head, int n_joints, int depth, int side, float* partials, float* logits_out,
                    hipStream_t stream) {
    if (!head_f16_supported(c_in, c_head, n_joints, depth, side)) {
        set_error("head_f16: unsupported head (c_in %d, %d channels = %d joints x depth %d, side %d)", c_in, c_head, n_joints, depth, side);
        return METRO_ERR_UNSUPPORTED;
    }
    const int variant = head_f16_variant(n, c_in, c_head, side);
    const int jgrp = head_f16_group_joints(c_head, depth);
    const int wrows = jgrp > 0 ? 160 : c_head <= 144 ? 144 : 160;
    char grp[16] = "";
    if (jgrp > 0) snprintf(grp, sizeof(grp), ",g%d", (n_joints + jgrp - 1) / jgrp);
    if (variant >= 2 ? note_kernel("head_f16<%dx%d,k%d%s>", wrows, variant == 2 ? 256 : variant == 3 ? 128 : 64, variant == 2 ? 2 : 4, grp)
                     : note_kernel(variant == 1 ? "head_f16<160x256>" : "head_f16<160x64>"))
        return METRO_OK;
    HeadArgs a;
    a.x = static_cast<const half_t*>(x); a.w = static_cast<const half_t*>(w); a.bias = bias;
    a.pro_scale = static_cast<const half_t*>(pro_scale); a.pro_shift = static_cast<const half_t*>(pro_shift);
    a.partials = partials; a.logits_out = logits_out;
    a.K = c_in; a.C = c_head; a.J = n_joints; a.D = depth; a.side = side; a.pixels = side * side;
    a.slabs = head_f16_records(n, c_in, c_head, side);
    a.JG = jgrp;
    if (variant == 2) return launch_head_ring<256, 2, 144>(a, n, side, stream);
    if (variant == 3) return wrows == 144 ? launch_head_ring<128, 4, 144>(a, n, side, stream) : launch_head_ring<128, 4, 160>(a, n, side, stream);
    if (variant == 4) return wrows == 144 ? launch_head_ring<64, 4, 144>(a, n, side, stream) : launch_head_ring<64, 4, 160>(a, n, side, stream);
    if (variant == 1) {
        static PerDeviceInt done2;
        if (const int st = ensure_dyn_lds(reinterpret_cast<const void*>(head_f16_kernel256), hd2::LDS_BYTES, done2, "head_f16<256>")) return st;
        hipLaunchKernelGGL(head_f16_kernel256, dim3(side * side / hd2::TN, n), dim3(hd2::NT), hd2::LDS_BYTES, stream, a);
        return launch_status("head_f16<256>");
    }
    static PerDeviceInt done;
    if (const int st = ensure_dyn_lds(reinterpret_cast<const void*>(head_f16_kernel), hd::LDS_BYTES, done, "head_f16")) return st;
    hipLaunchKernelGGL(head_f16_kernel, dim3(a.slabs, n), dim3(hd::NT), hd::LDS_BYTES, stream, a);
    return launch_status("head_f16");
}

}  // namespace metro
