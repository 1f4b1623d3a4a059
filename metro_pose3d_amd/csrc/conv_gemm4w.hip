// 1x1 convolution as a 256 x 256 x 64 GEMM with FOUR waves of 128 x 128 (gfx950) -- the deep-K 1x1 layers at large batch.
//
// experimental/conv_gemm8p.hip (round 2, no longer dispatched) gives each of its eight waves a 128 x 64 tile: 24 ds_read_b128 per 32 MFMAs, and at batch 256 hipBLASLt's
// 256^2 kernel (four waves, 128 x 128 wave tiles: 32 reads per 64 MFMAs = two thirds of the LDS fragment bytes per MFMA) was
// 22-28 % ahead on every deep-K shape (NOTES_dead_ends.md, "the ROCm libraries on the same shapes").  This is that geometry:
//   * one wave per SIMD, 256 fp32 accumulator registers (4 x 4 tiles of 32 x 32) + two fragment sets + one staged K tile:
//     the whole 512-register file of the SIMD belongs to the wave;
//   * operands are REGISTER staged: global_load_dwordx4 -> VGPR -> ds_write_b128 (XOR-swizzled rows as everywhere in this
//     repo).  An LDS-DMA costs its wave ~60 issue cycles (MI355X_MICROARCH.md); with two waves per SIMD the partner's MFMAs
//     cover that, with ONE wave per SIMD nothing does -- plain loads and ds_writes issue in the gaps between MFMAs;
//   * the K tile t+2 is requested while tile t is computed; tile t+1 (in registers since the previous tile) is written to the
//     other LDS buffer piece by piece during the first three k steps, the pre-activation (reference resnet_v2.py:119:
//     fp16 BN + ReLU on the consumer's side) applied ONCE per element on its way into LDS instead of once per fragment read;
//   * ONE barrier per K tile, placed in front of the last k step: its 16 MFMAs run behind the barrier and cover the first
//     fragment reads of the next tile;
//   * epilogue, fused-pair routing and arithmetic are conv_gemm8p's (fp32 accumulate, fp16(conv + bias), fp16 shortcut add:
//     reference resnet_v2.py:119-138 under tfu.py:426-440).  Same K order as every other fp16 conv kernel here (k ascending,
//     one fp32 accumulator per output): bit-identical results.
#include <type_traits>

#include "metro_common.h"

namespace metro {

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

namespace g4 {
constexpr int TM = 256, TN = 256, BK = 64, NT = 256;
constexpr int ROW_BYTES = BK * 2;                  // 128
constexpr int OPER_BYTES = 256 * ROW_BYTES;        // one operand image of a K tile: 32 KiB
constexpr int A_OFF = 0, B_OFF = 2 * OPER_BYTES;   // [A buf0 | A buf1 | B buf0 | B buf1]
constexpr int RING_BYTES = 4 * OPER_BYTES;         // 128 KiB
constexpr int OUT_ROW_BYTES = TM * 2 + 16;
constexpr int OUT_BYTES = TN * OUT_ROW_BYTES;      // 135168
constexpr int MAIN_BYTES = OUT_BYTES > RING_BYTES ? OUT_BYTES : RING_BYTES;
constexpr int PRO_BYTES = 2 * 2048 * 2;            // scale | shift, c_in <= 2048
constexpr int PIECES = 8;                          // 16-byte loads per lane per operand per K tile (256 rows x 128 B / 256 lanes / 16 B)
}  // namespace g4

__device__ __forceinline__ int g4_swz(int row) { return (row >> 1) & 7; }
typedef float g4_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int g4_u32x2 __attribute__((ext_vector_type(2)));

// NJ: 32-pixel fragments per wave.  4 = the 256 x 256 tile (wave 128 x 128); 2 (round 6) = a 256 cout x 128 pixel HALF tile (wave
// 128 x 64) for the deep-K layers of batch 64 that have only 128 whole tiles (block4 conv1: 2048 -> 512 on 16 384 pixels; the
// conv1 part of block4's shortcut + conv1 pair, whose 640 whole tiles were 2.5 rounds on 256 CUs run as three).  Same K order, same bits.
// tile_m0: first cout tile of this launch (a pair may be launched in two parts).
// `bid` of `nblk`: this block's index among the blocks of its part of the launch.
template <bool PROLOGUE, int NJ>
__device__ __forceinline__ void conv_gemm4w_body(
    const ConvArgs& a, const half_t* __restrict__ in, const half_t* __restrict__ w, const float* __restrict__ bias,
    const half_t* __restrict__ pro_scale, const half_t* __restrict__ pro_shift, const half_t* __restrict__ residual,
    half_t* __restrict__ out, half_t* __restrict__ out2, int tiles_m, int mgroups, int tile_m0, int bid, int nblk) {
    using namespace g4;
    constexpr int TNV = NJ * 64;                       // pixels per tile: 256, 128 or (NJ 1: a QUARTER tile, wave 128 x 32) 64
    constexpr int PB = PIECES * NJ / 4;                // pixel pieces per lane per K tile: 8, 4 or 2
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1;            // 128-cout half
    const int wc = wave & 1;             // pixel half (NJ * 32 pixels)

    // XCD-aware (bijective) block -> tile map: the blocks of one XCD share pixel tiles in their L2
    int lid;
    {
        const int b = bid;
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = b & 7, idx = b >> 3;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int tile_n = lid / tiles_m;
    int tile_m = lid % tiles_m;
    if (mgroups > 1) {
        // wide outputs (block4's shortcut + conv1 pair: ten cout tiles = 5.2 MB of weights, more than an XCD's 4 MB L2): XCD x
        // owns cout-tile group x % mgroups and pixel-tile set x / mgroups, so its weight working set is 1 / mgroups of the
        // layer and stays L2-resident while each pixel tile is fetched by mgroups XCDs instead of one (host checks divisibility)
        const int xcd = bid & 7, idx = bid >> 3;
        const int tmg = tiles_m / mgroups, sets = 8 / mgroups;
        const int tns = (nblk / tiles_m) / sets;
        tile_m = (xcd % mgroups) * tmg + idx % tmg;
        tile_n = (xcd / mgroups) * tns + idx / tmg;
    }
    const int m0 = tile_n * TNV;
    const int n0 = (tile_m0 + tile_m) * TM;
    const int K = a.c_in;
    const int nk = K / BK;
    half_t* pro_lds = reinterpret_cast<half_t*>(smem + MAIN_BYTES);

    // ---- staging coordinates: piece e of an operand = rows 32 e + 8 wave + (lane >> 3), 16-byte chunk lane & 7 ------
    const int lrow = lane >> 3, lch = lane & 7;
    const int srow = wave * 8 + lrow;                                  // row of piece 0; piece e adds 32 e (same swizzle)
    const unsigned st_off = (unsigned)(srow * ROW_BYTES + ((lch ^ g4_swz(srow)) << 4));   // LDS byte offset inside an operand image
    // buffer loads: wave-uniform descriptor of the tile's operand rows + ONE per-lane 32-bit offset for both operands (the piece
    // and the K tile go into the scalar offset): no 64-bit address arithmetic in the loop, no address registers
    const auto wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(w + (size_t)n0 * K), 0, 256 * K * 2, 0x00020000);
    const auto xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(in + (size_t)m0 * K), 0, TNV * K * 2, 0x00020000);
    const int voff = (srow * K + lch * 8) * 2;
    const int piece_stride = 32 * K * 2;                               // bytes

    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#ifndef METRO_G4_DEPTH
#define METRO_G4_DEPTH 1     // 2 (a second staged tile, three tiles of lead) measured the same: the loads are not late
#endif
    constexpr int DEPTH = METRO_G4_DEPTH;                              // staged K tiles in registers: tile k lives in set k % DEPTH
    u32x4 ra[DEPTH][PIECES], rb[DEPTH][PB];
    // the pixel piece that travels with weight piece e (-1: none): NJ 4: e; NJ 2: every second weight piece carries one
    auto bpiece = [](int e) { return NJ == 4 ? e : (e % (4 / NJ) == 0 ? e / (4 / NJ) : -1); };
    auto load_tile = [&](int set, int kt) {
#pragma unroll
        for (int e = 0; e < PIECES; ++e) {
            ra[set][e] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, voff, e * piece_stride + kt * BK * 2, 0);
            if (bpiece(e) >= 0) rb[set][bpiece(e)] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, voff, bpiece(e) * piece_stride + kt * BK * 2, 0);
        }
    };
    // pre-activation BN + ReLU on a staged pixel chunk (fp16 FMA, one rounding: resnet_v2.py:119); a lane's chunk is always
    // channels kt*64 + 8 lch .. +7: scale / shift are read once per K tile
    half8_t pro_sc = {}, pro_sh = {};
    auto load_pro = [&](int kt) {
        if constexpr (PROLOGUE) {
            pro_sc = *reinterpret_cast<const half8_t*>(pro_lds + kt * BK + lch * 8);
            pro_sh = *reinterpret_cast<const half8_t*>(pro_lds + 2048 + kt * BK + lch * 8);
        }
    };
    auto store_piece = [&](int buf, int set, int e) {
        *reinterpret_cast<u32x4*>(smem + A_OFF + buf * OPER_BYTES + st_off + e * 32 * ROW_BYTES) = ra[set][e];
        const int eb = bpiece(e);
        if (eb < 0) return;
        if constexpr (PROLOGUE) {
            const half8_t z = {};
            half8_t x = *reinterpret_cast<const half8_t*>(&rb[set][eb]);
            x = __builtin_elementwise_max(x * pro_sc + pro_sh, z);
            *reinterpret_cast<half8_t*>(smem + B_OFF + buf * OPER_BYTES + st_off + eb * 32 * ROW_BYTES) = x;
        } else {
            *reinterpret_cast<u32x4*>(smem + B_OFF + buf * OPER_BYTES + st_off + eb * 32 * ROW_BYTES) = rb[set][eb];
        }
    };

    floatx16 acc[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int frag_row = lane & 31, frag_half = lane >> 5;
    unsigned a_base[4], b_base[NJ];      // fragment addresses (buffer 0); k step kk = XOR with kk << 5 on the swizzled chunk bits
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = wr * 128 + i * 32 + frag_row;
        a_base[i] = A_OFF + row * ROW_BYTES + ((frag_half ^ g4_swz(row)) << 4);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int row = wc * (NJ * 32) + j * 32 + frag_row;
        b_base[j] = B_OFF + row * ROW_BYTES + ((frag_half ^ g4_swz(row)) << 4);
    }

    // ---- prologue: table, tile 0 -> LDS buffer 0, tile 1 -> registers -----------------------------------------------
    load_tile(0, 0);
    if (PROLOGUE) {
        for (int c = tid * 8; c < K; c += NT * 8) {
            *reinterpret_cast<uint4*>(pro_lds + c) = *reinterpret_cast<const uint4*>(pro_scale + c);
            *reinterpret_cast<uint4*>(pro_lds + 2048 + c) = *reinterpret_cast<const uint4*>(pro_shift + c);
        }
        __syncthreads();
    }
    load_pro(0);
#pragma unroll
    for (int e = 0; e < PIECES; ++e) store_piece(0, 0, e);
    load_tile(1 % DEPTH, nk > 1 ? 1 : 0);                 // tile 1 (and, two deep, tile 2) wait in registers
    if (DEPTH == 2) load_tile(0, nk > 2 ? 2 : nk - 1);
    __syncthreads();

    half8_t af[2][4], bf[2][NJ];         // two fragment sets: the k step being multiplied and the next one
    auto load_frags = [&](int set, int buf, int kk) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) bf[set][j] = *reinterpret_cast<const half8_t*>(smem + buf * OPER_BYTES + (b_base[j] ^ (kk << 5)));
#pragma unroll
        for (int i = 0; i < 4; ++i) af[set][i] = *reinterpret_cast<const half8_t*>(smem + buf * OPER_BYTES + (a_base[i] ^ (kk << 5)));
    };
    auto mma = [&](int set) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[set][i], bf[set][j], acc[i][j], 0, 0, 0);
    };
    load_frags(0, 0, 0);

    // One K tile: k steps 0-2 multiply, write the staged tile t+1 into the OTHER buffer (3 + 3 + 2 pieces) and re-request the
    // registers for tile t+2; then every wave's reads of this buffer and writes of the other one are complete -> barrier ->
    // first fragments of tile t+1 -> k step 3 (16 MFMAs behind the barrier).
    // (Past the end the LAST tile is requested and staged again -- valid memory, a buffer nobody reads any more -- so that the
    // loop body is one straight-line block: conditional requests made hipcc keep the staged tile in scratch memory.)
    auto ktile = [&](auto buf_c, int kt) {
        constexpr int BUF = decltype(buf_c)::value;
        const int kt1 = kt + 1 < nk ? kt + 1 : nk - 1;
        const int kt2 = kt + 1 + DEPTH < nk ? kt + 1 + DEPTH : nk - 1;       // the tile requested into the set that is written now
        const int knext = kt2 * BK * 2;
        constexpr int SET = DEPTH == 2 ? (BUF ^ 1) : 0;                       // tile kt + 1 (kt even <=> BUF 0: the K loop is unrolled by two)
        load_pro(kt1);
        auto step = [&](auto kk_c) {
            constexpr int kk = decltype(kk_c)::value;
            constexpr int first[4] = {0, 3, 6, 8};
            constexpr int NP = first[kk + 1] - first[kk];
            load_frags((kk + 1) & 1, BUF, kk + 1);
#pragma unroll
            for (int e = first[kk]; e < first[kk + 1]; ++e) store_piece(BUF ^ 1, SET, e);
#pragma unroll
            for (int e = first[kk]; e < first[kk + 1]; ++e) {
                ra[SET][e] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, voff, e * piece_stride + knext, 0);
                if (bpiece(e) >= 0) rb[SET][bpiece(e)] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, voff, bpiece(e) * piece_stride + knext, 0);
            }
            mma(kk & 1);
            // emitted order of the step: a fragment read (and the pre-activation VALU of the pieces about to be written) behind each
            // of the first 8 MFMAs, then one LDS write + one request behind each of the next ones
            if (PROLOGUE && kk == 0) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // this tile's scale / shift
            if constexpr (NJ == 4) {
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if (PROLOGUE) __builtin_amdgcn_sched_group_barrier(0x002, NP, 0);
                }
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (m < 2 * NP) {
                        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                }
            } else {
                // half tile: 8 MFMAs per k step carry 6 fragment reads, the pre-activation of <= 2 pixel pieces (8 packed ops each)
                // and NP weight + NPB pixel pieces written and re-requested: a read, then a write + a request behind every MFMA
                // (quarter tile: 4 MFMAs per k step, 5 reads, <= 4 writes + requests: two of each behind every MFMA)
                constexpr int STR = 4 / NJ;                                                  // every STR-th weight piece carries a pixel piece
                constexpr int NPB = (first[kk + 1] + STR - 1) / STR - (first[kk] + STR - 1) / STR;
                constexpr int NWR = NP + NPB;
                constexpr int NM = 4 * NJ, NRD = 4 + NJ;
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (NJ == 1 && m == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // quarter tile: 5 reads behind 4 MFMAs
                    if (m < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if (PROLOGUE && NPB > 0) __builtin_amdgcn_sched_group_barrier(0x002, 8 * NPB / NM, 0);
                    if (NJ == 2 ? m >= NM - NWR : true) {
                        constexpr int PER = NJ == 2 ? 1 : (NWR + NM - 1) / NM;
                        __builtin_amdgcn_sched_group_barrier(0x200, PER, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, PER, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{});
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        load_frags(0, BUF ^ 1, 0);
        mma(1);                                  // k step 3: fragments read before the barrier
#pragma unroll
        for (int m = 0; m < (NJ == 4 ? 8 : NJ == 2 ? 6 : 4); ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (NJ == 1 && m == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        if (NJ != 1) __builtin_amdgcn_sched_group_barrier(0x008, NJ == 4 ? 8 : 2, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int t = 0; t < nk; t += 2) {
        ktile(std::integral_constant<int, 0>{}, t);
        ktile(std::integral_constant<int, 1>{}, t + 1);
    }
    __syncthreads();                             // every wave is done with the ring: the epilogue tile overlays it

    // ---- epilogue: accumulators (+bias, ReLU) -> LDS [pixel][cout] fp16 -> full-line stores (+ shortcut) ----
    const bool second = a.split > 0 && n0 >= a.split;     // fused pair: this cout tile belongs to one of the two outputs
    const int o_c = a.split > 0 ? (second ? a.c_out2 : a.split) : a.c_out;
    const int o_n0 = second ? n0 - a.split : n0;
    const int o_relu = second ? a.relu2 : a.relu;
    half_t* o_ptr = second ? out2 : out;
    // 256 accumulators per lane leave through ~1 150 VALU operations when written element by element (accvgpr_read, add, max + select
    // for the run-time ReLU flag, convert): 7 000 - 7 800 cycles of the tile's 12 000-cycle epilogue with one wave per SIMD
    // (tools/gemm4w_clock.py).  Here: the 16 bias quads fetched up front, packed fp32 adds, v_cvt_pk_f16_f32 (round to nearest even, as
    // the (half_t) casts), and the ReLU on the packed fp16 result -- max(fp16(v), 0) == fp16(max(v, 0)): rounding is monotonic and keeps
    // the sign -- under a block-uniform branch.  Same bits.
    floatx4 bvq[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) bvq[i][q] = *reinterpret_cast<const floatx4*>(bias + n0 + wr * 128 + i * 32 + 8 * q + 4 * frag_half);
    auto write_tile = [&](auto relu_c) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = wr * 128 + i * 32 + 8 * q + 4 * frag_half;
                const g4_f32x2 blo = {bvq[i][q][0], bvq[i][q][1]}, bhi = {bvq[i][q][2], bvq[i][q][3]};
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int prow = wc * (NJ * 32) + j * 32 + frag_row;
                    g4_f32x2 lo = {acc[i][j][4 * q], acc[i][j][4 * q + 1]}, hi = {acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                    asm("v_pk_add_f32 %0, %1, %2" : "=v"(lo) : "v"(lo), "v"(blo));
                    asm("v_pk_add_f32 %0, %1, %2" : "=v"(hi) : "v"(hi), "v"(bhi));
                    g4_u32x2 r;
                    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r.x) : "v"(lo.x), "v"(lo.y));
                    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r.y) : "v"(hi.x), "v"(hi.y));
                    half4_t hv = __builtin_bit_cast(half4_t, r);
                    if constexpr (decltype(relu_c)::value) {
                        const half4_t z = {};
                        hv = __builtin_elementwise_max(hv, z);
                    }
                    *reinterpret_cast<half4_t*>(smem + prow * OUT_ROW_BYTES + col * 2) = hv;
                }
            }
        }
    };
    if (o_relu) write_tile(std::true_type{});
    else write_tile(std::false_type{});
    __syncthreads();
    constexpr int CPRO = TM / 8;                       // 16-byte chunks per tile row
    constexpr int EPI_ITERS = TNV * CPRO / NT;         // 32 (16 for the half tile)
    const bool res_same = a.res_stride == 1 && a.res_offset == 0 && a.res_h == a.h_out && a.res_w == a.w_out;
    const int hw_out = a.h_out * a.w_out;
    // no shortcut to add (every layer this kernel is dispatched for) and a whole 256-channel tile: the chunk reads of eight
    // iterations back to back, then eight full-line stores.  The general loop below carries a per-lane guard and the shortcut
    // branches in every iteration -- hipcc emits read - wait - store 32 times in a row (~2.5 us per 256 x 256 tile).
    if (residual == nullptr && o_n0 + TM <= o_c && m0 + TNV <= a.m_total) {     // block-uniform; m % 256 == 0 is also a host-side precondition
        const int ch = tid & (CPRO - 1), pr0 = tid / CPRO;         // NT % CPRO == 0: a lane keeps its chunk column
#pragma unroll
        for (int it0 = 0; it0 < EPI_ITERS; it0 += 8) {
            uint4 vv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) vv[u] = *reinterpret_cast<const uint4*>(smem + (pr0 + (it0 + u) * (NT / CPRO)) * OUT_ROW_BYTES + ch * 16);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                store_out16<2>(o_ptr + (size_t)(m0 + pr0 + (it0 + u) * (NT / CPRO)) * o_c + o_n0 + ch * 8, vv[u]);
        }
        return;
    }
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

template <bool PROLOGUE, int NJ>
__global__ __launch_bounds__(g4::NT) void conv_gemm4w_kernel(
    ConvArgs a, const half_t* __restrict__ in, const half_t* __restrict__ w, const float* __restrict__ bias,
    const half_t* __restrict__ pro_scale, const half_t* __restrict__ pro_shift, const half_t* __restrict__ residual,
    half_t* __restrict__ out, half_t* __restrict__ out2, int tiles_m, int mgroups, int tile_m0) {
    conv_gemm4w_body<PROLOGUE, NJ>(a, in, w, bias, pro_scale, pro_shift, residual, out, out2, tiles_m, mgroups, tile_m0, blockIdx.x, gridDim.x);
}

// ONE grid for both parts of a pair: blocks [0, n_whole) = whole tiles of cout tiles [0, tm_whole), the rest = half tiles of cout
// tiles [tm_whole, tm_whole + tm_half).  The dispatcher hands out blocks in order, so the half tiles fill the CUs the last round
// of whole tiles leaves idle -- no launch boundary between the parts.  (n_whole % 8 == 0: a block's XCD is blockIdx % 8 in both.)
template <bool PROLOGUE>
__global__ __launch_bounds__(g4::NT) void conv_gemm4w_mixed_kernel(
    ConvArgs a, const half_t* __restrict__ in, const half_t* __restrict__ w, const float* __restrict__ bias,
    const half_t* __restrict__ pro_scale, const half_t* __restrict__ pro_shift, half_t* __restrict__ out, half_t* __restrict__ out2,
    int tm_whole, int tm_half, int n_whole) {
    if ((int)blockIdx.x < n_whole)
        conv_gemm4w_body<PROLOGUE, 4>(a, in, w, bias, pro_scale, pro_shift, nullptr, out, out2, tm_whole, 1, 0, blockIdx.x, n_whole);
    else
        conv_gemm4w_body<PROLOGUE, 2>(a, in, w, bias, pro_scale, pro_shift, nullptr, out, out2, tm_half, 1, tm_whole, blockIdx.x - n_whole,
                                      gridDim.x - n_whole);
}

// What the kernel can run: 1x1, stride 1, no padding, dense NHWC fp16 in/out, an even number of 64-channel K tiles,
// WHOLE tiles (no zero page here; at stride 16 every image is exactly one 256-pixel tile); a fused pair splits on a
// tile boundary and its second output is a whole number of cout tiles (256 = conv1 of block3, 512 = conv1 of block4).
bool conv_gemm4w_shape_ok(const MetroConvDesc& d, const ConvSplit* split) {
    if (!(d.kh == 1 && d.kw == 1 && d.stride == 1 && d.pad_top == 0 && d.pad_left == 0 && d.in_pix_stride == d.c_in &&
          d.h_in == d.h_out && d.w_in == d.w_out && d.in_dtype == METRO_F16 && d.out_dtype == METRO_F16))
        return false;
    const long m = (long)d.n * d.h_out * d.w_out;
    if (d.c_in % 128 != 0 || d.c_in < 128 || d.c_in > 2048 || d.c_out % 256 != 0 || m % 256 != 0) return false;
    if (split != nullptr && split->split > 0 && (split->split % 256 != 0 || split->c_out2 % 256 != 0 || d.has_residual)) return false;
    return true;
}

// Half tiles (256 cout x 128 pixels, kernel comment) for `tiles_m` cout tiles of a layer with m pixels: when its whole tiles would
// leave more than a third of the CUs without work in their last round while the half tiles fill it.  METRO_G4_HALF: 0 = never.
// ... and QUARTER tiles (256 cout x 64 pixels) where even the half tiles leave half of the CUs idle.  Measured (same-box A/B,
// profiles/r06_ab_gemm4w_quarter_tiles.txt): block4's conv1 at batch 32 (2048 -> 512 on 8 192 pixels) -0.5 % of that step; block3's
// conv1 at batch 64 (1024 -> 256 on 16 384 pixels: 64 whole / 128 half / 256 quarter tiles) the same as the ring kernel's 128 x 128
// tiles (22.1 vs 22.2 us) -- hence K >= 2048.  METRO_G4_QUARTER: 0 = never.
static bool g4_quarter_tiles_pay(long tiles_m, long m, int c_in) {
    static const int enabled = tuning_knob("METRO_G4_QUARTER", 1);
    static const int min_k = tuning_knob("METRO_G4_QUARTER_MIN_K", 2048);
    const long half = tiles_m * (m / 128), quarter = tiles_m * (m / 64);
    return enabled && c_in >= min_k && m % 64 == 0 && half < 224 && quarter >= 224 && quarter <= 512;
}

static bool g4_half_tiles_pay(long tiles_m, long m, int c_in) {
    static const int enabled = tuning_knob("METRO_G4_HALF", 1);
    static const int min_k = tuning_knob("METRO_G4_HALF_MIN_K", 1024);
    const long whole = tiles_m * (m / 256), half = tiles_m * (m / 128);
    return enabled && c_in >= min_k && m % 128 == 0 && whole < 256 && half >= 224 && half <= 512;
}

// ... and when the dispatcher prefers it over the ring kernel: a pre-activated layer (every conv1 / projection shortcut / pair
// is one) with deep K -- the 256 x 256 loop needs tiles to amortise its 128 KiB prologue and its epilogue -- and at least one
// tile per CU.  Measured (MI355X): K = 512 with barely one tile per CU loses to the 128 x 256 ring kernel (block3's pair at
// batch 64: 44 vs 41 us), at four tiles per CU it wins (batch 256: 125 vs 134 us).
bool conv_gemm4w_supported(const MetroConvDesc& d, const ConvSplit* split) {
    static const int enabled = tuning_knob("METRO_GEMM4W", 1);
    // (the knobs were named METRO_GEMM8P_* after the round-2 kernel this one replaced; the old names stay as aliases)
    static const int min_tiles = tuning_knob("METRO_GEMM4W_MIN_TILES", tuning_knob("METRO_GEMM8P_MIN_TILES", 256));
    static const int min_k = tuning_knob("METRO_GEMM4W_MIN_K", tuning_knob("METRO_GEMM8P_MIN_K", 512));
    if (!enabled || !d.has_prologue || !conv_gemm4w_shape_ok(d, split) || d.c_in < min_k) return false;
    const long m = (long)d.n * d.h_out * d.w_out;
    const long tiles = (long)(d.c_out / 256) * (m / 256);
    if (tiles >= min_tiles && (d.c_in >= 1024 || tiles >= 4 * min_tiles)) return true;
    // round 6: one half / quarter tile per CU (block4 / block3 conv1 at batch 64)
    return !(split != nullptr && split->split > 0) && (g4_half_tiles_pay(d.c_out / 256, m, d.c_in) || g4_quarter_tiles_pay(d.c_out / 256, m, d.c_in));
}

template <bool PRO, int NJ>
static int launch_g4_part(const ConvArgs& a, const half_t* in, const half_t* w, const float* bias, const half_t* ps, const half_t* pb,
                          const half_t* r, half_t* out, half_t* out2, int tile_m0, int tiles_m, int mgroups, hipStream_t stream) {
    auto kern = conv_gemm4w_kernel<PRO, NJ>;
    constexpr int lds = g4::MAIN_BYTES + (PRO ? g4::PRO_BYTES : 0);
    static PerDeviceInt done;
    if (const int st = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds, done, "conv_gemm4w")) return st;
    const int tiles_n = a.m_total / (NJ * 64);
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(g4::NT), lds, stream, a, in, w, bias, ps, pb, r, out, out2, tiles_m, mgroups, tile_m0);
    return launch_status("conv_gemm4w");
}

int launch_conv_gemm4w(const MetroConvDesc& d, const void* in_, const void* w_, const float* bias, const void* ps_,
                       const void* pb_, const void* res, void* out_, hipStream_t stream, const ConvSplit* split) {
    if (!conv_gemm4w_shape_ok(d, split)) {
        set_error("conv_gemm4w: needs a 1x1 stride-1 fp16 layer with c_in %% 128 == 0 (<= 2048), c_out %% 256 == 0 and pixels %% 256 == 0 "
                  "(got c_in %d, c_out %d, %d x %d x %d pixels)", d.c_in, d.c_out, d.n, d.h_out, d.w_out);
        return METRO_ERR_UNSUPPORTED;
    }
    ConvArgs a = make_conv_args(d);
    void* out2_ = nullptr;
    if (split != nullptr && split->split > 0) {
        a.split = split->split; a.c_out2 = split->c_out2; a.relu2 = split->relu2;
        out2_ = split->out2;
    }
    const int tiles_m = d.c_out / g4::TM;
    const int tiles_n = a.m_total / g4::TN;
    // which part runs on which tile: everything on whole tiles; a lone layer with too few of them on half tiles; a pair whose
    // FIRST output fills whole rounds of 256 CUs and whose second output does not: the second output's cout tiles on half tiles
    // in a launch of their own (block4's pair at batch 64: 512 whole tiles + 256 half tiles instead of 640 whole tiles = 2.5 rounds)
    int tm_whole = tiles_m, tm_half = 0;
    if (a.split > 0) {
        const int tm_a = a.split / g4::TM, tm_b = a.c_out2 / g4::TM;
        if (((long)tm_a * tiles_n) % 256 == 0 && g4_half_tiles_pay(tm_b, a.m_total, d.c_in)) { tm_whole = tm_a; tm_half = tm_b; }
    } else if ((long)tiles_m * tiles_n < 256 && g4_half_tiles_pay(tiles_m, a.m_total, d.c_in)) {
        tm_whole = 0; tm_half = tiles_m;
    } else if ((long)tiles_m * tiles_n < 256 && g4_quarter_tiles_pay(tiles_m, a.m_total, d.c_in)) {
        if (note_kernel("conv_gemm4w<256x64%s>%s", d.has_prologue ? ",pro" : "", d.has_residual ? "+res" : "")) return METRO_OK;
        const half_t* rq = d.has_residual ? static_cast<const half_t*>(res) : nullptr;
        return d.has_prologue ? launch_g4_part<true, 1>(a, static_cast<const half_t*>(in_), static_cast<const half_t*>(w_), bias, static_cast<const half_t*>(ps_),
                                                        static_cast<const half_t*>(pb_), rq, static_cast<half_t*>(out_), nullptr, 0, tiles_m, 1, stream)
                              : launch_g4_part<false, 1>(a, static_cast<const half_t*>(in_), static_cast<const half_t*>(w_), bias, nullptr, nullptr, rq,
                                                         static_cast<half_t*>(out_), nullptr, 0, tiles_m, 1, stream);
    }
    const char* pro_s = d.has_prologue ? ",pro" : "";
    const char* res_s = d.has_residual ? "+res" : "";
    const char* pair_s = a.split > 0 ? "+pair" : "";
    bool dry = false;
    if (tm_whole > 0) dry = note_kernel("conv_gemm4w<256x256%s>%s%s", pro_s, res_s, pair_s);
    if (tm_half > 0) dry = note_kernel("conv_gemm4w<256x128%s>%s%s", pro_s, res_s, pair_s) || dry;
    if (dry) return METRO_OK;
    // cout-tile groups per XCD set (kernel comment): only where the layer's weights exceed an XCD's L2
    static const int mg_knob = tuning_knob("METRO_G4_MGROUPS", 1);
    static const int mg_min_w = tuning_knob("METRO_G4_MGROUPS_MIN_WBYTES", 4 << 20);
    int mgroups = 1;
    if (tm_half == 0 && mg_knob > 1 && (long)d.c_out * d.c_in * 2 > mg_min_w && 8 % mg_knob == 0 && tiles_m % mg_knob == 0 &&
        tiles_n % (8 / mg_knob) == 0)
        mgroups = mg_knob;
    const half_t* in = static_cast<const half_t*>(in_);
    const half_t* w = static_cast<const half_t*>(w_);
    const half_t* ps = static_cast<const half_t*>(ps_);
    const half_t* pb = static_cast<const half_t*>(pb_);
    const half_t* r = d.has_residual ? static_cast<const half_t*>(res) : nullptr;
    half_t* out = static_cast<half_t*>(out_);
    half_t* out2 = static_cast<half_t*>(out2_);
    static const int mixed = tuning_knob("METRO_G4_MIXED", 1);
    if (mixed && tm_whole > 0 && tm_half > 0 && r == nullptr && d.has_prologue && ((long)tm_whole * tiles_n) % 8 == 0) {
        auto kern = conv_gemm4w_mixed_kernel<true>;
        constexpr int lds = g4::MAIN_BYTES + g4::PRO_BYTES;
        static PerDeviceInt done;
        if (const int st = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds, done, "conv_gemm4w<mixed>")) return st;
        const int n_whole = tm_whole * tiles_n, n_half = tm_half * (a.m_total / 128);
        hipLaunchKernelGGL(kern, dim3(n_whole + n_half), dim3(g4::NT), lds, stream, a, in, w, bias, ps, pb, out, out2, tm_whole, tm_half, n_whole);
        return launch_status("conv_gemm4w<mixed>");
    }
    int st = METRO_OK;
    if (tm_whole > 0)
        st = d.has_prologue ? launch_g4_part<true, 4>(a, in, w, bias, ps, pb, r, out, out2, 0, tm_whole, mgroups, stream)
                            : launch_g4_part<false, 4>(a, in, w, bias, nullptr, nullptr, r, out, out2, 0, tm_whole, mgroups, stream);
    if (st == METRO_OK && tm_half > 0)
        st = d.has_prologue ? launch_g4_part<true, 2>(a, in, w, bias, ps, pb, r, out, out2, tm_whole, tm_half, 1, stream)
                            : launch_g4_part<false, 2>(a, in, w, bias, nullptr, nullptr, r, out, out2, tm_whole, tm_half, 1, stream);
    return st;
}

}  // namespace metro
