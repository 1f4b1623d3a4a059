// 1x1 convolution as a 256 x 256 x 64 GEMM with FOUR waves of 128 x 128 and BOTH operands by LDS-DMA (gfx950): the deep-K 1x1 layers.
//
// Where the two older forms of this GEMM stand (NOTES_dead_ends.md, "the ROCm libraries on the same shapes"): conv_gemm8p.hip (eight waves
// of 128 x 64, LDS-DMA) reads 24 fragments per 32 MFMAs and is bound by the LDS; conv_gemm4w.hip (four waves of 128 x 128: 32 reads
// per 64 MFMAs) stages through registers, and its ds_write_b128s (13 cycles each on the VGPR -> LDS path) + their waits sit in the
// one wave per SIMD that also issues the MFMAs.  The library's own best kernel on these shapes (hipBLASLt MT256x256x64, four waves,
// DirectToLds for both operands) has the geometry of the second and the staging of the first.  This kernel is that combination,
// with this repo's pre-activation on the consumer side:
//   * one wave per SIMD, 256 fp32 accumulators (4 x 4 tiles of 32 x 32) in the accumulator file;
//   * K tiles of 64 channels, TWO 64 KiB buffers (cout image 256 rows x 128 B | pixel image 256 rows x 128 B).  Nothing passes
//     through registers on its way into LDS: sixteen global_load_lds_dwordx4 per wave and K tile, each 8 rows x 128 B -- FULL
//     cache lines: tools/dma_depth_bench.hip measures 13.8 TB/s of L2 -> LDS delivery for 128-byte row pieces against 8.7 TB/s for
//     64-byte ones (the first form of this kernel, a ring of four 32-channel slots, was delivery bound at exactly that rate),
//     whatever the number of requests in flight, the instruction form or the number of waves;
//   * FOUR fragment sets (one per k step of a K tile): during the MFMAs of k step kk the pixel fragments of k step kk + 1 are
//     pre-activated and the fragments of k step kk + 2 are read.  One wave per SIMD issues in order, so a VALU op waiting for its
//     ds_read holds up every MFMA behind it: here neither a VALU op nor an MFMA ever waits for an LDS read of its own k step;
//   * ONE barrier per K tile, between k steps 1 and 2 (every read of this buffer retired, the next tile landed everywhere); the
//     requests of tile t + 2 go into the buffer that has just been retired during k steps 2, 3 and 0: pixel rows (first touch from
//     HBM) first, cout rows (weights: L2 resident) last.  Requests lead their use by 2-4 k steps;
//   * 128-byte LDS rows, 16-byte chunk index XOR (row >> 1) & 7 -- applied to the per-lane SOURCE address of the DMA (its LDS
//     image is lane-linear) and to the fragment reads, as in conv_gemm8p.hip;
//   * pre-activation BN + ReLU (reference resnet_v2.py:119) on the pixel fragments after their ds_read (4 v_pk_fma_f16 + 4
//     v_pk_max_f16 per fragment, between the MFMAs); epilogue, fused-pair routing and arithmetic are conv_gemm8p's (fp32
//     accumulate in ascending k, fp16(conv + bias), fp16 shortcut add: reference resnet_v2.py:119-138 under tfu.py:426-440): the
//     three kernels give the same bits.
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

namespace g4d {
constexpr int BK = 64, NT = 256;
constexpr int ROW_BYTES = BK * 2;                  // 128
constexpr int PRO_BYTES = 2 * 2048 * 2;            // scale | shift, c_in <= 2048
// MI x NJ 32 x 32 MFMA tiles per wave; 2 x 2 waves: block tile 64 MI couts x 64 NJ pixels
template <int MI, int NJ>
struct Geo {
    static constexpr int TM = 64 * MI, TN = 64 * NJ;
    static constexpr int OPER_A = TM * ROW_BYTES, OPER_B = TN * ROW_BYTES;      // one operand image of a K tile
    static constexpr int B_BASE = 2 * OPER_A;                                    // [A buf0 | A buf1 | B buf0 | B buf1]
    static constexpr int RING_BYTES = 2 * (OPER_A + OPER_B);
    static constexpr int OUT_ROW_BYTES = TM * 2 + 16;
    static constexpr int OUT_BYTES = TN * OUT_ROW_BYTES;
    static constexpr int MAIN_BYTES = OUT_BYTES > RING_BYTES ? OUT_BYTES : RING_BYTES;
    static constexpr int RA = 2 * MI, RB = 2 * NJ, R = RA + RB;                  // LDS-DMA requests per wave and K tile
    static constexpr int Q = (R + 2) / 3;                                        // ... issued per k step (k steps 2, 3 and 0)
    static constexpr int SLOTS = 2 * MI;                                         // request slots of a k step (two per MFMA segment)
    static_assert(Q <= SLOTS, "request schedule does not fit the k steps");
};
}  // namespace g4d

__device__ __forceinline__ int g4d_swz(int row) { return (row >> 1) & 7; }

typedef __attribute__((address_space(3))) void g4d_lds_void_t;

// one LDS-DMA wave-instruction: lane l's 16 bytes land at (lds_base + LDS_IMM) + 16 l.  Inline asm: hipcc treats the builtin as a
// may-alias LDS write and drains it with vmcnt(0) before every ds_read.  M0 is written in the statement that reads it.
template <int LDS_IMM>
__device__ __forceinline__ void g4d_dma16(const void* sbase, unsigned voff, unsigned lds_base) {
    asm volatile(
        "s_add_u32 m0, %2, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %1"
        :
        : "v"(voff), "s"(sbase), "s"(lds_base), "n"(LDS_IMM)
        : "scc");
}
template <int N>
__device__ __forceinline__ void g4d_wait_vm_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}
// f(integral_constant<0>) ... f(integral_constant<N - 1>)
template <int N, typename F>
__device__ __forceinline__ void g4d_for(F&& f) {
    if constexpr (N > 0) {
        g4d_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// INPLACE (round 5, experiment): the pre-activation is applied ONCE per element, in place in the landed pixel image, by the wave whose
// LDS-DMA brought the rows in (8 pieces of 1 KiB per wave and K tile: read - 8 packed ops - write, software-pipelined through the
// four MFMA segments of k step 1 behind a counted vmcnt that leaves the cout requests in flight) instead of on every fragment read
// (each pixel fragment is read by the two waves of a pixel half: twice the VALU work, and on the MFMA wave's critical path).
template <bool PROLOGUE, int MI, int NJ, bool INPLACE = false>
__global__ __launch_bounds__(g4d::NT, (MI * NJ <= 4 ? 2 : 1)) void conv_gemm4d_kernel(
    ConvArgs a, const half_t* __restrict__ in, const half_t* __restrict__ w, const float* __restrict__ bias,
    const half_t* __restrict__ pro_scale, const half_t* __restrict__ pro_shift, const half_t* __restrict__ residual,
    half_t* __restrict__ out, half_t* __restrict__ out2, int tiles_m) {
    using namespace g4d;
    using G = Geo<MI, NJ>;
    constexpr int TM = G::TM, TN = G::TN;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1;            // cout half
    const int wc = wave & 1;             // pixel half

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
    const int nk = K / BK;               // even (the launcher guarantees c_in % 128 == 0)
    const unsigned smem_base = (unsigned)(size_t)(g4d_lds_void_t*)smem;
    half_t* pro_lds = reinterpret_cast<half_t*>(smem + G::MAIN_BYTES);

    // ---- LDS-DMA sources.  One instruction = 8 rows x 128 B (lane l: row l >> 3, physical chunk l & 7 = logical chunk ^ swz(row)).
    // Wave w issues row groups 4 e + w (e = 0..), i.e. rows 32 e + 8 w + ...: 2 MI requests for the cout image, 2 NJ for the pixels
    const int lrow = lane >> 3, lch = lane & 7;
    constexpr int RMAX = G::RA > G::RB ? G::RA : G::RB;
    unsigned voff[RMAX];           // per-lane byte offset from the operand base of the tile (the same for both operands)
#pragma unroll
    for (int e = 0; e < RMAX; ++e) {
        const int row = (4 * e + wave) * 8 + lrow;
        voff[e] = (unsigned)(row * K + ((lch ^ g4d_swz(row)) * 8)) * 2u;
    }
    const unsigned lds_a = __builtin_amdgcn_readfirstlane(smem_base + wave * 8 * ROW_BYTES);            // group e adds e * 4 KiB
    const unsigned lds_b = lds_a + G::B_BASE;
#ifdef METRO_DBG_G4D_SAME_TILE                     // timing experiment: every block streams tile (0, 0): all requests hit the L2
    const half_t* wbase = w;
    const half_t* xbase = in;
#else
    const half_t* wbase = w + (size_t)n0 * K;      // wave-uniform operand bases of this tile
    const half_t* xbase = in + (size_t)m0 * K;
#endif
    // request g (0 .. R - 1: the pixel image first -- first touch from HBM --, then the cout image) of K tile kt into buffer BUF
    auto req = [&](auto buf_c, auto g_c, int kt) {
        constexpr int BUF = decltype(buf_c)::value, GI = decltype(g_c)::value;
#ifndef METRO_DBG_G4D_NO_DMA
        if constexpr (GI < G::RB) g4d_dma16<BUF * G::OPER_B + GI * 4096>(xbase + kt * BK, voff[GI], lds_b);
        else if constexpr (GI < G::R) g4d_dma16<BUF * G::OPER_A + (GI - G::RB) * 4096>(wbase + kt * BK, voff[GI - G::RB], lds_a);
#endif
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;

    floatx16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int frag_row = lane & 31, frag_half = lane >> 5;
    unsigned a_base[MI], b_base[NJ];     // fragment addresses (buffer 0, k step 0); k step kk: XOR kk << 5 (the swizzle is an XOR)
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int row = wr * (TM / 2) + i * 32 + frag_row;
        a_base[i] = row * ROW_BYTES + ((frag_half ^ g4d_swz(row)) << 4);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int row = wc * (TN / 2) + j * 32 + frag_row;
        b_base[j] = G::B_BASE + row * ROW_BYTES + ((frag_half ^ g4d_swz(row)) << 4);
    }

    // INPLACE: this wave's own pixel pieces of a K tile (piece g: rows 32 g + 8 wave + (lane >> 3), physical chunk lane & 7 -- the
    // logical chunk, hence the lane's 8 channels, is the same for all eight pieces: swz(row) does not see g)
    const int rmw_chunk = lch ^ ((wave * 4 + (lrow >> 1)) & 7);
    half8_t prs = {}, prh = {};
    half8_t rp[G::RB];
    auto rmw_pro = [&](int kt) {
        prs = *reinterpret_cast<const half8_t*>(pro_lds + kt * BK + rmw_chunk * 8);
        prh = *reinterpret_cast<const half8_t*>(pro_lds + 2048 + kt * BK + rmw_chunk * 8);
    };
    auto rmw_addr = [&](int buf, int g) { return smem + G::B_BASE + buf * G::OPER_B + g * 4096 + wave * 1024 + lane * 16; };
    auto rmw_read = [&](int buf, int g) { rp[g] = *reinterpret_cast<const half8_t*>(rmw_addr(buf, g)); };
    auto rmw_write = [&](int buf, int g) {
        const half8_t z = {};
        *reinterpret_cast<half8_t*>(rmw_addr(buf, g)) = __builtin_elementwise_max(rp[g] * prs + prh, z);
    };
    // ---- prologue: the table, K tile 0, and of tile 1 what the loop would have requested by now; tile 0 landed --------------------
    if (PROLOGUE) {
        // the pre-activation table by LDS-DMA too (wave w: 1 KiB of each vector), FIRST: a compiler-visible global load next to the
        // DMAs would be waited for with vmcnt(0), and being the oldest requests they are covered by every counted wait below
        const int idx = wave * 512 + lane * 8;
        if (wave * 512 < K) {
            const unsigned dst = __builtin_amdgcn_readfirstlane(smem_base + G::MAIN_BYTES + wave * 1024);
            g4d_dma16<0>(pro_scale, (unsigned)((idx < K ? idx : 0) * 2), dst);
            g4d_dma16<4096>(pro_shift, (unsigned)((idx < K ? idx : 0) * 2), dst);
        }
    }
    g4d_for<G::R>([&](auto g_c) { req(I0{}, g_c, 0); });
    constexpr int HEAD = 2 * G::Q < G::R ? 2 * G::Q : G::R;       // requests of a tile that k steps 2 and 3 issue
    {
        const int k1 = nk > 1 ? 1 : 0;
        g4d_for<HEAD>([&](auto g_c) { req(I1{}, g_c, k1); });
    }
#ifdef METRO_DBG_G4D_NO_DMA
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#else
    g4d_wait_vm_barrier<HEAD>();                                  // tile 0 (and the table) landed everywhere
#endif
    if constexpr (INPLACE && PROLOGUE) {
        rmw_pro(0);
#pragma unroll
        for (int g = 0; g < G::RB; ++g) rmw_read(0, g);
#pragma unroll
        for (int g = 0; g < G::RB; ++g) rmw_write(0, g);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }

    // FOUR fragment sets, one per k step of a K tile: while the MFMAs of k step kk run on set kk, the pixel fragments of set kk + 1
    // (read during k step kk - 1) are pre-activated and set kk + 2 is read -- neither a VALU op nor an MFMA ever waits for an LDS
    // read of its own k step (one wave per SIMD issues in order: a waiting VALU op holds up every MFMA behind it)
    half8_t af[4][MI], bf[4][NJ];
    half8_t sc[4] = {}, sh[4] = {};      // the pre-activation of a set's 8 channels per lane
    auto read_pro = [&](auto kk_c, int kt) {
        constexpr int KK = decltype(kk_c)::value;
        if constexpr (PROLOGUE && !INPLACE) {
            sc[KK] = *reinterpret_cast<const half8_t*>(pro_lds + kt * BK + KK * 16 + frag_half * 8);
            sh[KK] = *reinterpret_cast<const half8_t*>(pro_lds + 2048 + kt * BK + KK * 16 + frag_half * 8);
        }
    };
    auto read_b = [&](auto buf_c, auto kk_c, int j) {
        constexpr int BUF = decltype(buf_c)::value, KK = decltype(kk_c)::value;
        bf[KK][j] = *reinterpret_cast<const half8_t*>(smem + BUF * G::OPER_B + (b_base[j] ^ (KK << 5)));
    };
    auto read_a = [&](auto buf_c, auto kk_c, int i) {
        constexpr int BUF = decltype(buf_c)::value, KK = decltype(kk_c)::value;
        af[KK][i] = *reinterpret_cast<const half8_t*>(smem + BUF * G::OPER_A + (a_base[i] ^ (KK << 5)));
    };
    // pre-activation BN + ReLU of a pixel fragment (fp16 FMA, one rounding: resnet_v2.py:119)
    auto act_b = [&](auto kk_c, int j) {
        constexpr int KK = decltype(kk_c)::value;
#ifndef METRO_DBG_G4D_NO_PRO
        if constexpr (PROLOGUE && !INPLACE) {
            const half8_t z = {};
            bf[KK][j] = __builtin_elementwise_max(bf[KK][j] * sc[KK] + sh[KK], z);
        }
#endif
    };
    // NJ MFMAs (cout tile i against the wave's pixel tiles) on fragment set KK
    auto mma_row = [&](auto kk_c, int i) {
        constexpr int KK = decltype(kk_c)::value;
#ifndef METRO_DBG_G4D_NO_MFMA
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[KK][i], bf[KK][j], acc[i][j], 0, 0, 0);
#else
        const half8_t a_ = af[KK][i], b0_ = bf[KK][0], b1_ = bf[KK][NJ - 1];
        asm volatile("" ::"v"(a_), "v"(b0_), "v"(b1_));
#endif
    };
    // One k step: MI segments of NJ MFMAs on set KK; in segment i the cout fragment i and NJ / MI pixel fragments of set KK + 2 are
    // read from buffer RB (table rows of K tile kr) and NJ / MI pixel fragments of set KK + 1 are pre-activated; two request slots
    // behind every segment (dma(e): request slot e of this k step, or nothing).  Memory operations do not cross an asm volatile
    // statement: reads and requests stay in the segment they are written in; the emitted order inside a segment is pinned.
    constexpr int BPS = NJ / MI > 0 ? NJ / MI : 1;          // pixel fragments read / pre-activated per segment
    int rmw_kt = 0;                                         // INPLACE: the K tile whose pixel image k step 1 pre-activates
    auto kstep = [&](auto kk_c, auto rbuf_c, int kr, auto dma) {
        constexpr int KK = decltype(kk_c)::value;
        using S = std::integral_constant<int, KK>;
        using V = std::integral_constant<int, (KK + 1) & 3>;
        using R = std::integral_constant<int, (KK + 2) & 3>;
        using RB = decltype(rbuf_c);
        constexpr bool RMW = INPLACE && PROLOGUE && KK == 1 && MI == 4 && NJ == 4;   // pieces 2 I, 2 I + 1 read in segment I, written in I + 1
        constexpr int NB = RB::value ^ 1;                                             // the buffer the NEXT tile lands in
        if constexpr (RMW) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::RA) : "memory");              // own pixel pieces landed; the cout requests may fly
            rmw_pro(rmw_kt);
        }
        g4d_for<MI>([&](auto i_c) {
            constexpr int I = decltype(i_c)::value;
            if constexpr (RMW) {
                rmw_read(NB, 2 * I); rmw_read(NB, 2 * I + 1);
                if constexpr (I > 0) { rmw_write(NB, 2 * I - 2); rmw_write(NB, 2 * I - 1); }
            }
            if constexpr (I == 0) read_pro(R{}, kr);
#pragma unroll
            for (int b = 0; b < BPS; ++b)
                if (I * BPS + b < NJ) { read_b(RB{}, R{}, I * BPS + b); act_b(V{}, I * BPS + b); }
            read_a(RB{}, R{}, I);
            mma_row(S{}, I);
            constexpr int NREAD = BPS + 1 + ((I == 0 && PROLOGUE && !INPLACE) ? 2 : 0) + (RMW ? 2 + (I == 0 ? 2 : 0) : 0);
            constexpr int NVALU = INPLACE ? (RMW && I > 0 ? 16 : 0) : 8 * BPS;
#pragma unroll
            for (int m = 0; m < NJ; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (m * ((NREAD + NJ - 1) / NJ) < NREAD) {
                    if constexpr ((NREAD + NJ - 1) / NJ == 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    else __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
                if constexpr (PROLOGUE && NVALU / NJ == 2) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                if constexpr (PROLOGUE && NVALU / NJ == 4) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                if constexpr (RMW && I > 0) { if (m == 1 || m == 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); }
            }
            dma(std::integral_constant<int, 2 * I>{});
            dma(std::integral_constant<int, 2 * I + 1>{});
        });
        if constexpr (RMW) { rmw_write(NB, 2 * MI - 2); rmw_write(NB, 2 * MI - 1); }
        __builtin_amdgcn_sched_barrier(0);
    };

    // k steps 0 and 1 of tile 0 read, set 0 pre-activated (the loop pre-activates set 1 during k step 0)
    read_pro(I0{}, 0);
    read_pro(I1{}, 0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) { read_b(I0{}, I0{}, j); read_b(I0{}, I1{}, j); }
#pragma unroll
    for (int i = 0; i < MI; ++i) { read_a(I0{}, I0{}, i); read_a(I0{}, I1{}, i); }
#pragma unroll
    for (int j = 0; j < NJ; ++j) act_b(I0{}, j);

    // One K tile (BUF = kt % 2: the loop is unrolled by two).  Requests past the end re-request the LAST tile (valid memory, a
    // buffer nobody reads any more) so that the body and its waits are the same for every tile.  The R requests of tile kt + 2 go
    // into this buffer (retired at the barrier) in three shares: Q in k step 2, Q in k step 3, the rest in k step 0 of the next tile.
    auto ktile = [&](auto buf_c, int kt) {
        constexpr int BUF = decltype(buf_c)::value;
        using B = std::integral_constant<int, BUF>;
        using N = std::integral_constant<int, BUF ^ 1>;
        const int k1 = kt + 1 < nk ? kt + 1 : nk - 1;
        const int k2 = kt + 2 < nk ? kt + 2 : nk - 1;
        kstep(I0{}, B{}, kt, [&](auto e_c) {
            constexpr int E = decltype(e_c)::value;
            if constexpr (2 * G::Q + E < G::R) req(N{}, std::integral_constant<int, 2 * G::Q + E>{}, k1);
        });
        rmw_kt = k1;
        kstep(I1{}, B{}, kt, [&](auto) {});
        // every read of this buffer retired, tile kt + 1 landed (this wave's share; the barrier makes it everybody's)
#if defined(METRO_DBG_G4D_NO_BARRIER)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#else
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
        kstep(I2{}, N{}, k1, [&](auto e_c) {
            constexpr int E = decltype(e_c)::value;
            if constexpr (E < G::Q && E < G::R) req(B{}, e_c, k2);
        });
        kstep(I3{}, N{}, k1, [&](auto e_c) {
            constexpr int E = decltype(e_c)::value;
            if constexpr (E < G::Q && G::Q + E < G::R) req(B{}, std::integral_constant<int, G::Q + E>{}, k2);
        });
    };
    for (int t = 0; t < nk; t += 2) {
        ktile(I0{}, t);
        ktile(I1{}, t + 1);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave's re-requested tail has landed: the
                                                                                 // epilogue tile overlays the ring

    // ---- epilogue: accumulators (+bias, ReLU) -> LDS [pixel][cout] fp16 -> full-line stores (+ shortcut) ----
    const bool second = a.split > 0 && n0 >= a.split;     // fused pair: this cout tile belongs to one of the two outputs
    const int o_c = a.split > 0 ? (second ? a.c_out2 : a.split) : a.c_out;
    const int o_n0 = second ? n0 - a.split : n0;
    const int o_relu = second ? a.relu2 : a.relu;
    half_t* o_ptr = second ? out2 : out;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = wr * (TM / 2) + i * 32 + 8 * q + 4 * frag_half;
            const floatx4 bv = *reinterpret_cast<const floatx4*>(bias + n0 + col);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int prow = wc * (TN / 2) + j * 32 + frag_row;
                half4_t hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[i][j][4 * q + e] + bv[e];
                    if (o_relu) v = fmaxf(v, 0.f);
                    hv[e] = (half_t)v;
                }
                *reinterpret_cast<half4_t*>(smem + prow * G::OUT_ROW_BYTES + col * 2) = hv;
            }
        }
    }
    __syncthreads();
    constexpr int CPRO = TM / 8;                       // 16-byte chunks per tile row
    constexpr int EPI_ITERS = TN * CPRO / NT;
    const bool res_same = a.res_stride == 1 && a.res_offset == 0 && a.res_h == a.h_out && a.res_w == a.w_out;
    const int hw_out = a.h_out * a.w_out;
#pragma unroll 4
    for (int it = 0; it < EPI_ITERS; ++it) {
        const int idx = tid + it * NT;
        const int prow = idx / CPRO;
        const int ch = idx - prow * CPRO;
        const int m = m0 + prow;
        const int co = o_n0 + ch * 8;
        if (co + 8 > o_c) continue;                    // narrow second output of a fused pair
        uint4 v = *reinterpret_cast<const uint4*>(smem + prow * G::OUT_ROW_BYTES + ch * 16);
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

// What a tile geometry can run: 1x1, stride 1, dense NHWC fp16 in / out, c_in a multiple of 128 (two 64-channel K tiles per loop
// iteration), whole tiles of 64 MI couts x 64 NJ pixels; fused pairs split on a tile boundary with a 256-channel second output
template <int MI, int NJ>
static bool g4d_shape_ok(const MetroConvDesc& d, const ConvSplit* split) {
    if (!(d.kh == 1 && d.kw == 1 && d.stride == 1 && d.pad_top == 0 && d.pad_left == 0 && d.in_pix_stride == d.c_in &&
          d.h_in == d.h_out && d.w_in == d.w_out && d.in_dtype == METRO_F16 && d.out_dtype == METRO_F16))
        return false;
    const long m = (long)d.n * d.h_out * d.w_out;
    if (d.c_in % 128 != 0 || d.c_in < 128 || d.c_in > 2048 || d.c_out % (64 * MI) != 0 || m % (64 * NJ) != 0) return false;
    if (split != nullptr && split->split > 0 && (split->split % (64 * MI) != 0 || split->c_out2 != 256 || d.has_residual)) return false;
    return true;
}
bool conv_gemm4d_shape_ok(const MetroConvDesc& d, const ConvSplit* split) { return g4d_shape_ok<4, 4>(d, split); }
// tile geometry ids: 0 = 256 x 256 (4 x 4 tiles per wave), 1 = 128 couts x 128 pixels (2 x 2, two blocks per CU), 2 = 128 couts x 256 pixels
bool conv_gemm4d_geo_ok(const MetroConvDesc& d, const ConvSplit* split, int geo) {
    return geo == 0 ? g4d_shape_ok<4, 4>(d, split) : geo == 1 ? g4d_shape_ok<2, 2>(d, split) : geo == 2 ? g4d_shape_ok<2, 4>(d, split)
           : geo == 3 ? (d.has_prologue && g4d_shape_ok<4, 4>(d, split)) : false;
}

template <int MI, int NJ, bool INPLACE = false>
static int g4d_launch(const MetroConvDesc& d, const ConvArgs& a, const void* in, const void* w, const float* bias, const void* ps,
                      const void* pb, const half_t* r, void* out, void* out2, hipStream_t stream) {
    using G = g4d::Geo<MI, NJ>;
    const int tiles_m = (d.c_out + G::TM - 1) / G::TM;
    const int tiles_n = (a.m_total + G::TN - 1) / G::TN;
    if (d.has_prologue) {
        auto kern = conv_gemm4d_kernel<true, MI, NJ, INPLACE>;
        constexpr int lds = G::MAIN_BYTES + g4d::PRO_BYTES;
        static PerDeviceInt done;
        if (const int st = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds, done, "conv_gemm4d<pro>")) return st;
        hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(g4d::NT), lds, stream, a, static_cast<const half_t*>(in),
                           static_cast<const half_t*>(w), bias, static_cast<const half_t*>(ps), static_cast<const half_t*>(pb), r,
                           static_cast<half_t*>(out), static_cast<half_t*>(out2), tiles_m);
    } else {
        auto kern = conv_gemm4d_kernel<false, MI, NJ>;
        constexpr int lds = G::MAIN_BYTES;
        static PerDeviceInt done;
        if (const int st = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds, done, "conv_gemm4d")) return st;
        hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(g4d::NT), lds, stream, a, static_cast<const half_t*>(in),
                           static_cast<const half_t*>(w), bias, nullptr, nullptr, r, static_cast<half_t*>(out),
                           static_cast<half_t*>(out2), tiles_m);
    }
    return launch_status("conv_gemm4d");
}

int launch_conv_gemm4d(const MetroConvDesc& d, const void* in, const void* w, const float* bias, const void* ps,
                       const void* pb, const void* res, void* out, hipStream_t stream, const ConvSplit* split, int geo) {
    if (!conv_gemm4d_geo_ok(d, split, geo)) {
        set_error("conv_gemm4d: needs a 1x1 stride-1 fp16 layer with c_in %% 128 == 0 (<= 2048) and whole tiles of geometry %d "
                  "(got c_in %d, c_out %d, %d x %d x %d pixels)", geo, d.c_in, d.c_out, d.n, d.h_out, d.w_out);
        return METRO_ERR_UNSUPPORTED;
    }
    ConvArgs a = make_conv_args(d);
    void* out2 = nullptr;
    if (split != nullptr && split->split > 0) {
        a.split = split->split; a.c_out2 = split->c_out2; a.relu2 = split->relu2;
        out2 = split->out2;
    }
    static const char* const names[4] = {"256x256", "128x128", "128x256", "256x256,inplace"};
    if (note_kernel("conv_gemm4d<%s%s>%s%s", names[geo], d.has_prologue ? ",pro" : "", d.has_residual ? "+res" : "", a.split > 0 ? "+pair" : ""))
        return METRO_OK;
    const half_t* r = d.has_residual ? static_cast<const half_t*>(res) : nullptr;
    if (geo == 0) return g4d_launch<4, 4>(d, a, in, w, bias, ps, pb, r, out, out2, stream);
    if (geo == 3) return g4d_launch<4, 4, true>(d, a, in, w, bias, ps, pb, r, out, out2, stream);
    if (geo == 1) return g4d_launch<2, 2>(d, a, in, w, bias, ps, pb, r, out, out2, stream);
    return g4d_launch<2, 4>(d, a, in, w, bias, ps, pb, r, out, out2, stream);
}

}  // namespace metro
