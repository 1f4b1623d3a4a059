// block1's conv3(u) + conv1(u+1) launches with the 256-channel sum on chip: a PRODUCER / CONSUMER form of conv_pw64.hip's kernel.
//
// What this replaces and why (round 5).  conv_pw64_kernel<k64,wm4,next,projsc[,rebuild]> runs the phases of a 64-pixel tile one
// after the other in its single resident block -- operands landed -> GEMMs on the MFMA pipe -> accumulators to the LDS tile ->
// barrier -> row-wise pass (shortcut add, store, next unit's pre-activation written back) -> barrier -> second GEMM from the tile ->
// store: 3.4 us per tile for ~0.7 us of MFMA work.  Taking the 537 MB store of the sum out of block1/unit_1's launch (OUTM = 1:
// nothing reads it any more) did not make it faster (221 -> 233 us at batch 256): the launch was never HBM bound, it is a chain of
// LDS round trips and barriers that one block per CU cannot overlap with anything.  Here the two halves of a tile run
// CONCURRENTLY on the two waves of every SIMD, one tile apart:
//   * waves 0-7 (producers; twelve waves per block: two producers and one consumer per SIMD, whose latencies cover one another --
//     the first form, four producers of 64 channels with one wave per SIMD, exposed an LDS round trip per fragment: 1.3 us per
//     16-MFMA GEMM): wave w owns output channels 32 w .. + 31 of all 64 pixels.  It requests its share of the input tiles
//     two tiles ahead (LDS-DMA into a ring of three), runs the 64 -> 256 GEMMs of the tile -- conv3 of the unit on t2, the
//     block's projection shortcut on the pre-activated x0 and, REB, conv3 of the previous unit on ITS t2 (the identity shortcut
//     rebuilt instead of read: conv_pw64.hip header) -- adds them in fp16 as the separate launches would (resnet_v2.py:138) and
//     writes the sum into one of TWO [pixel][channel] LDS tiles.  It issues loads and no stores: one counted vmcnt per tile.
//   * waves 8-11 (consumers): wave w owns pixels 16 w .. + 15 of the PREVIOUS tile and all 64 output channels of the next unit's
//     conv1 (W1 in registers: 128 VGPRs).  A lane's 16-byte read of the tile IS the B fragment of v_mfma_f32_16x16x32_f16 for
//     one k step (pixel = lane & 15, channels 32 ks + 8 (lane >> 4) ..): it is stored (OUTM 0: the whole sum; OUTM 2: only the
//     pixels a strided next unit's shortcut reads, compactly; OUTM 1: not at all), pre-activated in registers
//     (resnet_v2.py:119) and multiplied -- no write-back into the tile, no barrier between the row-wise pass and the GEMM, no
//     dependence on another consumer wave.  It issues stores and no loads: nothing to wait for.
//   * ONE s_barrier per tile hands tile j from the producers to the consumers and tile j - 2's LDS tile back.
// Arithmetic (MFMA shapes, k order, fp16 rounding points) is conv_pw64_kernel's: the two forms give the same bits
// (tests/test_kernel_coverage.py compares them); metro_forward_upto stopping at such a layer runs the classic form.
#include "metro_common.h"

namespace metro {

typedef _Float16 half_t;
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

namespace b1 {
constexpr int NPROD = 8, NCONS = 4, NT = 64 * (NPROD + NCONS), TN = 64, K = 64, CB = 256, C2 = 64, NBUF = 3;
constexpr int X_BYTES = TN * 128;              // one input tile: 64 pixels x 64 fp16, 128-byte rows, chunk-swizzled
constexpr int OUT_ROW = CB * 2 + 16;           // padded rows of a [pixel][channel] tile
constexpr int OUT_BYTES = TN * OUT_ROW;
constexpr int W1S_ROW = CB * 2 + 16;           // staged W1 rows (prologue only, in the second output tile)
static_assert(48 * W1S_ROW <= OUT_BYTES, "W1's stage must fit the second output tile");
template <bool REB>
struct Lay {
    static constexpr int NIN = REB ? 3 : 2;                    // input tiles per pixel tile: t2 | x0 | (t2 of the previous unit)
    static constexpr int X_OFF = 0;                            // [NBUF][NIN][X_BYTES]
    static constexpr int OUT_OFF = NBUF * NIN * X_BYTES;       // two tiles
    static constexpr int PAR_OFF = OUT_OFF + 2 * OUT_BYTES;
    // bias3[256] f32 | bias_sc[256] f32 | bias_b[256] f32 | bias2[64] f32 | pro scale[64] shift[64] fp16 | scale2[256] | shift2[256] fp16
    static constexpr int B3 = 0, BSC = 1024, BB = 2048, B2 = 3072, PRO = 3328, SC2 = 3584, SH2 = 4096, PAR_BYTES = 4608;
    // rows 48 .. 63 of W1 in fragment order [k step][lane] (8 KiB): the consumers keep rows 0 .. 47 in registers (96 VGPRs); with all
    // 64 rows they would need 128 + ~45 of the 168 registers three waves per SIMD leave a wave
    static constexpr int W1T_OFF = PAR_OFF + PAR_BYTES;
    static constexpr int LDS = W1T_OFF + 8 * 64 * 16;
};
}  // namespace b1

struct B1Args {
    const half_t* in;          // [m_total][64]  conv2 output of the unit
    const half_t* w;           // [256][64]      conv3 of the unit
    const float* bias;         // [256]
    const half_t* x_sc;        // [m_total][64]  the block's raw input (pooled stem output)
    const half_t* w_sc;        // [256][64]      projection shortcut
    const float* bias_sc;      // [256]
    const half_t* pro_scale;   // [64]           pre-activation of the block's first unit (applies to x_sc)
    const half_t* pro_shift;
    const half_t* in_b;        // REB: [m_total][64] conv2 output of the PREVIOUS unit
    const half_t* w_b;         // REB: [256][64] its conv3
    const float* bias_b;       // REB: [256]
    const half_t* w2;          // [64][256]      conv1 of the next unit (BN folded)
    const float* bias2;        // [64]
    const half_t* scale2;      // [256]          pre-activation of the next unit
    const half_t* shift2;
    half_t* out;               // OUTM 0: [m_total][256]
    half_t* out_sub;           // OUTM 2: compact [n][h_sub][w_sub][256]
    half_t* out2;              // [m_total][64]
    int m_total, n_tiles;
    int h_out, w_out, lw_out, sub_off, h_sub, w_sub;
};

__device__ __forceinline__ int b1_swz(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ void b1_dma16(const void* gsrc, unsigned lds_addr) {
    asm volatile(
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, off"
        :
        : "v"(gsrc), "s"(lds_addr));
}
template <int N>
__device__ __forceinline__ void b1_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// fp16(a + bias) for four accumulators as two v_cvt_pk_f16_f32 (round to nearest even, like the (half_t) casts of conv_pw64.hip).
// Written out because the launch is INSTRUCTION-ISSUE bound (SQ counters: ~3 200 wave instructions per 64-pixel tile, the SIMDs
// issuing 70 % of the time, matrix pipe 32 % busy): hipcc's SLP pass turned the element-wise form into v_pk_add_f32 on shuffled
// register pairs + single v_cvt_f16_f32 + v_pack_b32_f16 / v_alignbit_b32 -- 313 instructions per producer tile instead of ~200.
typedef unsigned int b1_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ half4_t b1_cvt4(float a0, float a1, float a2, float a3) {
    b1_u32x2 r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r.x) : "v"(a0), "v"(a1));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r.y) : "v"(a2), "v"(a3));
    return __builtin_bit_cast(half4_t, r);
}
typedef float b1_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ half4_t b1_bias_cvt(const floatx16& acc, int q, const floatx4& bv) {
    // two v_pk_add_f32 on the accumulator's own (even-aligned) register pairs, two v_cvt_pk_f16_f32: 4 issue slots per 4 outputs
    b1_f32x2 lo = {acc[4 * q], acc[4 * q + 1]}, hi = {acc[4 * q + 2], acc[4 * q + 3]};
    const b1_f32x2 blo = {bv[0], bv[1]}, bhi = {bv[2], bv[3]};
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(lo) : "v"(lo), "v"(blo));
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(hi) : "v"(hi), "v"(bhi));
    return b1_cvt4(lo.x, lo.y, hi.x, hi.y);
}
__device__ __forceinline__ void b1_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <bool REB, int OUTM>
__global__ __launch_bounds__(b1::NT) void conv_b1_chain_kernel(B1Args a) {
    using namespace b1;
    using L = Lay<REB>;
    constexpr int NIN = L::NIN;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void lds_void_t;
    const unsigned smem_base = (unsigned)(size_t)(lds_void_t*)smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave < NPROD;
    const int G = gridDim.x;
    const int t0 = blockIdx.x;
    if (t0 >= a.n_tiles) return;
    const int T = (a.n_tiles - t0 + G - 1) / G;             // tiles of this block: t0, t0 + G, ...
    char* par = smem + L::PAR_OFF;
    {
        float* f = reinterpret_cast<float*>(par);
        if (tid < 256) {
            f[L::B3 / 4 + tid] = a.bias[tid];
            f[L::BSC / 4 + tid] = a.bias_sc[tid];
            if (REB) f[L::BB / 4 + tid] = a.bias_b[tid];
            reinterpret_cast<half_t*>(par + L::SC2)[tid] = a.scale2[tid];
            reinterpret_cast<half_t*>(par + L::SH2)[tid] = a.shift2[tid];
        }
        if (tid < 512) {         // W1 rows 48 .. 63 as 16x16x32 A fragments: [k step tid >> 6][lane tid & 63]
            const int ks = tid >> 6, ln = tid & 63;
            *reinterpret_cast<uint4*>(smem + L::W1T_OFF + tid * 16) =
                *reinterpret_cast<const uint4*>(a.w2 + (size_t)(48 + (ln & 15)) * CB + ks * 32 + (ln >> 4) * 8);
        }
        // W1 rows 0 .. 47 (the consumers' register-resident part) take the same road as every weight-stationary kernel's fragments
        // since round 5 (metro_common.h: load_w_frags_staged): fetched ONCE per block in whole 512-byte rows into the second output
        // tile's LDS (idle until the producers' tile 1; rows padded to 528 bytes) instead of by each of the four consumer waves in
        // 64-byte pieces -- 192 requests per block instead of 1 536 (the L2s answer a near-constant request rate: DESIGN.md section 5)
        for (int c = tid; c < 48 * 32; c += NT)
            *reinterpret_cast<uint4*>(smem + L::OUT_OFF + OUT_BYTES + (c >> 5) * W1S_ROW + (c & 31) * 16) =
                *reinterpret_cast<const uint4*>(a.w2 + (size_t)(c >> 5) * CB + (c & 31) * 8);
        if (tid < 64) {
            f[L::B2 / 4 + tid] = a.bias2[tid];
            reinterpret_cast<half_t*>(par + L::PRO)[tid] = a.pro_scale[tid];
            reinterpret_cast<half_t*>(par + L::PRO)[64 + tid] = a.pro_shift[tid];
        }
    }
    const int frag_row = lane & 31, frag_half = lane >> 5;

    // ONE branch per role from here to the end of the kernel (a wave's role never changes): the producers' weight matrices and the
    // consumers' W1 never share a register budget.  Every wave executes 1 + (T + 1) barriers.
    if (producer) {
        // W3 | Wsc | (W3 of the previous unit) as 32x32x16 A fragments of this wave's 32 output channels
        // (staged through a wave-private LDS scratch in whole 128-byte rows: 8 requests per instruction instead of 32)
        half8_t wf[4], wsf[4], wbf[4];
        static_assert(NPROD * W_STAGE_BYTES <= L::OUT_OFF + OUT_BYTES, "the staging scratch must stay below the second output tile (W1's stage)");
        load_w_frags_staged<K>(a.w + (size_t)wave * 32 * K, wf, smem + wave * W_STAGE_BYTES, lane);
        load_w_frags_staged<K>(a.w_sc + (size_t)wave * 32 * K, wsf, smem + wave * W_STAGE_BYTES, lane);
        if constexpr (REB) load_w_frags_staged<K>(a.w_b + (size_t)wave * 32 * K, wbf, smem + wave * W_STAGE_BYTES, lane);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            // pin: an empty asm makes the fragment the result of an instruction that cannot be re-executed -- under register pressure
            // hipcc otherwise REMATERIALISES such loads inside the tile loop (24 global loads per tile in the first build)
            asm volatile("" : "+v"(wf[kk]), "+v"(wsf[kk]));
            if constexpr (REB) asm volatile("" : "+v"(wbf[kk]));
        }
        // an input tile = 8 LDS-DMA instructions of 8 rows x 128 B: producer w issues instruction w of every input
        const int xrow = wave * 8 + (lane >> 3);
        const int xoff = xrow * K + (((lane & 7) ^ b1_swz(xrow)) * 8);
        auto issue_tile = [&](int tile, int slot) {           // whole tiles only (the launcher checks h * w % 64 == 0): no zero page
            const size_t m0 = (size_t)tile * TN;
            const unsigned dst = smem_base + L::X_OFF + slot * NIN * X_BYTES + wave * 1024;
            b1_dma16(a.in + m0 * K + xoff, __builtin_amdgcn_readfirstlane(dst));
            b1_dma16(a.x_sc + m0 * K + xoff, __builtin_amdgcn_readfirstlane(dst + X_BYTES));
            if constexpr (REB) b1_dma16(a.in_b + m0 * K + xoff, __builtin_amdgcn_readfirstlane(dst + 2 * X_BYTES));
        };
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");     // parameters in LDS, weights in registers
        // The block's pre-activation (resnet_v2.py:119) is applied to the x0 tile ONCE, in place, by the wave whose LDS-DMA
        // brought the rows in (lane l's 16 bytes: row 8 w + (l >> 3), logical chunk (l & 7) ^ swizzle), right behind its own
        // vmcnt wait and in front of the tile's barrier -- not by every producer on every fragment it reads (eight waves repeating
        // the same 64 packed operations per tile: a quarter of the launch's VALU work, which is what bounds it).
        half8_t pre_s, pre_b;
        {
            const int ch = ((lane & 7) ^ b1_swz(xrow)) * 8;
            pre_s = *reinterpret_cast<const half8_t*>(par + L::PRO + ch * 2);
            pre_b = *reinterpret_cast<const half8_t*>(par + L::PRO + 128 + ch * 2);
        }
        issue_tile(t0, 0);
        if (T > 1) issue_tile(t0 + G, 1);
        // the conv3 bias of this wave's 32 channels in registers (16 VGPRs); the shortcuts' biases stay in LDS (four 16-byte reads each
        // per tile: with them in registers too the producers spill)
        floatx4 b3_r[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) b3_r[q] = *reinterpret_cast<const floatx4*>(par + L::B3 + (wave * 32 + 8 * q + 4 * frag_half) * 4);
        const float* bias_sc_l = reinterpret_cast<const float*>(par + L::BSC);
        const float* bias_b_l = reinterpret_cast<const float*>(par + L::BB);
        // fragment address of pixel half p (32 pixels), k step kk: row p * 32 + frag_row, chunk 2 kk + frag_half, swizzled (the swizzle
        // (row >> 1) & 7 does not see bit 5 of the row: the two halves differ by 32 rows x 128 B)
        const int boff0 = frag_row * 128 + ((frag_half ^ b1_swz(frag_row)) << 4);
        int slot = 0;                                       // ring slot of tile j
        for (int j = 0; j <= T; ++j) {
            // tile j's operands have landed (this wave's share; the barrier makes it every producer's) and the consumers are done
            // with the LDS tile of j - 2, which tile j overwrites
            if (j < T) {
                if (j + 1 < T) b1_wait_vm<NIN>();            // tile j + 1's requests may stay in flight
                else b1_wait_vm<0>();
            }
            if (j < T) {
                half8_t* mine = reinterpret_cast<half8_t*>(smem + L::X_OFF + (slot * NIN + 1) * X_BYTES + wave * 1024 + lane * 16);
                const half8_t z = {};
                *mine = __builtin_elementwise_max(*mine * pre_s + pre_b, z);
            }
            b1_barrier();
            if (j + 2 < T) issue_tile(t0 + (j + 2) * G, slot >= 1 ? slot - 1 : 2);       // slot of j + 2 = (slot + 2) % 3
            if (j < T) {
                const char* xs = smem + L::X_OFF + slot * NIN * X_BYTES;
                char* ot = smem + L::OUT_OFF + (j & 1) * OUT_BYTES;
                floatx16 acc[2];                             // [pixel half]
                half4_t xsum[2][4];                          // the shortcut of the tile: [pixel half][quad], packed fp16
                // ---- the projection shortcut: fp16(Wsc . pre(x0) + bsc) ----------------------------------------------------------
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const half8_t bs = *reinterpret_cast<const half8_t*>(xs + X_BYTES + p * 4096 + (boff0 ^ (kk << 5)));   // pre-activated above
                        acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wsf[kk], bs, acc[p], 0, 0, 0);
                    }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const floatx4 bv = *reinterpret_cast<const floatx4*>(bias_sc_l + wave * 32 + 8 * q + 4 * frag_half);
#pragma unroll
                    for (int p = 0; p < 2; ++p) xsum[p][q] = b1_bias_cvt(acc[p], q, bv);
                }
                // ---- REB: + fp16(W3_prev . t2_prev + b3_prev): the previous unit's sum x_prev (its own fp16 Add, resnet_v2.py:138) ----
                if constexpr (REB) {
#pragma unroll
                    for (int p = 0; p < 2; ++p)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                        for (int p = 0; p < 2; ++p) {
                            const half8_t bb = *reinterpret_cast<const half8_t*>(xs + 2 * X_BYTES + p * 4096 + (boff0 ^ (kk << 5)));
                            acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wbf[kk], bb, acc[p], 0, 0, 0);
                        }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const floatx4 bv = *reinterpret_cast<const floatx4*>(bias_b_l + wave * 32 + 8 * q + 4 * frag_half);
#pragma unroll
                        for (int p = 0; p < 2; ++p) xsum[p][q] = b1_bias_cvt(acc[p], q, bv) + xsum[p][q];
                    }
                }
                // ---- conv3 of this unit + its fp16 Add, into the LDS tile -------------------------------------------------------
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const half8_t bf = *reinterpret_cast<const half8_t*>(xs + p * 4096 + (boff0 ^ (kk << 5)));
                        acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[kk], bf, acc[p], 0, 0, 0);
                    }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int col = wave * 32 + 8 * q + 4 * frag_half;
#pragma unroll
                    for (int p = 0; p < 2; ++p)
                        *reinterpret_cast<half4_t*>(ot + (p * 32 + frag_row) * OUT_ROW + col * 2) = b1_bias_cvt(acc[p], q, b3_r[q]) + xsum[p][q];
                }
            }
            slot = slot == 2 ? 0 : slot + 1;
        }
    } else {
        // W1 of the next unit over K = 256 as 16x16x32 A fragments: rows 0 .. 47 in registers, rows 48 .. 63 in LDS (above)
        half8_t w2r[3][8];
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");     // parameters and W1's stage in LDS
        // (read before this wave's first loop barrier; the producers overwrite the stage behind their second one)
#pragma unroll
        for (int mt = 0; mt < 3; ++mt)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                w2r[mt][ks] = *reinterpret_cast<const half8_t*>(smem + L::OUT_OFF + OUT_BYTES + (mt * 16 + (lane & 15)) * W1S_ROW + ks * 64 + (lane >> 4) * 16);
                asm volatile("" : "+v"(w2r[mt][ks]));        // pinned in registers (see the producers' weights)
            }
        const int wb = wave - NPROD;
        const int px = wb * 16 + (lane & 15), kg = lane >> 4;
        for (int j = 0; j <= T; ++j) {
            b1_barrier();                                   // tile j - 1's sum is complete in its LDS tile
            if (j == 0) continue;
            // ---- 16 pixels of tile j - 1 x all channels: store | pre-activate | conv1 of the next unit -------------------------------
            const int tile = t0 + (j - 1) * G;
            const char* ot = smem + L::OUT_OFF + ((j - 1) & 1) * OUT_BYTES;
            const size_t m = (size_t)tile * TN + px;
            bool sub_ok = false;
            size_t sub_row = 0;
            if constexpr (OUTM == 2) {
                // pixels (sub_off + 2 i, sub_off + 2 j) of the map: a wave's 16 pixels lie in one map row (w_out >= 16, a power of two)
                const int hw = a.h_out * a.w_out;
                const int m0 = tile * TN + wb * 16;
                const int img = m0 / hw, rem = m0 - img * hw;
                const int hr = (rem >> a.lw_out) - a.sub_off;
                const int wo = ((rem & (a.w_out - 1)) + (lane & 15)) - a.sub_off;
                sub_ok = hr >= 0 && (hr & 1) == 0 && (hr >> 1) < a.h_sub && wo >= 0 && (wo & 1) == 0 && (wo >> 1) < a.w_sub;
                sub_row = ((size_t)img * a.h_sub + (hr >> 1)) * a.w_sub + (wo >> 1);
            }
            floatx4 dacc[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) dacc[mt] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const uint4 v = *reinterpret_cast<const uint4*>(ot + px * OUT_ROW + ks * 64 + kg * 16);
                if constexpr (OUTM == 0) store_out16<1>(a.out + m * CB + ks * 32 + kg * 8, v);
                if constexpr (OUTM == 2) {
                    if (sub_ok) store_out16<1>(a.out_sub + sub_row * CB + ks * 32 + kg * 8, v);
                }
                const half8_t s = *reinterpret_cast<const half8_t*>(par + L::SC2 + (ks * 32 + kg * 8) * 2);
                const half8_t b = *reinterpret_cast<const half8_t*>(par + L::SH2 + (ks * 32 + kg * 8) * 2);
                const half8_t z = {};
                half8_t pv = *reinterpret_cast<const half8_t*>(&v);
                pv = __builtin_elementwise_max(pv * s + b, z);        // the next unit's pre-activation (fp16 BN + ReLU, one rounding)
                const half8_t w3 = *reinterpret_cast<const half8_t*>(smem + L::W1T_OFF + (ks * 64 + lane) * 16);
#pragma unroll
                for (int mt = 0; mt < 3; ++mt) dacc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2r[mt][ks], pv, dacc[mt], 0, 0, 0);
                dacc[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3, pv, dacc[3], 0, 0, 0);
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int co = mt * 16 + kg * 4;
                const floatx4 bv = *reinterpret_cast<const floatx4*>(par + L::B2 + co * 4);
                half4_t hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) hv[e] = (half_t)fmaxf(dacc[mt][e] + bv[e], 0.f);
                *reinterpret_cast<half4_t*>(a.out2 + m * C2 + co) = hv;
            }
        }
    }
}

// thread-local test switch (metro_conv_b1_form): 1 = always the classic single-role kernel of conv_pw64.hip
static thread_local int g_b1_force_classic = 0;
void conv_b1_set_form(int classic) { g_b1_force_classic = classic; }
bool classic_forms_forced() { return g_b1_force_classic != 0; }

bool conv_b1_chain_preferred() {
    static const int enabled = tuning_knob("METRO_B1_SPLIT", 1);
    return enabled != 0 && g_b1_force_classic == 0;
}

template <bool REB, int OUTM>
static int launch_b1(B1Args a, hipStream_t stream) {
    if (note_kernel("conv_b1_chain<%s%s>", REB ? "rebuild" : "projsc", OUTM == 1 ? ",noout" : OUTM == 2 ? ",subout" : ""))
        return METRO_OK;
    auto kern = conv_b1_chain_kernel<REB, OUTM>;
    constexpr int lds = b1::Lay<REB>::LDS;
    a.n_tiles = a.m_total / b1::TN;
    static PerDeviceInt cap;
    int grid_cap = 0;
    if (const int st = ensure_dyn_lds_and_grid_cap(reinterpret_cast<const void*>(kern), b1::NT, lds, cap, "conv_b1_chain", 1, &grid_cap))
        return st;
    const int grid = a.n_tiles < grid_cap ? a.n_tiles : grid_cap;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(b1::NT), lds, stream, a);
    return launch_status("conv_b1_chain");
}

// d: the conv3 layer (1x1, 64 -> 256, no prologue, no residual tensor); psc, f2 as for conv_pw64's mode 3; rb: rebuild / output mode
int launch_conv_b1_chain(const MetroConvDesc& d, const void* in, const void* w, const float* bias, void* out, hipStream_t stream,
                         const ConvFuse2& f2, const ConvProjSc& psc, const ConvRebuild& rb) {
    const int hw = d.h_out * d.w_out;
    if (!(d.kh == 1 && d.kw == 1 && d.stride == 1 && d.c_in == 64 && d.c_out == 256 && d.in_pix_stride == 64 && !d.has_prologue &&
          !d.has_residual && !d.relu && d.in_dtype == METRO_F16 && d.out_dtype == METRO_F16 && f2.c2 == 64 && hw % 64 == 0 &&
          d.w_out >= 16 && (d.w_out & (d.w_out - 1)) == 0)) {
        set_error("conv_b1_chain: built for the conv3 + next conv1 launches of block1 (1x1, 64 -> 256 -> 64, whole 64-pixel tiles, map width a power of two >= 16)");
        return METRO_ERR_UNSUPPORTED;
    }
    B1Args a;
    a.in = static_cast<const half_t*>(in); a.w = static_cast<const half_t*>(w); a.bias = bias;
    a.x_sc = static_cast<const half_t*>(psc.x); a.w_sc = static_cast<const half_t*>(psc.w_sc); a.bias_sc = psc.bias_sc;
    a.pro_scale = static_cast<const half_t*>(psc.pro_scale); a.pro_shift = static_cast<const half_t*>(psc.pro_shift);
    a.in_b = static_cast<const half_t*>(rb.t2_prev); a.w_b = static_cast<const half_t*>(rb.w3_prev); a.bias_b = rb.bias3_prev;
    a.w2 = static_cast<const half_t*>(f2.w2); a.bias2 = f2.bias2;
    a.scale2 = static_cast<const half_t*>(f2.scale2); a.shift2 = static_cast<const half_t*>(f2.shift2);
    a.out = static_cast<half_t*>(out); a.out_sub = static_cast<half_t*>(rb.out_sub); a.out2 = static_cast<half_t*>(f2.out2);
    a.m_total = d.n * hw; a.n_tiles = 0;
    a.h_out = d.h_out; a.w_out = d.w_out; a.lw_out = 0;
    while ((1 << a.lw_out) < d.w_out) ++a.lw_out;
    a.sub_off = rb.sub_off; a.h_sub = rb.h_sub; a.w_sub = rb.w_sub;
    const bool reb = rb.t2_prev != nullptr;
    if (reb) {
        if (rb.out_mode == 2) return launch_b1<true, 2>(a, stream);
        if (rb.out_mode == 0) return launch_b1<true, 0>(a, stream);
        set_error("conv_b1_chain: a rebuilt residual whose sum is neither stored nor sub-sampled has no consumer");
        return METRO_ERR_INVALID_ARG;
    }
    if (rb.out_mode == 1) return launch_b1<false, 1>(a, stream);
    if (rb.out_mode == 0) return launch_b1<false, 0>(a, stream);
    set_error("conv_b1_chain: the sub-sampled copy exists with the rebuilt residual only");
    return METRO_ERR_INVALID_ARG;
}

}  // namespace metro
