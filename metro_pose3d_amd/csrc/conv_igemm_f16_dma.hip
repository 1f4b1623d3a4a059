// fp16 implicit-GEMM convolution: LDS-DMA ring + coalesced epilogue.
//
// D[cout][pixel] = sum_{tap,c} W[cout][tap][c] * X[pixel + tap][c]: MFMA A operand = weight rows, B operand = pixels,
// v_mfma_f32_32x32x16_f16, LDS operand tiles [rows][BK] fp16 with the 16-byte chunk index XOR-swizzled by the row:
//   * operand tiles go global -> LDS directly with global_load_lds_dwordx4 (no VGPR staging, no
//     ds_write pass): one wave-instruction fills 8 tile rows x 128 B.  The LDS image of an
//     LDS-DMA is lane-linear, so the XOR chunk swizzle is applied to the per-lane SOURCE address;
//   * TF zero padding, ragged tile edges and the c_in tail are served from a 16-byte zero page
//     (a lane cannot be masked out of an LDS-DMA without leaving stale bytes in its slot);
//   * a STAGES-deep ring keeps STAGES-1 K-steps in flight: each iteration waits with a COUNTED
//     s_waitcnt vmcnt(N) for its own step only, one raw s_barrier per step orders both the landing
//     of step k (RAW) and the reuse of the slot freed by step k-1 (WAR);
//   * the pre-activation BatchNorm+ReLU (reference resnet_v2.py:119,229) is applied to the pixel
//     fragment after its ds_read_b128 (4 v_pk_fma_f16 + 4 v_pk_max_f16), scale/shift in LDS;
//   * the epilogue transposes the accumulators through the (now idle) ring: every lane then owns
//     16 contiguous bytes of an NHWC row, so residual loads and output stores are full-line.
//     conv+bias is rounded to fp16 before the shortcut add, exactly the reference's fp16 graph
//     (BiasAdd output is fp16, then Add: resnet_v2.py:134-138 under tfu.py:426-440).
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

__device__ __attribute__((aligned(16))) unsigned int g_zero_page[4];   // zero-initialised

template <int WAVES_M_, int WAVES_N_, int WM_, int WN_, int STAGES_, int BK_ = 64>
struct DmaCfg {
    static constexpr int WAVES_M = WAVES_M_, WAVES_N = WAVES_N_, WM = WM_, WN = WN_, STAGES = STAGES_;
    static constexpr int BK = BK_;                        // K elements per step: 64 or 32
    static constexpr int CPR = BK / 8;                    // 16-byte chunks per tile row
    static constexpr int RPI = 64 / CPR;                  // tile rows filled by one DMA wave-instruction
    static constexpr int NW = WAVES_M * WAVES_N;
    static constexpr int NT = 64 * NW;
    static constexpr int TM = WAVES_M * WM * 32;          // output channels per block
    static constexpr int TN = WAVES_N * WN * 32;          // pixels per block
    static constexpr int WI = TM / (RPI * NW);            // weight DMA instructions per wave per step
    static constexpr int XI = TN / (RPI * NW);            // pixel  DMA instructions per wave per step
    static constexpr int LPS = WI + XI;
    static constexpr int ROW_BYTES = BK * 2;
    static constexpr int STAGE_BYTES = (TM + TN) * ROW_BYTES;
    static constexpr int RING_BYTES = STAGES * STAGE_BYTES;
    static constexpr int OUT_ROW_BYTES = TM * 2 + 16;     // epilogue tile [TN][TM] fp16, padded rows
    static constexpr int OUT_BYTES = TN * OUT_ROW_BYTES;
    static constexpr int MAIN_BYTES = RING_BYTES > OUT_BYTES ? RING_BYTES : OUT_BYTES;
    static constexpr int PRO_BYTES = 2 * 2048 * 2;        // scale + shift, c_in <= 2048
    static_assert(TM % (RPI * NW) == 0 && TN % (RPI * NW) == 0, "tile/loader mismatch");
    static_assert(BK == 64 || BK == 32, "BK must be 32 or 64");
    static_assert(STAGES >= 1 && STAGES <= 6, "ring depth 1..6");
};

// chunk swizzle so that ds_read_b128 of 32 rows x one chunk hits 16 distinct 16-byte slots per
// 16-lane group: BK=64 (128-byte rows, 2 per bank row): (row>>1)&7; BK=32 (64-byte rows): (row>>2)&3
template <int BK>
__device__ __forceinline__ int swzk(int row) { return BK == 64 ? (row >> 1) & 7 : (row >> 2) & 3; }

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

// One LDS-DMA wave-instruction: 64 lanes x 16 bytes, lane l lands at lds_addr + 16*l.
// Inline asm on purpose: hipcc tracks the builtin form as an LDS write that may alias every
// later ds_read and drains it with s_waitcnt vmcnt(0), which serialises the ring.  The asm form
// is invisible to its bookkeeping; completion is ordered by the counted waits below.
// lds_addr must be wave-uniform (it goes through M0, saved/restored around the instruction).
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_addr) {
    // M0 is written in the same statement that consumes it and is not preserved: nothing else in
    // these kernels uses M0 (gfx9+ LDS instructions do not need it).
    asm volatile(
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, off"
        :
        : "v"(gsrc), "s"(lds_addr));   // no "memory" clobber: ordering is carried by the barrier asm
}

__device__ __forceinline__ unsigned lds_offset_of(const void* p) {
    return (unsigned)(size_t)(lds_void_t*)p;
}

template <int N>
__device__ __forceinline__ void wait_vm_and_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

struct Fuse2Args {
    const half_t* w2; const float* bias2; const half_t* scale2; const half_t* shift2; half_t* out2; int c2;
};

template <class Cfg, bool PROLOGUE, bool FASTK, bool FUSE2>
__device__ __forceinline__ void conv_dma_body(
    const ConvArgs& a, const half_t* __restrict__ in, const half_t* __restrict__ w,
    const float* __restrict__ bias, const half_t* __restrict__ pro_scale,
    const half_t* __restrict__ pro_shift, const half_t* __restrict__ residual,
    void* __restrict__ out, int out_f32, int tiles_m, void* __restrict__ out2, const Fuse2Args& f2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BK = Cfg::BK, STAGES = Cfg::STAGES, NW = Cfg::NW;
    constexpr int CPR = Cfg::CPR, RPI = Cfg::RPI;
    constexpr int SLICES = BK / 16;
    // FUSE2 LDS regions behind the main area: W2 image [c_out/64 chunks][64 rows][64 k] + scale2|shift2
    // (W2 fragments fetched from L2 per tile instead measured 16 us slower per launch)
    constexpr int F2_W_OFF = Cfg::MAIN_BYTES + (PROLOGUE ? Cfg::PRO_BYTES : 0);
    constexpr int F2_W_BYTES = 64 * Cfg::TM * 2;
    constexpr int F2_P_OFF = F2_W_OFF + F2_W_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave / Cfg::WAVES_N;
    const int wave_n = wave % Cfg::WAVES_N;

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
    const int m0 = tile_n * Cfg::TN;
    const int n0 = tile_m * Cfg::TM;

    const int taps = a.kh * a.kw;
    const int k_total = taps * a.c_in;
    const int kc_steps = (a.c_in + BK - 1) / BK;
    const int nk = taps * kc_steps;
    const int hw_out = a.h_out * a.w_out;
    const half_t* zero = reinterpret_cast<const half_t*>(g_zero_page);
    const unsigned smem_base = lds_offset_of(smem);

    half_t* pro_lds = reinterpret_cast<half_t*>(smem + Cfg::MAIN_BYTES);   // scale | shift (behind the ring)

    // ---- per-lane DMA source coordinates ----------------------------------------------------
    const int lrow = lane / CPR;    // row within the RPI-row group one DMA instruction fills
    const int lch = lane % CPR;     // physical 16-byte chunk within the row
    const half_t* wsrc[Cfg::WI];    // FASTK: running pointer; else row base
    int wkoff[Cfg::WI], winc[Cfg::WI];
    bool wvalid[Cfg::WI];
#pragma unroll
    for (int i = 0; i < Cfg::WI; ++i) {
        const int row = (i * NW + wave) * RPI + lrow;
        const int co = n0 + row;
        wvalid[i] = co < a.c_out;
        wkoff[i] = (lch ^ swzk<BK>(row)) * 8;
        const half_t* base = w + (size_t)(wvalid[i] ? co : 0) * k_total;
        // weight rows are contiguous across taps: with c_in % BK == 0 every step is +BK elements
        wsrc[i] = FASTK ? (wvalid[i] ? base + wkoff[i] : zero) : base;
        winc[i] = wvalid[i] ? BK : 0;
    }
    int xh[Cfg::XI], xw[Cfg::XI], xn[Cfg::XI], xkoff[Cfg::XI], xinc[Cfg::XI];
    const half_t* xptr[Cfg::XI];
    bool xvalid[Cfg::XI];
#pragma unroll
    for (int i = 0; i < Cfg::XI; ++i) {
        const int row = (i * NW + wave) * RPI + lrow;
        const int m = m0 + row;
        xvalid[i] = m < a.m_total;
        const int mm = xvalid[i] ? m : 0;
        const int img = mm / hw_out;
        const int rem = mm - img * hw_out;
        const int ho = rem / a.w_out;
        const int wo = rem - ho * a.w_out;
        xh[i] = ho * a.stride - a.pad_top;
        xw[i] = wo * a.stride - a.pad_left;
        xn[i] = img * a.h_in * a.w_in;
        xkoff[i] = (lch ^ swzk<BK>(row)) * 8;
        xptr[i] = zero;
        xinc[i] = 0;
    }

    // ---- DMA issue ----------------------------------------------------------------------------
    // The instruction stream around the MFMAs is the scarce resource (PMC: ~100 VALU per K-step were
    // address arithmetic), so with FASTK (c_in % BK == 0) source pointers are INCREMENTAL: within a
    // tap every step moves BK channels along a contiguous row; only a new tap recomputes the pixel
    // pointers; out-of-range lanes sit on the zero page with increment 0.  !FASTK (c_in tail: toy
    // specs) recomputes everything per step.  One step is issued in SLICES parts so it interleaves
    // with the MFMA groups of the step being computed.
    int is_tap = 0, is_c0 = 0, is_r = 0, is_s = 0;
    auto x_new_tap = [&]() {
#pragma unroll
        for (int i = 0; i < Cfg::XI; ++i) {
            const int hi = xh[i] + is_r * a.dil;
            const int wi = xw[i] + is_s * a.dil;
            const bool ok = xvalid[i] && (unsigned)hi < (unsigned)a.h_in && (unsigned)wi < (unsigned)a.w_in;
            xptr[i] = ok ? in + (size_t)(xn[i] + hi * a.w_in + wi) * a.in_pix_stride + xkoff[i] : zero;
            xinc[i] = ok ? BK : 0;
        }
    };
    if (FASTK) x_new_tap();
    unsigned is_wl = 0, is_xl = 0;
    int is_kbase = 0;
    auto issue_begin = [&](int buf) {
        is_wl = __builtin_amdgcn_readfirstlane(smem_base + buf * Cfg::STAGE_BYTES + wave * RPI * Cfg::ROW_BYTES);
        is_xl = is_wl + Cfg::TM * Cfg::ROW_BYTES;
        is_kbase = is_tap * a.c_in + is_c0;
    };
    auto issue_part = [&](int part) {
#pragma unroll
        for (int i = 0; i < Cfg::WI; ++i) {
            if ((i % SLICES) != part) continue;
            if (FASTK) {
                dma16(wsrc[i], is_wl + i * NW * RPI * Cfg::ROW_BYTES);
                wsrc[i] += winc[i];
            } else {
                const int c = is_c0 + wkoff[i];
                const half_t* src = (wvalid[i] && c < a.c_in) ? wsrc[i] + is_kbase + wkoff[i] : zero;
                dma16(src, is_wl + i * NW * RPI * Cfg::ROW_BYTES);
            }
        }
#pragma unroll
        for (int i = 0; i < Cfg::XI; ++i) {
            if (((i + Cfg::WI) % SLICES) != part) continue;
            if (FASTK) {
                dma16(xptr[i], is_xl + i * NW * RPI * Cfg::ROW_BYTES);
                xptr[i] += xinc[i];
            } else {
                const int hi = xh[i] + is_r * a.dil;
                const int wi = xw[i] + is_s * a.dil;
                const int c = is_c0 + xkoff[i];
                const bool ok = xvalid[i] && c < a.c_in && (unsigned)hi < (unsigned)a.h_in &&
                                (unsigned)wi < (unsigned)a.w_in;
                const half_t* src = ok ? in + (size_t)(xn[i] + hi * a.w_in + wi) * a.in_pix_stride + c : zero;
                dma16(src, is_xl + i * NW * RPI * Cfg::ROW_BYTES);
            }
        }
    };
    auto issue_end = [&]() {
        is_c0 += BK;
        if (is_c0 >= a.c_in) {
            is_c0 = 0;
            ++is_tap;
            if (++is_s == a.kw) { is_s = 0; ++is_r; }
            if (FASTK && is_tap < taps) x_new_tap();
        }
    };
    auto issue_step = [&](int buf) {
        issue_begin(buf);
#pragma unroll
        for (int p = 0; p < SLICES; ++p) issue_part(p);
        issue_end();
    };

    floatx16 acc[Cfg::WM][Cfg::WN];
#pragma unroll
    for (int i = 0; i < Cfg::WM; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::WN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int frag_row = lane & 31;
    const int frag_half = lane >> 5;

    // WITH_ISSUE is a compile-time tag so the step body stays one straight-line basic block
    auto compute_step = [&](int buf, int c0, auto with_issue) {
        const char* wl = smem + buf * Cfg::STAGE_BYTES;
        const char* xl = wl + Cfg::TM * Cfg::ROW_BYTES;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            half8_t af[Cfg::WM], bf[Cfg::WN];
            const int chunk = kk * 2 + frag_half;
#pragma unroll
            for (int i = 0; i < Cfg::WM; ++i) {
                const int row = (wave_m * Cfg::WM + i) * 32 + frag_row;
                af[i] = *reinterpret_cast<const half8_t*>(wl + row * Cfg::ROW_BYTES + ((chunk ^ swzk<BK>(row)) << 4));
            }
#pragma unroll
            for (int j = 0; j < Cfg::WN; ++j) {
                const int row = (wave_n * Cfg::WN + j) * 32 + frag_row;
                bf[j] = *reinterpret_cast<const half8_t*>(xl + row * Cfg::ROW_BYTES + ((chunk ^ swzk<BK>(row)) << 4));
            }
            if (PROLOGUE) {
                const half8_t sc = *reinterpret_cast<const half8_t*>(pro_lds + c0 + chunk * 8);
                const half8_t sh = *reinterpret_cast<const half8_t*>(pro_lds + 2048 + c0 + chunk * 8);
                const half8_t z = {};
#pragma unroll
                for (int j = 0; j < Cfg::WN; ++j) bf[j] = __builtin_elementwise_max(bf[j] * sc + sh, z);
            }
#ifdef METRO_SETPRIO
            __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
            for (int i = 0; i < Cfg::WM; ++i)
#pragma unroll
                for (int j = 0; j < Cfg::WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
#ifdef METRO_SETPRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            if constexpr (decltype(with_issue)::value) issue_part(kk);
        }
    };
    using Yes = std::integral_constant<bool, true>;
    using No = std::integral_constant<bool, false>;

    // ---- residual prefetch -------------------------------------------------------------------
    // The shortcut tile is loaded into registers in the layout of the row-wise epilogue (16 bytes
    // per lane), issued right AFTER the last LDS-DMA of the block (start of the drain phase): VMEM
    // returns in order, so loads issued earlier would hold up the ring's first stages behind their
    // HBM latency.  Being younger than every DMA, they stay in flight across the drain steps: the
    // drain waits count them in (vmcnt(N + EPI_ITERS)).
    constexpr int CPRO = Cfg::TM / 8;                 // 16-byte chunks per tile row
    constexpr int EPI_ITERS = Cfg::TN * CPRO / Cfg::NT;
    static_assert(Cfg::TN * CPRO % Cfg::NT == 0, "epilogue loop must divide evenly");
    const bool res_same = a.res_stride == 1 && a.res_offset == 0 && a.res_h == a.h_out && a.res_w == a.w_out;
    const bool has_res = residual != nullptr && !out_f32;
    uint4 rres[EPI_ITERS];
#pragma unroll
    for (int it = 0; it < EPI_ITERS; ++it) rres[it] = make_uint4(0, 0, 0, 0);
    auto prefetch_residual = [&]() {
#pragma unroll
        for (int it = 0; it < EPI_ITERS; ++it) {
            const int idx = tid + it * Cfg::NT;
            const int prow = idx / CPRO;
            const int ch = idx - prow * CPRO;
            const int m = m0 + prow;
            const int co = n0 + ch * 8;
            // every lane issues exactly one load per iteration (clamped address) so that the
            // per-wave VMEM count the drain waits rely on is uniform
            const bool ok = m < a.m_total && co + 8 <= a.c_out;
            size_t rp = ok ? (size_t)m : 0;          // same-geometry shortcut: residual pixel == output pixel
            if (!res_same && ok) {
                const int img = m / hw_out;
                const int rem = m - img * hw_out;
                const int ho = rem / a.w_out;
                const int wo = rem - ho * a.w_out;
                rp = (size_t)(img * a.res_h + ho * a.res_stride + a.res_offset) * a.res_w +
                     (wo * a.res_stride + a.res_offset);
            }
            rres[it] = *reinterpret_cast<const uint4*>(residual + rp * a.c_out + (ok ? co : 0));
        }
    };

    // ---- pre-activation BN parameters (scale | shift behind the ring) by LDS-DMA, BEFORE the ring's first requests: the oldest
    // requests of their waves, so the counted wait of the first K step covers them and no __syncthreads (= vmcnt(0): every prologue
    // stage landed, not just the first) stands between the prologue and the first MFMA.  1 KiB pieces of 512 channels; lanes past
    // c_in take the zero page like the operand DMAs: the non-FASTK tail (c_in % BK != 0) DOES read scale / shift of channels
    // c_in .. cpad, and 0 x (zero-filled operand) + 0 must stay 0 whatever channels 0-7 hold (a non-finite scale there would turn
    // the padding into NaN).  (The fused-next-conv form keeps its ordinary loads + barrier.)
    constexpr bool PRO_DMA = PROLOGUE && !FUSE2;
    if constexpr (PRO_DMA) {
#pragma unroll
        for (int i = 0; i < (8 + NW - 1) / NW; ++i) {
            const int id = wave * ((8 + NW - 1) / NW) + i;         // 0 ... 7: vector id >> 2 (scale, shift), piece id & 3
            const int piece = id & 3;
            if (id < 8 && piece * 512 < a.c_in) {
                const int idx = piece * 512 + lane * 8;
                const half_t* src = idx < a.c_in ? ((id >> 2) ? pro_shift : pro_scale) + idx : zero;
                dma16(src, __builtin_amdgcn_readfirstlane(smem_base + Cfg::MAIN_BYTES + (id >> 2) * 4096 + piece * 1024));
            }
        }
    }
    // ---- main loop: STAGES-1 steps in flight -------------------------------------------------
    if constexpr (STAGES > 1) {
#pragma unroll
        for (int s = 0; s < STAGES - 1; ++s)
            if (s < nk) issue_step(s);
    } else {
        issue_step(0);
    }
    // ---- pre-activation BN parameters into LDS (behind the ring).  Done AFTER the first ring stages
    // are in flight so that the latency of these ordinary loads overlaps the first DMAs (a block that
    // waits for them first pays a second, serial memory round trip: +30 us on block1's shortcut).
    if constexpr (FUSE2) {
        // W2 [64][TM] as TM/64 swizzled stage images of 64 rows x 64 k: issued before the residual
        // prefetch, so the counted wait of the K step covers it; next unit's pre-activation scale/shift
        // by ordinary loads (visible after the ring barrier)
        static_assert(BK == 64 && (F2_W_BYTES / 1024) % NW == 0, "FUSE2 layout");
#pragma unroll
        for (int i = 0; i < F2_W_BYTES / 1024 / NW; ++i) {
            const int vrow = (i * NW + wave) * 8 + (lane >> 3);          // row of the [TM/64 * 64][64] image
            const int kc = vrow >> 6, r = vrow & 63;
            const half_t* src = f2.w2 + (size_t)r * Cfg::TM + kc * 64 + (((lane & 7) ^ swzk<64>(r)) * 8);
            dma16(src, __builtin_amdgcn_readfirstlane(smem_base + F2_W_OFF + (i * NW + wave) * 1024));
        }
        half_t* p2 = reinterpret_cast<half_t*>(smem + F2_P_OFF);
        for (int c = tid * 8; c < Cfg::TM; c += Cfg::NT * 8) {
            *reinterpret_cast<uint4*>(p2 + c) = *reinterpret_cast<const uint4*>(f2.scale2 + c);
            *reinterpret_cast<uint4*>(p2 + Cfg::TM + c) = *reinterpret_cast<const uint4*>(f2.shift2 + c);
        }
    }
    if (PROLOGUE && !PRO_DMA) {
        const int cpad = kc_steps * BK;
        for (int c = tid * 8; c < cpad; c += Cfg::NT * 8) {
            uint4 sv = make_uint4(0, 0, 0, 0), bv = make_uint4(0, 0, 0, 0);
            if (c < a.c_in) {
                sv = *reinterpret_cast<const uint4*>(pro_scale + c);
                bv = *reinterpret_cast<const uint4*>(pro_shift + c);
            }
            *reinterpret_cast<uint4*>(pro_lds + c) = sv;
            *reinterpret_cast<uint4*>(pro_lds + 2048 + c) = bv;
        }
    }

    if (PROLOGUE && !PRO_DMA) __syncthreads();   // pro_lds written
    if constexpr (STAGES == 1) {
        // single slot (short-K layers: small LDS footprint -> several blocks per CU overlap instead)
        int cc0 = 0;
        for (int k = 0; k < nk; ++k) {
            if (k > 0) {
                wait_vm_and_barrier<0>();             // WAR: everyone is done reading the slot
                issue_step(0);
            }
            if (has_res && k == nk - 1) {
                prefetch_residual();
                wait_vm_and_barrier<EPI_ITERS>();     // the step has landed; the residual may still fly
            } else {
                wait_vm_and_barrier<0>();             // RAW: the step has landed for every wave
            }
            compute_step(0, cc0, No{});
            cc0 += BK;
            if (cc0 >= a.c_in) cc0 = 0;
        }
    } else {
        int cbuf = 0, ibuf = STAGES - 1, cc0 = 0;
        auto advance = [&]() {
            cbuf = cbuf + 1 == STAGES ? 0 : cbuf + 1;
            ibuf = ibuf + 1 == STAGES ? 0 : ibuf + 1;
            cc0 += BK;
            if (cc0 >= a.c_in) cc0 = 0;
        };
        // steady state: every step has STAGES-2 younger steps in flight and issues one more
        const int n_main = nk - (STAGES - 1);
        for (int k = 0; k < n_main; ++k) {
            wait_vm_and_barrier<(STAGES - 2) * Cfg::LPS>();
            issue_begin(ibuf);
            compute_step(cbuf, cc0, Yes{});
            issue_end();
            advance();
        }
        // drain: nothing left to issue, the in-flight count runs down (residual loads ride along)
        if (has_res) prefetch_residual();
        for (int k = n_main < 0 ? 0 : n_main; k < nk; ++k) {
            int ahead = nk - 1 - k;                      // younger steps still allowed in flight
            if (ahead > STAGES - 2) ahead = STAGES - 2;
            if (has_res) {
                switch (ahead) {
                    case 4: wait_vm_and_barrier<(STAGES >= 6 ? 4 : 0) * Cfg::LPS + EPI_ITERS>(); break;
                    case 3: wait_vm_and_barrier<(STAGES >= 5 ? 3 : 0) * Cfg::LPS + EPI_ITERS>(); break;
                    case 2: wait_vm_and_barrier<(STAGES >= 4 ? 2 : 0) * Cfg::LPS + EPI_ITERS>(); break;
                    case 1: wait_vm_and_barrier<(STAGES >= 3 ? 1 : 0) * Cfg::LPS + EPI_ITERS>(); break;
                    default: wait_vm_and_barrier<EPI_ITERS>(); break;
                }
            } else {
                switch (ahead) {
                    case 4: wait_vm_and_barrier<(STAGES >= 6 ? 4 : 0) * Cfg::LPS>(); break;
                    case 3: wait_vm_and_barrier<(STAGES >= 5 ? 3 : 0) * Cfg::LPS>(); break;
                    case 2: wait_vm_and_barrier<(STAGES >= 4 ? 2 : 0) * Cfg::LPS>(); break;
                    case 1: wait_vm_and_barrier<(STAGES >= 3 ? 1 : 0) * Cfg::LPS>(); break;
                    default: wait_vm_and_barrier<0>(); break;
                }
            }
            compute_step(cbuf, cc0, No{});
            advance();
        }
    }

    // ---- epilogue -----------------------------------------------------------------------
    // fused pair (ConvSplit): this block's cout tile belongs to exactly one of the two outputs
    const bool second = a.split > 0 && n0 >= a.split;
    const int o_c = a.split > 0 ? (second ? a.c_out2 : a.split) : a.c_out;   // channels of the target tensor
    const int o_n0 = second ? n0 - a.split : n0;                              // tile offset inside it
    const int o_relu = second ? a.relu2 : a.relu;
    void* o_ptr = second ? out2 : out;
    if (out_f32) {
        // fp32 output (logits): direct stores, 16 bytes per lane
#pragma unroll
        for (int j = 0; j < Cfg::WN; ++j) {
            const int m = m0 + (wave_n * Cfg::WN + j) * 32 + frag_row;
            if (m >= a.m_total) continue;
#pragma unroll
            for (int i = 0; i < Cfg::WM; ++i) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int co = n0 + (wave_m * Cfg::WM + i) * 32 + 8 * q + 4 * frag_half;
                    if (co >= a.c_out) continue;
                    const floatx4 bv = *reinterpret_cast<const floatx4*>(bias + co);
                    floatx4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e] + bv[e];
                    if (a.relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    *reinterpret_cast<floatx4*>(reinterpret_cast<float*>(out) + (size_t)m * a.c_out + co) = v;
                }
            }
        }
        return;
    }

    __syncthreads();   // every wave is done reading the ring
    // 1) accumulators (+bias, ReLU) -> LDS tile [pixel][cout] fp16
#pragma unroll
    for (int i = 0; i < Cfg::WM; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = (wave_m * Cfg::WM + i) * 32 + 8 * q + 4 * frag_half;   // within the tile
            const int co = n0 + col;
            floatx4 bv = {0.f, 0.f, 0.f, 0.f};
            if (co < a.c_out) bv = *reinterpret_cast<const floatx4*>(bias + co);
#pragma unroll
            for (int j = 0; j < Cfg::WN; ++j) {
                const int prow = (wave_n * Cfg::WN + j) * 32 + frag_row;
                half4_t hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[i][j][4 * q + e] + bv[e];
                    if (o_relu) v = fmaxf(v, 0.f);
                    hv[e] = (half_t)v;
                }
                *reinterpret_cast<half4_t*>(smem + prow * Cfg::OUT_ROW_BYTES + col * 2) = hv;
            }
        }
    }
    constexpr int T2M = 2, T2N = Cfg::TN / 32;              // 64 conv1 outputs x TN pixels in 32x32 MFMA tiles
    constexpr int F2_KSTEPS = FUSE2 ? Cfg::TM / 16 : 1;
    __syncthreads();
    // 2) row-wise: 16 bytes per lane, (+ prefetched residual), full-line stores
    half_t* outh = reinterpret_cast<half_t*>(o_ptr);
    // A tile that lies inside the tensor (block-uniform test: every tile of the production shapes but ragged last ones) takes the
    // guard-free form: the chunk reads of up to eight iterations back to back, then the sums and stores.  The guarded loop below
    // is an exec-mask branch per iteration, across which hipcc moves no LDS read: read - wait - store, EPI_ITERS times in a row
    // (8-16 exposed LDS round trips per tile: ~1.2 us of a 15 us one-tile-per-CU launch at batch 64).
    constexpr int EPI_G = EPI_ITERS % 8 == 0 ? 8 : EPI_ITERS % 4 == 0 ? 4 : EPI_ITERS % 2 == 0 ? 2 : 1;
    const bool interior = m0 + Cfg::TN <= a.m_total && o_n0 + Cfg::TM <= o_c;
    if (interior) {
#pragma unroll
        for (int it0 = 0; it0 < EPI_ITERS; it0 += EPI_G) {
            uint4 vv[EPI_G];
#pragma unroll
            for (int u = 0; u < EPI_G; ++u) {
                const int idx = tid + (it0 + u) * Cfg::NT;
                const int prow = idx / CPRO;
                vv[u] = *reinterpret_cast<const uint4*>(smem + prow * Cfg::OUT_ROW_BYTES + (idx - prow * CPRO) * 16);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < EPI_G; ++u) {
                const int idx = tid + (it0 + u) * Cfg::NT;
                const int prow = idx / CPRO;
                const int ch = idx - prow * CPRO;
                uint4 v = vv[u];
                if (residual != nullptr) {
                    half2_t* x = reinterpret_cast<half2_t*>(&v);
                    const half2_t* r = reinterpret_cast<const half2_t*>(&rres[it0 + u]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] = x[e] + r[e];    // fp16 Add, like the reference graph
                }
                store_out16<2>(outh + (size_t)(m0 + prow) * o_c + o_n0 + ch * 8, v);
                if constexpr (FUSE2) *reinterpret_cast<uint4*>(smem + prow * Cfg::OUT_ROW_BYTES + ch * 16) = v;
            }
        }
    } else
#pragma unroll
    for (int it = 0; it < EPI_ITERS; ++it) {
        const int idx = tid + it * Cfg::NT;
        const int prow = idx / CPRO;
        const int ch = idx - prow * CPRO;
        const int m = m0 + prow;
        const int co = o_n0 + ch * 8;                // channel inside the target tensor
        if (m >= a.m_total || co >= o_c) continue;
        uint4 v = *reinterpret_cast<const uint4*>(smem + prow * Cfg::OUT_ROW_BYTES + ch * 16);
        if (co + 8 <= o_c) {
            if (residual != nullptr) {
                half2_t* x = reinterpret_cast<half2_t*>(&v);
                const half2_t* r = reinterpret_cast<const half2_t*>(&rres[it]);
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = x[e] + r[e];    // fp16 Add, like the reference graph
            }
            store_out16<2>(outh + (size_t)m * o_c + co, v);
            if constexpr (FUSE2) *reinterpret_cast<uint4*>(smem + prow * Cfg::OUT_ROW_BYTES + ch * 16) = v;
        } else {
            // ragged channel tail (c_out % 8 != 0 never carries a residual: see conv_f16_dma_supported)
            const half8_t x = *reinterpret_cast<const half8_t*>(&v);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (co + e < o_c) outh[(size_t)m * o_c + co + e] = x[e];
        }
    }
    if constexpr (FUSE2) {
        // ---- second GEMM: t1 = relu(W2 * relu(x_out*scale2 + shift2) + bias2) from the LDS-resident tile ----
        __syncthreads();                      // final x_out rows are back in LDS
        if (wave < T2M * T2N) {
            const int i2 = wave / T2N, j2 = wave % T2N;
            floatx16 acc2;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[e] = 0.f;
            const half_t* p2 = reinterpret_cast<const half_t*>(smem + F2_P_OFF);
            const int prow2 = j2 * 32 + frag_row;
            const int arow = i2 * 32 + frag_row;
            const char* w2l = smem + F2_W_OFF + arow * 128;
#pragma unroll
            for (int ks = 0; ks < F2_KSTEPS; ++ks) {
                const int k0 = ks * 16 + frag_half * 8;
                const half8_t af = *reinterpret_cast<const half8_t*>(
                    w2l + (ks >> 2) * 8192 + ((((ks & 3) * 2 + frag_half) ^ swzk<64>(arow)) << 4));
                half8_t bf = *reinterpret_cast<const half8_t*>(smem + prow2 * Cfg::OUT_ROW_BYTES + k0 * 2);
                const half8_t sc = *reinterpret_cast<const half8_t*>(p2 + k0);
                const half8_t sh = *reinterpret_cast<const half8_t*>(p2 + Cfg::TM + k0);
                const half8_t z = {};
                bf = __builtin_elementwise_max(bf * sc + sh, z);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc2, 0, 0, 0);
            }
            const int m = m0 + prow2;
            if (m < a.m_total) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int co = i2 * 32 + 8 * q + 4 * frag_half;
                    const floatx4 bv = *reinterpret_cast<const floatx4*>(f2.bias2 + co);
                    half4_t hv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) hv[e] = (half_t)fmaxf(acc2[4 * q + e] + bv[e], 0.f);
                    *reinterpret_cast<half4_t*>(f2.out2 + (size_t)m * f2.c2 + co) = hv;
                }
            }
        }
    }
}

template <class Cfg, bool PROLOGUE, bool FASTK>
__global__ __launch_bounds__(Cfg::NT) void conv_igemm_f16_dma_kernel(
    ConvArgs a, const half_t* __restrict__ in, const half_t* __restrict__ w,
    const float* __restrict__ bias, const half_t* __restrict__ pro_scale,
    const half_t* __restrict__ pro_shift, const half_t* __restrict__ residual,
    void* __restrict__ out, int out_f32, int tiles_m, void* __restrict__ out2) {
    conv_dma_body<Cfg, PROLOGUE, FASTK, false>(a, in, w, bias, pro_scale, pro_shift, residual, out, out_f32, tiles_m,
                                               out2, Fuse2Args{});
}

// conv3 + next conv1 (block1)
template <class Cfg>
__global__ __launch_bounds__(Cfg::NT) void conv_igemm_f16_fuse2_kernel(
    ConvArgs a, const half_t* __restrict__ in, const half_t* __restrict__ w, const float* __restrict__ bias,
    const half_t* __restrict__ residual, void* __restrict__ out, Fuse2Args f2) {
    conv_dma_body<Cfg, false, true, true>(a, in, w, bias, nullptr, nullptr, residual, out, 0, 1, nullptr, f2);
}

static int env_int(const char* name, int dflt) { return tuning_knob(name, dflt); }

static thread_local void* g_out2 = nullptr;   // second output of a fused pair (set by launch_conv_f16_dma)

template <class Cfg, bool PROLOGUE, bool FASTK>
static int launch_dma_cfg2(const ConvArgs& a, const half_t* in, const half_t* w, const float* bias,
                           const half_t* ps, const half_t* pb, const half_t* res, void* out, int out_f32,
                           hipStream_t stream) {
    if (a.split > 0 && (a.split % Cfg::TM) != 0) {
        set_error("conv_igemm_f16_dma: split %d is not a multiple of the %d-wide cout tile", a.split, Cfg::TM);
        return METRO_ERR_INVALID_ARG;
    }
    if (note_kernel("conv_igemm_f16_dma<%dx%d,bk%d,s%d%s%s>%s%s%s", Cfg::TM, Cfg::TN, Cfg::BK, Cfg::STAGES, PROLOGUE ? ",pro" : "",
                    FASTK ? "" : ",ktail", res ? "+res" : "", a.split > 0 ? "+pair" : "", out_f32 ? "+f32out" : ""))
        return METRO_OK;
    auto kern = conv_igemm_f16_dma_kernel<Cfg, PROLOGUE, FASTK>;
    constexpr int lds = Cfg::MAIN_BYTES + (PROLOGUE ? Cfg::PRO_BYTES : 0);
    static PerDeviceInt attr_done;
    if (const int st = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds, attr_done, "conv_igemm_f16_dma")) return st;
    const int tiles_m = (a.c_out + Cfg::TM - 1) / Cfg::TM;
    const int tiles_n = (a.m_total + Cfg::TN - 1) / Cfg::TN;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(Cfg::NT), lds, stream, a, in, w, bias, ps, pb,
                       res, out, out_f32, tiles_m, g_out2);
    return launch_status("conv_igemm_f16_dma");
}

template <class Cfg, bool PROLOGUE>
static int launch_dma_cfg(const ConvArgs& a, const half_t* in, const half_t* w, const float* bias,
                          const half_t* ps, const half_t* pb, const half_t* res, void* out, int out_f32,
                          hipStream_t stream) {
    // FASTK needs whole K steps per tap and 16-byte-aligned weight rows for the running pointers
    if (a.c_in % Cfg::BK == 0)
        return launch_dma_cfg2<Cfg, PROLOGUE, true>(a, in, w, bias, ps, pb, res, out, out_f32, stream);
    return launch_dma_cfg2<Cfg, PROLOGUE, false>(a, in, w, bias, ps, pb, res, out, out_f32, stream);
}

//                        WAVES_M WAVES_N WM WN STAGES BK   tile (cout x pixels), waves, LDS
using Dma128x256s3 = DmaCfg<2, 4, 2, 2, 3>;        // 128 x 256, 8 waves, 144 KiB: deep-K, >= 256 tiles
using Dma128x128s4 = DmaCfg<2, 4, 2, 1, 4>;        // 128 x 128, 8 waves, 128 KiB: deep-K, few tiles
using Dma128x128s2 = DmaCfg<2, 4, 2, 1, 2>;        // 128 x 128, 8 waves,  64 KiB: 2 blocks / CU
using Dma128x128s1 = DmaCfg<2, 4, 2, 1, 1>;        // 128 x 128, 8 waves,  34 KiB: 3 blocks / CU (K <= 64)
using Dma256x256s4k32 = DmaCfg<2, 4, 4, 2, 4, 32>; // same tile, BK 32, 4 stages
using Dma128x256s3k32 = DmaCfg<2, 4, 2, 2, 3, 32>; // 128 x 256, BK 32, 72 KiB: 2 blocks / CU (epilogue of one overlaps the loop of the other)
using Dma64x128s3 = DmaCfg<1, 4, 2, 1, 3>;         //  64 x 128, 4 waves,  72 KiB
using Dma64x128s1 = DmaCfg<1, 4, 2, 1, 1>;         //  64 x 128, 4 waves,  24 KiB
using Dma64x128s3k32 = DmaCfg<1, 4, 2, 1, 3, 32>;  //  64 x 128, 4 waves, BK 32 (the stem's 32-wide taps)

using DmaFuse256x64 = DmaCfg<4, 2, 2, 1, 1>;       // 256 cout x 64 px, 8 waves (wave 64 x 32), one K step: conv3 + next conv1

bool conv_f16_fuse2_supported(const MetroConvDesc& d, int c2) {
    static const int enabled = env_int("METRO_FUSE2", 1);
    return enabled && d.c_out == DmaFuse256x64::TM && c2 == 64 && d.kh == 1 && d.kw == 1 && d.c_in == 64 &&
           d.in_pix_stride == 64 && d.stride == 1 && d.pad_top == 0 && d.pad_left == 0 && !d.has_prologue &&
           d.out_dtype == METRO_F16 && d.in_dtype == METRO_F16 && d.h_in == d.h_out && d.w_in == d.w_out;
}

static int launch_fuse2(const ConvArgs& a, const half_t* in, const half_t* w, const float* bias, const half_t* res,
                        void* out, const ConvFuse2& f, hipStream_t stream) {
    using Cfg = DmaFuse256x64;
    if (note_kernel("conv_igemm_f16_fuse2<256x64>%s", res ? "+res" : "")) return METRO_OK;
    auto kern = conv_igemm_f16_fuse2_kernel<Cfg>;
    constexpr int lds = Cfg::MAIN_BYTES + 64 * Cfg::TM * 2 + 2 * Cfg::TM * 2;
    static PerDeviceInt attr_done;
    if (const int st = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds, attr_done, "conv_igemm_f16_fuse2")) return st;
    const int tiles_n = (a.m_total + Cfg::TN - 1) / Cfg::TN;
    hipLaunchKernelGGL(kern, dim3(tiles_n), dim3(Cfg::NT), lds, stream, a, in, w, bias, res, out,
                       Fuse2Args{static_cast<const half_t*>(f.w2), f.bias2, static_cast<const half_t*>(f.scale2),
                                 static_cast<const half_t*>(f.shift2), static_cast<half_t*>(f.out2), f.c2});
    return launch_status("conv_igemm_f16_dma<fuse2>");
}

bool conv_f16_dma_supported(const MetroConvDesc& d) {
    // in_pix_stride % 4: the 4-channel bordered stem image gives 8-byte-aligned 16-byte sources
    return d.in_pix_stride % 4 == 0 && d.c_in % 8 == 0 && d.c_in <= 2048 && d.c_out % 4 == 0 &&
           (!d.has_residual || d.c_out % 8 == 0) && (d.in_pix_stride % 8 == 0 || !d.has_prologue);
}

int launch_conv_f16_dma(const MetroConvDesc& d, const void* in_, const void* w_, const float* bias,
                        const void* ps_, const void* pb_, const void* res_, void* out, hipStream_t stream,
                        const ConvSplit* split, const ConvFuse2* fuse2, const ConvProjSc* psc, const ConvRebuild* rebuild) {
    ConvArgs a = make_conv_args(d);
    g_out2 = nullptr;
    if (psc != nullptr && psc->x != nullptr) {        // projection shortcut computed in the launch: the persistent kernel only
        const bool rb = rebuild != nullptr && rebuild->t2_prev != nullptr;
        if (fuse2 != nullptr && fuse2->w2 != nullptr && fuse2->c2 == 64 && conv_pw64_supported(d, rb ? 4 : 3))
            return launch_conv_pw64(d, in_, w_, bias, nullptr, nullptr, nullptr, out, stream, nullptr, fuse2, psc, rebuild);
        set_error("conv3 with an in-launch projection shortcut: built for 1x1 stride-1 64 -> 256 + next conv1 (block1/unit_1) only");
        return METRO_ERR_UNSUPPORTED;
    }
    if (!(fuse2 != nullptr && fuse2->w2 != nullptr) && !(split != nullptr && split->split > 0) && conv_pws_supported(d))
        return launch_conv_pws(d, in_, w_, bias, res_, out, stream);
    {
        const int mode = (fuse2 != nullptr && fuse2->w2 != nullptr) ? 2 : (split != nullptr && split->split > 0) ? 1 : 0;
        if (conv_pw64_supported(d, mode) && (mode != 1 || (split->relu2 == 1 && ((split->split == 256 && split->c_out2 == 64) || (split->split == 512 && split->c_out2 == 128)))) &&
            (mode != 2 || fuse2->c2 == (d.c_in == 128 ? 128 : 64)))
            return launch_conv_pw64(d, in_, w_, bias, ps_, pb_, res_, out, stream, split, fuse2);
    }
    // deep-K pre-activated 1x1 layers with at least one 256 x 256 tile per CU: the four-wave GEMM (128 x 128 wave tiles,
    // register-staged operands, the pre-activation applied once per element on its way into LDS).  Two other forms of this GEMM
    // (8-phase two-wave-group LDS-DMA, round 2; four waves with both operands by LDS-DMA, round 4) give the same bits and
    // measured the same inside the forward: they live in csrc/experimental/ (libmetro_experimental.so), not in the product.
    if (!(fuse2 != nullptr && fuse2->w2 != nullptr) && conv_gemm4w_supported(d, split))
        return launch_conv_gemm4w(d, in_, w_, bias, ps_, pb_, res_, out, stream, split);
    if (fuse2 != nullptr && fuse2->w2 != nullptr) {
        if (!conv_f16_fuse2_supported(d, fuse2->c2)) { set_error("conv fuse2: unsupported layer shape (c_out %d c2 %d k %dx%d c_in %d pix_stride %d stride %d pad %d,%d pro %d dt %d/%d hw %dx%d -> %dx%d)",
                                                                 d.c_out, fuse2->c2, d.kh, d.kw, d.c_in, d.in_pix_stride, d.stride, d.pad_top, d.pad_left, d.has_prologue,
                                                                 d.in_dtype, d.out_dtype, d.h_in, d.w_in, d.h_out, d.w_out); return METRO_ERR_INVALID_ARG; }
        return launch_fuse2(a, static_cast<const half_t*>(in_), static_cast<const half_t*>(w_), bias,
                            d.has_residual ? static_cast<const half_t*>(res_) : nullptr, out, *fuse2, stream);
    }
    if (split != nullptr && split->split > 0) {
        if (d.has_residual || d.out_dtype != METRO_F16 || split->split + split->c_out2 != d.c_out ||
            split->split % 256 != 0 || split->c_out2 % 8 != 0) {
            set_error("conv_igemm_f16_dma: unsupported fused pair (split %d + %d vs c_out %d)", split->split,
                      split->c_out2, d.c_out);
            return METRO_ERR_INVALID_ARG;
        }
        a.split = split->split; a.c_out2 = split->c_out2; a.relu2 = split->relu2;
        g_out2 = split->out2;
    }
    const half_t* in = static_cast<const half_t*>(in_);
    const half_t* w = static_cast<const half_t*>(w_);
    const half_t* ps = static_cast<const half_t*>(ps_);
    const half_t* pb = static_cast<const half_t*>(pb_);
    const half_t* res = d.has_residual ? static_cast<const half_t*>(res_) : nullptr;
    const int out_f32 = d.out_dtype == METRO_F32;
    const bool pro = d.has_prologue != 0;
    const int tiles128 = (d.c_out + 127) / 128;
    // tuning knobs (A/B runs): K-steps up to which the 1-stage / 2-stage 128x128 configs are used
    static const int nk_s1 = env_int("METRO_NK_S1", 2);
    static const int nk_s2 = env_int("METRO_NK_S2", 8);
#define METRO_DMA(CFG)                                                                         \
    return pro ? launch_dma_cfg<CFG, true>(a, in, w, bias, ps, pb, res, out, out_f32, stream)  \
               : launch_dma_cfg<CFG, false>(a, in, w, bias, ps, pb, res, out, out_f32, stream)
    if (d.c_in <= 32 && !pro) {
        return launch_dma_cfg<Dma64x128s3k32, false>(a, in, w, bias, ps, pb, res, out, out_f32, stream);
    }
    const int nk = d.kh * d.kw * ((d.c_in + 63) / 64);
    // 64-cout tiles: narrow layers, and heads whose width wastes most of a second 128-tile (136 = 8*17)
    static const int head64 = env_int("METRO_DMA_HEAD64", 1);
    const bool narrow = d.c_out <= 64 || (head64 && d.c_out < 256 && d.c_out % 128 != 0 && (d.c_out % 128) <= 64 &&
                                          d.c_out / 128 <= 1);
    if (narrow) {
        if (nk <= nk_s1) { METRO_DMA(Dma64x128s1); }
        METRO_DMA(Dma64x128s3);
    }
    if (nk <= nk_s1) { METRO_DMA(Dma128x128s1); }
    static const int k32 = env_int("METRO_DMA_K32", 8);
    {
        const long b256 = (long)tiles128 * ((a.m_total + 255) / 256);
        if (k32 && nk <= k32 && b256 >= 512) { METRO_DMA(Dma128x256s3k32); }
    }
    if (nk <= nk_s2) { METRO_DMA(Dma128x128s2); }
    // 256-pixel tiles only when they still give every CU a block
    const long blocks256 = (long)tiles128 * ((a.m_total + 255) / 256);
    static const int big = env_int("METRO_DMA_BIG", 2);
    if (big && d.c_out % 256 == 0 && blocks256 / 2 >= 256) { METRO_DMA(Dma256x256s4k32); }
    if (blocks256 >= 256) { METRO_DMA(Dma128x256s3); }
    // deep-K layers whose 128 x 128 tiles leave CUs idle (block2/unit_4's strided 3x3 at batch 64: 128 tiles on 256 CUs): 64-cout
    // tiles double the blocks (same pixel gather per block, half the weight rows and MFMAs per K step); same bits
    static const int half_tiles = env_int("METRO_DMA_HALF_TILES", 1);
    const long blocks128 = (long)tiles128 * ((a.m_total + 127) / 128);
    if (half_tiles && d.c_out % 64 == 0 && blocks128 >= 64 && blocks128 < 224 && a.split == 0) { METRO_DMA(Dma64x128s3); }
    METRO_DMA(Dma128x128s4);
#undef METRO_DMA
}

// fp16 convolution dispatcher (metro_conv_f16 and every plain conv layer of the plan)
int launch_conv_f16(const MetroConvDesc& d, const void* in, const void* w, const float* bias, const void* ps,
                    const void* pb, const void* res, void* out, hipStream_t stream) {
    if (conv3x3_c64_supported(d)) return launch_conv3x3_c64(d, in, w, bias, out, stream);
    if (conv3x3_slab_supported(d)) return launch_conv3x3_slab(d, in, w, bias, out, stream);
    if (conv_f16_dma_supported(d)) return launch_conv_f16_dma(d, in, w, bias, ps, pb, res, out, stream);
    set_error("conv_f16: unsupported layer (c_in %d must be a multiple of 8 and <= 2048, in_pix_stride %d of 4 (8 with a "
              "prologue), c_out %d of 4 (8 with a residual))", d.c_in, d.in_pix_stride, d.c_out);
    return METRO_ERR_UNSUPPORTED;
}

}  // namespace metro
