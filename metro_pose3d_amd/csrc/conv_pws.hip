// conv3 + shortcut add of blocks 3-4 (1x1, K = 256 / 512 -> 4 K, + residual) as a persistent weight-stationary kernel whose two
// waves per SIMD run HALF A TILE APART: conv_pw64.hip's <k256|k512, wm8, res> with its phases overlapped.
//
// Why (round 5).  conv_pw64_kernel<512, 8, false, true, 0> runs the phases of a 32-pixel tile in lock step on its eight waves: all
// of them multiply (32 MFMAs per wave), all of them write accumulators to the shared [pixel][256 channel] LDS tile, barrier, all of
// them do the row-wise pass (shortcut add, store).  The matrix pipe idles during everything that is not the GEMM: block4's conv3
// runs 137 GFLOP in 177 us (0.77 PFLOP/s, 37 % of the pipe at the clock it holds) while moving 606 MB at 3.4 TB/s -- bound by
// neither.  The same diagnosis as block1's launches (conv_b1.hip), a different cure, because here the weights fill the register
// file of ALL eight waves (K / 4 VGPRs each) and none can be spared for another role:
//   * a wave owns 32 output channels of the block's 256-channel slab from the operand tile to the store: its accumulators go
//     through a WAVE-PRIVATE LDS tile [32 pixels][32 channels] (write in MFMA layout, read back as 16-byte row pieces), the
//     shortcut arrives by LDS-DMA in exactly that row-piece layout (every lane reads back the 16 bytes it requested), and the
//     sum leaves as 64-byte pieces of NHWC rows.  No other wave is involved: no barrier inside the epilogue;
//   * every wave executes two barriers per tile -- in front of its GEMM and in front of its epilogue -- and waves 4-7 execute ONE
//     extra barrier before their loop: the two waves of a SIMD are permanently one phase apart, one feeds the matrix pipe while
//     the other converts, adds and stores (the schedule of round 2's conv_gemm8p);
//   * operand tiles (32 pixels x K channels, 128-byte rows, chunk-swizzled) come by LDS-DMA into a ring of three, requested two
//     tiles ahead; per tile a wave issues [shortcut rows of tile j | its share of operand tile j + 2 | ... | stores of tile j] and
//     waits ONCE, with a counted vmcnt that leaves the operand requests in flight.
// Arithmetic is conv_pw64's (fp32 accumulation in ascending k on one accumulator, fp16(conv + bias), then the fp16 shortcut add:
// reference resnet_v2.py:134-138): the same bits.
#include "metro_common.h"

namespace metro {

typedef _Float16 half_t;
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

struct PwsArgs {
    const half_t* in;          // [m_total][K]
    const half_t* w;           // [c_out][K]
    const float* bias;         // [c_out]
    const half_t* residual;    // [m_total][c_out]
    half_t* out;               // [m_total][c_out]
    int m_total, n_tiles, c_out;
};

namespace pws {
constexpr int NW = 8, NT = 512, TN = 32, CB = 256, NBUF = 3;
constexpr int T_ROW = 80;                       // padded rows of a wave's private [32 pixels][32 channels] tile (64 B + 16)
constexpr int T_BYTES = TN * T_ROW;             // 2560
constexpr int R_BYTES = TN * 64;                // the wave's shortcut rows: 32 pixels x 64 B
template <int K>
struct Lay {
    static constexpr int KS = K / 64;                         // operand tile = KS slices [32 rows][64 k] (128-byte rows, swizzled)
    static constexpr int SL_BYTES = TN * 128;                 // 4 KiB
    static constexpr int X_BYTES = KS * SL_BYTES;
    static constexpr int NX = X_BYTES / 1024 / NW;            // operand LDS-DMA instructions per wave and tile
    static constexpr int X_OFF = 0;                           // NBUF tiles
    static constexpr int T_OFF = NBUF * X_BYTES;              // per wave: private transposition tile | shortcut rows
    static constexpr int BIAS_OFF = T_OFF + NW * (T_BYTES + R_BYTES);
    static constexpr int LDS = BIAS_OFF + CB * 4;
    static_assert(X_BYTES % (1024 * NW) == 0, "operand tile must split evenly over the waves");
};
}  // namespace pws

__device__ __forceinline__ int pws_swz(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ void pws_dma16(const void* gsrc, unsigned lds_addr) {
    asm volatile(
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, off"
        :
        : "v"(gsrc), "s"(lds_addr));
}
template <int N>
__device__ __forceinline__ void pws_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void pws_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// s_waitcnt lgkmcnt(n), n a compile-time value after unrolling (0 .. 6), tied to the register the wait is for
__device__ __forceinline__ void pws_wait_lgkm_dyn(half8_t& r, int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r)); break;
        case 1: asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(r)); break;
        case 2: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(r)); break;
        case 3: asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(r)); break;
        case 4: asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(r)); break;
        case 5: asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(r)); break;
        default: asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(r)); break;
    }
}

typedef unsigned int pws_u32x2 __attribute__((ext_vector_type(2)));
typedef float pws_f32x2 __attribute__((ext_vector_type(2)));
// fp16(acc + bias) for four accumulators: two v_pk_add_f32 on the accumulator's own register pairs, two v_cvt_pk_f16_f32 (RNE)
__device__ __forceinline__ half4_t pws_bias_cvt(const floatx16& acc, int q, const floatx4& bv) {
    pws_f32x2 lo = {acc[4 * q], acc[4 * q + 1]}, hi = {acc[4 * q + 2], acc[4 * q + 3]};
    const pws_f32x2 blo = {bv[0], bv[1]}, bhi = {bv[2], bv[3]};
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(lo) : "v"(lo), "v"(blo));
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(hi) : "v"(hi), "v"(bhi));
    pws_u32x2 r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r.x) : "v"(lo.x), "v"(lo.y));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r.y) : "v"(hi.x), "v"(hi.y));
    return __builtin_bit_cast(half4_t, r);
}

template <int K>
__global__ __launch_bounds__(pws::NT) void conv_pws_kernel(PwsArgs a) {
    using namespace pws;
    using L = Lay<K>;
    constexpr int KK = K / 16, NX = L::NX, X_BYTES = L::X_BYTES, SL_BYTES = L::SL_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void lds_void_t;
    const unsigned smem_base = (unsigned)(size_t)(lds_void_t*)smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frag_row = lane & 31, frag_half = lane >> 5;

    // CB-channel slabs of the output go to different blocks; the `halves` blocks that share an operand tile sit on the SAME XCD
    // (blocks are dealt round-robin to the 8 XCDs), as in conv_pw64.hip
    const int halves = a.c_out / CB;
    const int G = gridDim.x / halves;                       // tile streams
    int half, t0;
    if ((gridDim.x & 7) == 0 && ((gridDim.x >> 3) % halves) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        half = j % halves;
        t0 = xcd * ((gridDim.x >> 3) / halves) + j / halves;
    } else {
        half = blockIdx.x % halves;
        t0 = blockIdx.x / halves;
    }
    if (t0 >= a.n_tiles) return;
    const int T = (a.n_tiles - t0 + G - 1) / G;             // tiles of this block: t0, t0 + G, ...
    const int ldo = a.c_out;
    const int co0 = half * CB + wave * 32;                  // this wave's 32 output channels

    // ---- launch-resident operands: the wave's weight rows as 32x32x16 A fragments (K / 4 VGPRs), its bias ---------------------
    // (staged through a wave-private LDS scratch in 256-byte row pieces -- metro_common.h: fetched as 32-byte pieces straight from
    // global memory these K / 4 registers cost 17 us (K = 512) / 7 us (K = 256) of L2 requests before the first tile)
    half8_t wf[KK];
    static_assert(L::BIAS_OFF >= NW * W_STAGE_BYTES, "the staging scratch must not reach the bias block");
    load_w_frags_staged<K>(a.w + (size_t)co0 * K, wf, smem + wave * W_STAGE_BYTES, lane);
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) asm volatile("" : "+v"(wf[kk]));      // pinned: never rematerialised inside the tile loop
    float* bias_l = reinterpret_cast<float*>(smem + L::BIAS_OFF);
    if (tid < CB) bias_l[tid] = a.bias[half * CB + tid];

    // ---- per-lane coordinates ---------------------------------------------------------------------------------------------------
    // operand tile: DMA instruction q = i * 8 + wave fills 8 rows of slice q / 4 (32 rows per slice)
    int xoff[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int q = i * NW + wave;
        const int row = (q & 3) * 8 + (lane >> 3);
        xoff[i] = row * K + (q >> 2) * 64 + (((lane & 7) ^ pws_swz(row)) * 8);
    }
    auto issue_x = [&](int tile, int slot) {
        const half_t* src = a.in + (size_t)tile * TN * K;
#pragma unroll
        for (int i = 0; i < NX; ++i)
            pws_dma16(src + xoff[i], __builtin_amdgcn_readfirstlane(smem_base + L::X_OFF + slot * X_BYTES + (i * NW + wave) * 1024));
    };
    // shortcut rows / stores: lane l <-> pixel (l >> 2) + 16 it, 16-byte piece l & 3 of the wave's 64-byte row piece
    char* tw = smem + L::T_OFF + wave * (T_BYTES + R_BYTES);
    const unsigned rw_lds = smem_base + L::T_OFF + wave * (T_BYTES + R_BYTES) + T_BYTES;
    const int rpx = lane >> 2, rch = lane & 3;
    auto issue_res = [&](int tile) {
        const half_t* src = a.residual + ((size_t)tile * TN + rpx) * ldo + co0 + rch * 8;
        pws_dma16(src, __builtin_amdgcn_readfirstlane(rw_lds));
        pws_dma16(src + (size_t)16 * ldo, __builtin_amdgcn_readfirstlane(rw_lds + 1024));
    };
    // B fragment of k step kk: slice kk >> 2, row frag_row, chunk (2 (kk & 3) + frag_half) ^ swizzle
    const int boff0 = frag_row * 128 + ((frag_half ^ pws_swz(frag_row)) << 4);

    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");     // bias in LDS, weights in registers
    floatx4 bv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) bv[q] = *reinterpret_cast<const floatx4*>(bias_l + wave * 32 + 8 * q + 4 * frag_half);
    issue_x(t0, 0);
    if (T > 1) issue_x(t0 + G, 1);
    // this wave's share of tile 0 has landed BEFORE its first barrier (waves 0-3 multiply tile 0 right behind it)
    if (T > 1) pws_wait_vm<NX>();
    else pws_wait_vm<0>();
    if (wave >= 4) pws_barrier();        // waves 4-7 run one phase behind waves 0-3 from here on

    int slot = 0;
    for (int j = 0; j < T; ++j) {
        const int tile = t0 + j * G;
        // ---- phase 1: every wave's share of tile j has landed (each waited for its own: above for tile 0, behind the GEMM of tile
        //      j - 1 otherwise -- in both wave groups that wait lies in front of the barrier the OTHER group starts tile j behind) ----
        pws_barrier();
        issue_res(tile);                                         // consumed in this tile's epilogue
        if (j + 2 < T) issue_x(t0 + (j + 2) * G, slot >= 1 ? slot - 1 : 2);
        floatx16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        // During its GEMM phase this wave is ALONE on the matrix pipe of its SIMD (the other wave is in its epilogue): nothing hides an
        // LDS round trip, so the B fragments run DEPTH k steps ahead of their MFMA through a register ring.  Reads and waits are
        // inline asm with counted lgkmcnt: hipcc's own schedule requests a fragment one or two MFMAs ahead (~100 cycles per 32-cycle
        // MFMA), and collapses a ring written in C++ back into read - wait - MFMA.
        constexpr int DEPTH = 8;
        unsigned fbase[4];              // LDS byte address of k step c (c = kk & 3) of slice 0; slice kk >> 2 is the immediate offset
#pragma unroll
        for (int c = 0; c < 4; ++c) fbase[c] = smem_base + L::X_OFF + slot * X_BYTES + (boff0 ^ (c << 5));
        half8_t fr[DEPTH];
#pragma unroll
        for (int kk = 0; kk < DEPTH; ++kk)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[kk]) : "v"(fbase[kk & 3]), "n"((kk >> 2) * SL_BYTES));
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            // fragment kk has landed: at most min(DEPTH - 1, KK - 1 - kk) younger reads are still in flight
            if (KK - 1 - kk >= DEPTH - 1) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(fr[kk % DEPTH]) : "n"(DEPTH - 1));
            else pws_wait_lgkm_dyn(fr[kk % DEPTH], KK - 1 - kk);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[kk], fr[kk % DEPTH], acc, 0, 0, 0);
            if (kk + DEPTH < KK)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[kk % DEPTH]) : "v"(fbase[(kk + DEPTH) & 3]), "n"(((kk + DEPTH) >> 2) * SL_BYTES));
        }
        // the shortcut rows of this tile (and every older request: the operand tile j + 1 among them) have landed; the operand
        // requests of tile j + 2 stay in flight
        if (j + 2 < T) pws_wait_vm<NX>();
        else pws_wait_vm<0>();
        // ---- phase 2 (the other wave of this SIMD is in its phase 1 now) --------------------------------------------------------------
        pws_barrier();
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<half4_t*>(tw + frag_row * T_ROW + (8 * q + 4 * frag_half) * 2) = pws_bias_cvt(acc, q, bv[q]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the wave's own tile: no barrier
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int px = it * 16 + rpx;
            uint4 v = *reinterpret_cast<const uint4*>(tw + px * T_ROW + rch * 16);
            const uint4 rv = *reinterpret_cast<const uint4*>(tw + T_BYTES + it * 1024 + lane * 16);
            half8_t x = __builtin_bit_cast(half8_t, v);
            x = x + __builtin_bit_cast(half8_t, rv);             // the fp16 Add of the reference graph (resnet_v2.py:138)
            store_out16<1>(a.out + ((size_t)tile * TN + px) * ldo + co0 + rch * 8, __builtin_bit_cast(uint4, x));
        }
        slot = slot == 2 ? 0 : slot + 1;
    }
    if (wave < 4) pws_barrier();         // every wave executes 2 T + 1 barriers
}

bool conv_pws_supported(const MetroConvDesc& d) {
    static const int enabled = tuning_knob("METRO_PWS", 1);
    if (!enabled || classic_forms_forced()) return false;
    if (!(d.kh == 1 && d.kw == 1 && d.stride == 1 && d.pad_top == 0 && d.pad_left == 0 && d.in_pix_stride == d.c_in &&
          d.h_in == d.h_out && d.w_in == d.w_out && d.relu == 0 && d.out_dtype == METRO_F16 && d.in_dtype == METRO_F16))
        return false;
    const bool res_plain = d.has_residual && d.res_stride == 1 && d.res_offset == 0 && d.res_h == d.h_out && d.res_w == d.w_out;
    const long m = (long)d.n * d.h_out * d.w_out;
    if (!((d.c_in == 256 || d.c_in == 512) && d.c_out == 4 * d.c_in && !d.has_prologue && res_plain && m % pws::TN == 0)) return false;
    // Measured inside the forward (same box, tools/ab_bench.sh; batch 256 / 64).  K = 512 (block4): 175 -> 150 us / 51.5 -> 46: always.
    // K = 256 (block3): at batch 256 the lock-step kernel already runs at its HBM roof (302 MB in 67 us) and this one loses the
    // full-line stores (64-byte row pieces per wave): 68 -> 74 us, step +1 %; at batch 64 (8 work items per block: ramp and drain
    // count) it wins, 23.3 -> 21 us, step -0.55 %.  Same bits either way, so the choice may follow the batch.
    // (round 5, after the weight staging: at 16 work items per block -- RN101-s8 at batch 32, RN50-s16 at batch 128 -- the skewed kernel
    // still wins, -1.7 % / -1.0 % of the step; at 32 (batch 256) the lock-step one does, +1.3 %: profiles/r05_ab_pws_k256_items.txt)
    static const int k256_max_items = tuning_knob("METRO_PWS_K256_MAX_ITEMS", 16);
    if (d.c_in == 512) return true;
    const long items = (m / pws::TN) * (d.c_out / pws::CB);
    return items <= (long)k256_max_items * 256;
}

template <int K>
static int launch_pws(PwsArgs a, hipStream_t stream) {
    if (note_kernel("conv_pws<k%d,res>", K)) return METRO_OK;
    auto kern = conv_pws_kernel<K>;
    constexpr int lds = pws::Lay<K>::LDS;
    a.n_tiles = a.m_total / pws::TN;
    static PerDeviceInt cap;
    int grid_cap = 0;
    if (const int st = ensure_dyn_lds_and_grid_cap(reinterpret_cast<const void*>(kern), pws::NT, lds, cap, "conv_pws", 1, &grid_cap)) return st;
    const int halves = a.c_out / pws::CB;
    int grid = a.n_tiles * halves < grid_cap ? a.n_tiles * halves : grid_cap;
    grid -= grid % halves;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(pws::NT), lds, stream, a);
    return launch_status("conv_pws");
}

int launch_conv_pws(const MetroConvDesc& d, const void* in, const void* w, const float* bias, const void* res, void* out, hipStream_t stream) {
    if (!conv_pws_supported(d)) { set_error("conv_pws: unsupported layer"); return METRO_ERR_UNSUPPORTED; }
    PwsArgs a;
    a.in = static_cast<const half_t*>(in); a.w = static_cast<const half_t*>(w); a.bias = bias;
    a.residual = static_cast<const half_t*>(res); a.out = static_cast<half_t*>(out);
    a.m_total = d.n * d.h_out * d.w_out; a.n_tiles = 0; a.c_out = d.c_out;
    if (d.c_in == 512) return launch_pws<512>(a, stream);
    return launch_pws<256>(a, stream);
}

}  // namespace metro
