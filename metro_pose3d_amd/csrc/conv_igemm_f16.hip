// Implicit-GEMM NHWC convolution on the CDNA4 matrix cores: fp16 operands, fp32 accumulation
// with v_mfma_f32_32x32x16_f16, wave64.  One kernel covers every convolution of the backbone
// and head (reference call sites: resnet_v2.py:123-136,219-220,233-236 via slim.conv2d and
// resnet_utils.conv2d_same, resnet_utils.py:82-135):
//
//   D[cout][pixel] = sum_{tap, c} W[cout][tap][c] * X[pixel shifted by tap][c]
//
// * MFMA A-operand = weight rows (cout), B-operand = pixels, so every lane ends up holding 4
//   CONSECUTIVE output channels of one pixel per accumulator quad -> 8-byte NHWC stores.
// * Both operand tiles are "rows of contiguous 16-byte chunks" (a weight row is contiguous in
//   [cout][tap][c]; a pixel's channels are contiguous in NHWC), staged global -> VGPR -> LDS.
//   Register staging (not LDS-DMA) is deliberate: TF padding needs per-chunk zero fill and the
//   pre-activation BatchNorm+ReLU (resnet_v2.py:119,229) is applied to the activation chunk on
//   its way into LDS, so the residual stream is read once and never re-written normalised.
// * LDS tiles are [rows][BK] fp16 with the 16-byte chunk index XOR-swizzled by the row so that
//   the ds_read_b128 fragment reads (32 rows x one chunk per half-wave) are bank-conflict free.
// * Double-buffered LDS, one barrier per K step: loads for step k+1 are in flight while the
//   MFMAs of step k run.
// * Epilogue: + bias, optional ReLU, optional residual add with the (strided, shifted) gather
//   of the identity shortcut (resnet_v2.py:113-121, resnet_utils.py:76-79), one rounding.
#include <cstdlib>

#include "metro_common.h"

namespace metro {

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int WAVES_M_, int WAVES_N_, int WM_, int WN_>
struct TileCfg {
    static constexpr int WAVES_M = WAVES_M_, WAVES_N = WAVES_N_, WM = WM_, WN = WN_;
    static constexpr int BK = 64;                        // K elements per step
    static constexpr int NT = 64 * WAVES_M * WAVES_N;    // threads per block
    static constexpr int TM = WAVES_M * WM * 32;         // output channels per block
    static constexpr int TN = WAVES_N * WN * 32;         // pixels per block
    static constexpr int CPR = BK / 8;                   // 16-byte chunks per tile row
    static constexpr int ROWS_PER_PASS = NT / CPR;
    static constexpr int W_CHUNKS = TM / ROWS_PER_PASS;  // chunks each thread stages per step
    static constexpr int X_CHUNKS = TN / ROWS_PER_PASS;
    static constexpr int ROW_BYTES = BK * 2;
    static constexpr int W_TILE_BYTES = TM * ROW_BYTES;
    static constexpr int X_TILE_BYTES = TN * ROW_BYTES;
    static constexpr int BUF_BYTES = W_TILE_BYTES + X_TILE_BYTES;
    static constexpr int LDS_BYTES = 2 * BUF_BYTES;
    static_assert(TM % ROWS_PER_PASS == 0 && TN % ROWS_PER_PASS == 0, "tile/loader mismatch");
};

// chunk swizzle: 128-byte rows, 2 rows per 256-byte bank row -> XOR with (row>>1)&7
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

template <class Cfg, bool PROLOGUE, bool ALIGN8>
__global__ __launch_bounds__(Cfg::NT) void conv_igemm_f16_kernel(
    ConvArgs a, const half_t* __restrict__ in, const half_t* __restrict__ w,
    const float* __restrict__ bias, const half_t* __restrict__ pro_scale,
    const half_t* __restrict__ pro_shift, const half_t* __restrict__ residual,
    void* __restrict__ out, int out_f32, int tiles_m) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BK = Cfg::BK, CPR = Cfg::CPR;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wave_m = wave / Cfg::WAVES_N;
    const int wave_n = wave % Cfg::WAVES_N;

    // XCD-aware tile order: hardware places block b on XCD b % 8; give each XCD a contiguous
    // range of logical tiles so blocks that share a pixel tile (and its activations) share an L2.
    const int nblk = gridDim.x;
    int lid;
    {
        const int b = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = b & 7, idx = b >> 3;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = lid / tiles_m;   // pixel tile
    const int tile_m = lid % tiles_m;   // cout tile
    const int m0 = tile_n * Cfg::TN;    // first pixel
    const int n0 = tile_m * Cfg::TM;    // first output channel

    const int taps = a.kh * a.kw;
    const int k_total = taps * a.c_in;
    const int kc_steps = (a.c_in + BK - 1) / BK;
    const int nk = taps * kc_steps;

    // ---- loader coordinates -------------------------------------------------------------
    const int ch = tid % CPR;            // this thread's 16-byte chunk within a tile row
    const int row0 = tid / CPR;
    const half_t* wrow[Cfg::W_CHUNKS];
    bool wvalid[Cfg::W_CHUNKS];
#pragma unroll
    for (int i = 0; i < Cfg::W_CHUNKS; ++i) {
        const int co = n0 + row0 + i * Cfg::ROWS_PER_PASS;
        wvalid[i] = co < a.c_out;
        wrow[i] = w + (size_t)(wvalid[i] ? co : 0) * k_total;
    }
    int xh[Cfg::X_CHUNKS], xw[Cfg::X_CHUNKS], xn[Cfg::X_CHUNKS];
    bool xvalid[Cfg::X_CHUNKS];
    const int hw_out = a.h_out * a.w_out;
#pragma unroll
    for (int i = 0; i < Cfg::X_CHUNKS; ++i) {
        const int m = m0 + row0 + i * Cfg::ROWS_PER_PASS;
        xvalid[i] = m < a.m_total;
        const int mm = xvalid[i] ? m : 0;
        const int img = mm / hw_out;
        const int rem = mm - img * hw_out;
        const int ho = rem / a.w_out;
        const int wo = rem - ho * a.w_out;
        xh[i] = ho * a.stride - a.pad_top;
        xw[i] = wo * a.stride - a.pad_left;
        xn[i] = img * a.h_in * a.w_in;
    }

    uint4 wreg[Cfg::W_CHUNKS], xreg[Cfg::X_CHUNKS];
    uint4 sreg = make_uint4(0, 0, 0, 0), breg = make_uint4(0, 0, 0, 0);  // prologue scale/shift

    auto load_step = [&](int kstep) {
        const int tap = kstep / kc_steps;
        const int c = (kstep - tap * kc_steps) * BK + ch * 8;
        const bool cvalid = c < a.c_in;
        const int r = tap / a.kw;
        const int s = tap - r * a.kw;
        const int koff = tap * a.c_in + c;
#pragma unroll
        for (int i = 0; i < Cfg::W_CHUNKS; ++i) {
            wreg[i] = (wvalid[i] && cvalid) ? *reinterpret_cast<const uint4*>(wrow[i] + koff)
                                           : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < Cfg::X_CHUNKS; ++i) {
            const int hi = xh[i] + r * a.dil;
            const int wi = xw[i] + s * a.dil;
            const bool ok = xvalid[i] && cvalid && (unsigned)hi < (unsigned)a.h_in &&
                            (unsigned)wi < (unsigned)a.w_in;
            if (ok) {
                const half_t* p = in + (size_t)(xn[i] + hi * a.w_in + wi) * a.in_pix_stride + c;
                if (ALIGN8) {
                    const uint2 lo = reinterpret_cast<const uint2*>(p)[0];
                    const uint2 hi2 = reinterpret_cast<const uint2*>(p)[1];
                    xreg[i] = make_uint4(lo.x, lo.y, hi2.x, hi2.y);
                } else {
                    xreg[i] = *reinterpret_cast<const uint4*>(p);
                }
            } else {
                xreg[i] = make_uint4(0, 0, 0, 0);
            }
        }
        if (PROLOGUE) {
            if (cvalid) {
                sreg = *reinterpret_cast<const uint4*>(pro_scale + c);
                breg = *reinterpret_cast<const uint4*>(pro_shift + c);
            } else {
                sreg = make_uint4(0, 0, 0, 0);
                breg = make_uint4(0, 0, 0, 0);
            }
        }
    };

    auto store_step = [&](int buf) {
        char* wl = smem + buf * Cfg::BUF_BYTES;
        char* xl = wl + Cfg::W_TILE_BYTES;
#pragma unroll
        for (int i = 0; i < Cfg::W_CHUNKS; ++i) {
            const int row = row0 + i * Cfg::ROWS_PER_PASS;
            *reinterpret_cast<uint4*>(wl + row * Cfg::ROW_BYTES + ((ch ^ swz(row)) << 4)) = wreg[i];
        }
#pragma unroll
        for (int i = 0; i < Cfg::X_CHUNKS; ++i) {
            const int row = row0 + i * Cfg::ROWS_PER_PASS;
            uint4 v = xreg[i];
            if (PROLOGUE) {
                // relu(x * scale[c] + shift[c]) on 8 halfs: 4 v_pk_fma_f16 + 4 v_pk_max_f16
                const half2_t* sv = reinterpret_cast<const half2_t*>(&sreg);
                const half2_t* bv = reinterpret_cast<const half2_t*>(&breg);
                half2_t* xv = reinterpret_cast<half2_t*>(&v);
                const half2_t zero = {(half_t)0, (half_t)0};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    half2_t t = xv[e] * sv[e] + bv[e];
                    xv[e] = __builtin_elementwise_max(t, zero);
                }
            }
            *reinterpret_cast<uint4*>(xl + row * Cfg::ROW_BYTES + ((ch ^ swz(row)) << 4)) = v;
        }
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

    auto compute_step = [&](int buf) {
        const char* wl = smem + buf * Cfg::BUF_BYTES;
        const char* xl = wl + Cfg::W_TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            half8_t af[Cfg::WM], bf[Cfg::WN];
            const int chunk = kk * 2 + frag_half;
#pragma unroll
            for (int i = 0; i < Cfg::WM; ++i) {
                const int row = (wave_m * Cfg::WM + i) * 32 + frag_row;
                af[i] = *reinterpret_cast<const half8_t*>(wl + row * Cfg::ROW_BYTES +
                                                          ((chunk ^ swz(row)) << 4));
            }
#pragma unroll
            for (int j = 0; j < Cfg::WN; ++j) {
                const int row = (wave_n * Cfg::WN + j) * 32 + frag_row;
                bf[j] = *reinterpret_cast<const half8_t*>(xl + row * Cfg::ROW_BYTES +
                                                          ((chunk ^ swz(row)) << 4));
            }
#pragma unroll
            for (int i = 0; i < Cfg::WM; ++i)
#pragma unroll
                for (int j = 0; j < Cfg::WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    };

    // ---- main loop --------------------------------------------------------------------
    load_step(0);
    store_step(0);
    __syncthreads();
    for (int k = 0; k < nk; ++k) {
        const int buf = k & 1;
        if (k + 1 < nk) load_step(k + 1);
        compute_step(buf);
        if (k + 1 < nk) store_step(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue -----------------------------------------------------------------------
    // acc[i][j][r]: cout = n0 + (wave_m*WM+i)*32 + (r&3) + 8*(r>>2) + 4*(lane>>5)
    //               pixel = m0 + (wave_n*WN+j)*32 + (lane&31)
#pragma unroll
    for (int j = 0; j < Cfg::WN; ++j) {
        const int m = m0 + (wave_n * Cfg::WN + j) * 32 + frag_row;
        if (m >= a.m_total) continue;
        size_t res_pix = 0;
        if (residual != nullptr) {
            const int img = m / hw_out;
            const int rem = m - img * hw_out;
            const int ho = rem / a.w_out;
            const int wo = rem - ho * a.w_out;
            res_pix = (size_t)(img * a.res_h + ho * a.res_stride + a.res_offset) * a.res_w +
                      (wo * a.res_stride + a.res_offset);
        }
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
                if (residual != nullptr) {
                    const half4_t rv =
                        *reinterpret_cast<const half4_t*>(residual + res_pix * a.c_out + co);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
                }
                if (out_f32) {
                    *reinterpret_cast<floatx4*>(reinterpret_cast<float*>(out) + (size_t)m * a.c_out + co) = v;
                } else {
                    half4_t hv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) hv[e] = (half_t)v[e];
                    *reinterpret_cast<half4_t*>(reinterpret_cast<half_t*>(out) + (size_t)m * a.c_out + co) = hv;
                }
            }
        }
    }
}

template <class Cfg, bool PROLOGUE, bool ALIGN8>
static int launch_cfg(const ConvArgs& a, const half_t* in, const half_t* w, const float* bias,
                      const half_t* ps, const half_t* pb, const half_t* res, void* out, int out_f32,
                      hipStream_t stream) {
    auto kern = conv_igemm_f16_kernel<Cfg, PROLOGUE, ALIGN8>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(conv_igemm_f16): %s", hipGetErrorString(e));
            return METRO_ERR_HIP;
        }
        attr_set = true;
    }
    const int tiles_m = (a.c_out + Cfg::TM - 1) / Cfg::TM;
    const int tiles_n = (a.m_total + Cfg::TN - 1) / Cfg::TN;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(Cfg::NT), Cfg::LDS_BYTES, stream, a, in, w,
                       bias, ps, pb, res, out, out_f32, tiles_m);
    return launch_status("conv_igemm_f16");
}

using Cfg128x128 = TileCfg<2, 2, 2, 2>;   // 128 cout x 128 pixels, wave tile 64x64
using Cfg128x64 = TileCfg<2, 2, 2, 1>;    // 128 cout x  64 pixels, wave tile 64x32
using Cfg64x128 = TileCfg<1, 4, 2, 1>;    //  64 cout x 128 pixels, wave tile 64x32
using Cfg64x64 = TileCfg<2, 2, 1, 1>;     //  64 cout x  64 pixels, wave tile 32x32

// METRO_CONV_VARIANT=0 forces the first-generation register-staged kernel everywhere (A/B runs)
static int conv_variant() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("METRO_CONV_VARIANT");
        v = e ? atoi(e) : 1;
    }
    return v;
}

int launch_conv_f16(const MetroConvDesc& d, const void* in_, const void* w_, const float* bias,
                    const void* ps_, const void* pb_, const void* res_, void* out, hipStream_t stream) {
    if (conv_variant() != 0 && conv3x3_slab_supported(d))
        return launch_conv3x3_slab(d, in_, w_, bias, out, stream);
    if (conv_variant() != 0 && conv_f16_dma_supported(d))
        return launch_conv_f16_dma(d, in_, w_, bias, ps_, pb_, res_, out, stream);
    const ConvArgs a = make_conv_args(d);
    const half_t* in = static_cast<const half_t*>(in_);
    const half_t* w = static_cast<const half_t*>(w_);
    const half_t* ps = static_cast<const half_t*>(ps_);
    const half_t* pb = static_cast<const half_t*>(pb_);
    const half_t* res = d.has_residual ? static_cast<const half_t*>(res_) : nullptr;
    const int out_f32 = d.out_dtype == METRO_F32;
    const bool align8 = (d.in_pix_stride % 8) != 0;   // stem: 4-channel padded image
    const bool pro = d.has_prologue != 0;

    // tile choice: wide cout tiles when c_out fills them, narrower pixel tiles when the layer
    // would otherwise not give every CU two blocks (256 CUs, 2 blocks/CU resident).
    const bool big_m = d.c_out >= 128 && (d.c_out % 128 == 0 || d.c_out > 256);
    const int tm = big_m ? 128 : 64;
    const long blocks128 = (long)((d.c_out + tm - 1) / tm) * ((a.m_total + 127) / 128);
    const bool big_n = blocks128 >= 512;

    if (align8) {
        if (pro) { set_error("conv_f16: prologue with 8-byte-aligned input is not built"); return METRO_ERR_UNSUPPORTED; }
        return launch_cfg<Cfg64x128, false, true>(a, in, w, bias, ps, pb, res, out, out_f32, stream);
    }
#define METRO_DISPATCH(CFG)                                                                  \
    return pro ? launch_cfg<CFG, true, false>(a, in, w, bias, ps, pb, res, out, out_f32, stream) \
               : launch_cfg<CFG, false, false>(a, in, w, bias, ps, pb, res, out, out_f32, stream)
    if (big_m && big_n) { METRO_DISPATCH(Cfg128x128); }
    if (big_m) { METRO_DISPATCH(Cfg128x64); }
    if (big_n) { METRO_DISPATCH(Cfg64x128); }
    METRO_DISPATCH(Cfg64x64);
#undef METRO_DISPATCH
}

}  // namespace metro
