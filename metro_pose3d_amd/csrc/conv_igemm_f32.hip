// fp32 implicit-GEMM convolution on the fp32 matrix cores: the parity mode at fp32 SPEED (precision METRO_PREC_F32M).
//
// The f64 parity mode (conv_igemm_f64acc.hip) is what sits under the 1e-3 mm bar, at 3 390 crops/s.  This kernel is the
// arithmetic of the reference's own fp32 graph instead (reference src/options.py:73 `--dtype=float32`, the TF kernels of
// resnet_v2.py:123-136,219-220,233-236 / resnet_utils.py:82-135): fp32 activations in HBM, fp32 (BN-folded in fp64, rounded
// once) weights, every product and sum on v_mfma_f32_32x32x2_f32 -- exact fp32 products, fp32 accumulation in ascending k
// (bit for bit an fmaf chain, MI355X_MICROARCH.md) -- one fp32 rounding per output.  It therefore sits at the noise floor two
// correct fp32 implementations have between them (1-5e-3 mm on these nets), not under it.
//
// Same descriptor, padding rules, prologue (pre-activation BN + ReLU, fp32 FMA) and bias / ReLU / shortcut epilogue as the
// other conv kernels; any c_in (the 3-channel stem runs here with TF's explicit pad 3).  MFMA roles as in the f16 kernels:
// A = weight rows (cout), B = pixels.  Tile 128 cout x 128 pixels x 32 k, four waves of 64 x 64 (2 x 2 tiles of 32 x 32);
// a k step is 16 MFMA k-pairs = 64 MFMAs of 64 cycles per wave against 16 + 16 scalar gathers per thread: the launch is bound
// by the fp32 matrix pipe (157 TFLOP/s peak).  The gathers of step s+1 are issued before the MFMAs of step s (registers),
// written to the other LDS buffer behind them: one barrier per step.
#include "metro_common.h"

namespace metro {

typedef float floatx16 __attribute__((ext_vector_type(16)));

namespace f32m {
#ifndef METRO_F32M_BK
#define METRO_F32M_BK 32
#endif
constexpr int TN = 128, BK = METRO_F32M_BK, NT = 256;
constexpr int LD = BK + 1;                       // padded LDS rows (floats): conflict-free ds_read_b32 of 32 rows x one k
}  // namespace f32m

// TM = 128: wave tile 64 cout x 64 pixels;  TM = 64: 32 cout x 64 pixels -- the 64-channel layers of block1, and layers whose
// 128-cout tiling would leave CUs with fewer than two blocks (the MFMAs of one wave per SIMD do not cover a barrier per step)
// V = 4: the gathers are 16-byte loads of four consecutive channels (c_in % 4 == 0: every layer but the 3-channel stem) --
// a quarter of the load instructions and of the address arithmetic, which is what the launch was issue-bound by (60 % of the
// fp32 MFMA peak with scalar gathers)
template <bool PROLOGUE, int TM, int V>
__global__ __launch_bounds__(f32m::NT) void conv_igemm_f32_kernel(
    ConvArgs a, const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ pro_scale, const float* __restrict__ pro_shift, const float* __restrict__ residual,
    float* __restrict__ out) {
    using namespace f32m;
    constexpr int MI = TM / 64;                      // 32-cout MFMA tiles per wave

    constexpr int KC = BK / V;                       // loader columns of V consecutive k
    constexpr int KR = NT / KC;                      // rows a loader pass covers (thread -> k column V (tid % KC), row tid / KC)
    constexpr int XR = TN / KR;                      // pixel rows per loader thread per step
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float (*ws)[TM * LD] = reinterpret_cast<float (*)[TM * LD]>(smem_raw);
    float (*xs)[TN * LD] = reinterpret_cast<float (*)[TN * LD]>(smem_raw + 2 * TM * LD * sizeof(float));

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wave_m = wave >> 1;   // cout half
    const int wave_n = wave & 1;    // 64-pixel half
    const int tiles_m = (a.c_out + TM - 1) / TM;
    const int m0 = (blockIdx.x / tiles_m) * TN;      // first pixel
    const int n0 = (blockIdx.x % tiles_m) * TM;      // first cout
    const int k_total = a.kh * a.kw * a.c_in;
    const int hw_out = a.h_out * a.w_out;

    // loader: thread -> k column tid & 15, rows (tid >> 4) + 16 i of both operands (8 each)
    constexpr int WR = TM / KR;                      // weight rows a loader thread fills per step
    const int kcol = (tid % KC) * V;
    const int lrow = tid / KC;
    int xh[XR], xw[XR], xn[XR];
#pragma unroll
    for (int i = 0; i < XR; ++i) {
        const int m = m0 + lrow + KR * i;
        const bool ok = m < a.m_total;
        const int mm = ok ? m : 0;
        const int img = mm / hw_out;
        const int rem = mm - img * hw_out;
        const int ho = rem / a.w_out;
        const int wo = rem - ho * a.w_out;
        xh[i] = ok ? ho * a.stride - a.pad_top : -(1 << 28);     // out-of-range rows fail the bounds test below
        xw[i] = wo * a.stride - a.pad_left;
        xn[i] = img * a.h_in * a.w_in;
    }
    float xv[XR][V], wv[WR][V], g_sc[V], g_sh[V];
    bool xok[XR], wokv[WR];
    auto gather = [&](int k0) {
        const int k = k0 + kcol;
        const bool kvalid = k < k_total;
        const int kk = kvalid ? k : 0;
        const int tap = kk / a.c_in;
        const int c = kk - tap * a.c_in;
        const int r = tap / a.kw;
        const int s = tap - r * a.kw;
        typedef float vecV __attribute__((ext_vector_type(V)));
        // every load is issued unconditionally from a clamped address and masked afterwards: a load inside `if (ok)` makes hipcc
        // branch around it and wait for it alone -- sixteen dependent memory round trips per step.  (c is in range also for a
        // masked k column; with V = 4 the four k share tap and pixel: c_in % 4 == 0.)
        if (PROLOGUE) {
            if constexpr (V == 1) { g_sc[0] = pro_scale[c]; g_sh[0] = pro_shift[c]; }
            else {
                const vecV sv = *reinterpret_cast<const vecV*>(pro_scale + c), hv = *reinterpret_cast<const vecV*>(pro_shift + c);
#pragma unroll
                for (int e = 0; e < V; ++e) { g_sc[e] = sv[e]; g_sh[e] = hv[e]; }
            }
        }
        const size_t wrow = (size_t)(kvalid ? k : 0);
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            const int hi = xh[i] + r * a.dil;
            const int wi = xw[i] + s * a.dil;
            const bool ok = kvalid && (unsigned)hi < (unsigned)a.h_in && (unsigned)wi < (unsigned)a.w_in;
            const int pix = ok ? xn[i] + hi * a.w_in + wi : 0;
            if constexpr (V == 1) xv[i][0] = in[(size_t)pix * a.in_pix_stride + c];
            else {
                const vecV t = *reinterpret_cast<const vecV*>(in + (size_t)pix * a.in_pix_stride + c);
#pragma unroll
                for (int e = 0; e < V; ++e) xv[i][e] = t[e];
            }
            xok[i] = ok;
        }
#pragma unroll
        for (int i = 0; i < WR; ++i) {
            const int co = n0 + lrow + KR * i;
            const bool wok = kvalid && co < a.c_out;
            if constexpr (V == 1) wv[i][0] = w[(size_t)(wok ? co : 0) * k_total + wrow];
            else {
                const vecV t = *reinterpret_cast<const vecV*>(w + (size_t)(wok ? co : 0) * k_total + wrow);
#pragma unroll
                for (int e = 0; e < V; ++e) wv[i][e] = t[e];
            }
            wokv[i] = wok;
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int i = 0; i < XR; ++i)
#pragma unroll
            for (int e = 0; e < V; ++e) {
                float v = xv[i][e];
                if (PROLOGUE) v = fmaxf(fmaf(v, g_sc[e], g_sh[e]), 0.f);      // pre-activation BN + ReLU (resnet_v2.py:119,229), one rounding
                xs[buf][(lrow + KR * i) * LD + kcol + e] = xok[i] ? v : 0.f;
            }
#pragma unroll
        for (int i = 0; i < WR; ++i)
#pragma unroll
            for (int e = 0; e < V; ++e) ws[buf][(lrow + KR * i) * LD + kcol + e] = wokv[i] ? wv[i][e] : 0.f;
    };

    floatx16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int frag_row = lane & 31, frag_k = lane >> 5;       // A[m = lane & 31][k = lane >> 5], B likewise by pixel
    gather(0);
    commit(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < k_total; k0 += BK) {
        const bool more = k0 + BK < k_total;
        gather(more ? k0 + BK : k0);                 // in flight under the MFMAs below (past the end: the last step again, unused)
        const float* wl = ws[buf] + (wave_m * (TM / 2) + frag_row) * LD + frag_k;
        const float* xl = xs[buf] + (wave_n * 64 + frag_row) * LD + frag_k;
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
            const float b0 = xl[2 * ks], b1 = xl[32 * LD + 2 * ks];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const float ai = wl[i * 32 * LD + 2 * ks];
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ai, b0, acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ai, b1, acc[i][1], 0, 0, 0);
            }
        }
        commit(buf ^ 1);                             // the other buffer: last read before the previous barrier
        __syncthreads();
        buf ^= 1;
    }

    // epilogue: D[row = cout 8 q + 4 (lane >> 5) + e][col = pixel lane & 31] per 32 x 32 tile: a lane owns FOUR consecutive output
    // channels of a pixel per (i, q) -> 16-byte stores (and shortcut loads, all issued before the first use) when c_out % 4 == 0
    typedef float floatx4 __attribute__((ext_vector_type(4)));
    const bool vec = (a.c_out & 3) == 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = m0 + wave_n * 64 + j * 32 + frag_row;
        const bool mok = m < a.m_total;
        size_t res_pix = 0;
        if (residual != nullptr) {
            const int mm = mok ? m : 0;
            const int img = mm / hw_out;
            const int rem = mm - img * hw_out;
            const int ho = rem / a.w_out;
            const int wo = rem - ho * a.w_out;
            res_pix = (size_t)(img * a.res_h + ho * a.res_stride + a.res_offset) * a.res_w + (wo * a.res_stride + a.res_offset);
        }
        if (vec) {
            floatx4 rv[MI][4];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int co = n0 + wave_m * (TM / 2) + i * 32 + 8 * q + 4 * frag_k;
                    rv[i][q] = floatx4{0.f, 0.f, 0.f, 0.f};
                    if (residual != nullptr)        // wave-uniform branch; clamped address, masked below by the store predicate
                        rv[i][q] = *reinterpret_cast<const floatx4*>(residual + res_pix * a.c_out + (co < a.c_out ? co : 0));
                }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int co = n0 + wave_m * (TM / 2) + i * 32 + 8 * q + 4 * frag_k;
                    const floatx4 bv = *reinterpret_cast<const floatx4*>(bias + (co < a.c_out ? co : 0));
                    floatx4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = acc[i][j][4 * q + e] + bv[e];
                        if (a.relu) t = fmaxf(t, 0.f);
                        v[e] = t + rv[i][q][e];
                    }
                    if (mok && co < a.c_out) *reinterpret_cast<floatx4*>(out + (size_t)m * a.c_out + co) = v;
                }
        } else {
            if (!mok) continue;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int co = n0 + wave_m * (TM / 2) + i * 32 + 8 * q + 4 * frag_k + e;
                        if (co >= a.c_out) continue;
                        float v = acc[i][j][4 * q + e] + bias[co];
                        if (a.relu) v = fmaxf(v, 0.f);
                        if (residual != nullptr) v += residual[res_pix * a.c_out + co];
                        out[(size_t)m * a.c_out + co] = v;
                    }
        }
    }
}

int launch_conv_f32m(const MetroConvDesc& d, const void* in, const float* w, const float* bias, const float* ps, const float* pb,
                     const void* res, void* out, hipStream_t stream) {
    if (d.in_dtype != METRO_F32 || d.out_dtype != METRO_F32) {
        set_error("conv_f32m: fp32 tensors only (in/out dtypes %d/%d)", d.in_dtype, d.out_dtype);
        return METRO_ERR_UNSUPPORTED;
    }
    const ConvArgs a = make_conv_args(d);
    const bool pro = d.has_prologue != 0;
    // 64-cout tiles for the 64-channel layers and wherever 128-cout tiles give the CUs fewer than two blocks each
    const long tiles_n = (a.m_total + f32m::TN - 1) / f32m::TN;
    const bool small = a.c_out <= 64;          // (64-cout tiles for layers with < 2 blocks per CU measured 4 % slower on block3)
    const int tm = small ? 64 : 128;
    const bool vec4 = d.c_in % 4 == 0 && d.in_pix_stride % 4 == 0;       // 16-byte gathers of four consecutive channels
    if (note_kernel("conv_igemm_f32<%dx128,bk%d%s%s>%s", tm, f32m::BK, vec4 ? ",v4" : "", pro ? ",pro" : "", d.has_residual ? "+res" : "")) return METRO_OK;
    const int tiles_m = (a.c_out + tm - 1) / tm;
    const float* r = d.has_residual ? static_cast<const float*>(res) : nullptr;
    const dim3 grid((unsigned)(tiles_m * tiles_n)), blk(f32m::NT);
#define METRO_F32M_LAUNCH(PRO, TMV)                                                                                            \
    if (vec4) hipLaunchKernelGGL((conv_igemm_f32_kernel<PRO, TMV, 4>), grid, blk, 2 * (TMV + f32m::TN) * f32m::LD * sizeof(float), stream, a, \
                                 static_cast<const float*>(in), w, bias, ps, pb, r, static_cast<float*>(out));                 \
    else hipLaunchKernelGGL((conv_igemm_f32_kernel<PRO, TMV, 1>), grid, blk, 2 * (TMV + f32m::TN) * f32m::LD * sizeof(float), stream, a, static_cast<const float*>(in), w, bias, ps, pb, r, \
                       static_cast<float*>(out))
    {   // > 64 KiB of dynamic LDS needs the opt-in, once per device and instantiation
        static PerDeviceInt done[8];
        const int which = (pro ? 4 : 0) + (small ? 2 : 0) + (vec4 ? 1 : 0);
        const void* kps[8] = {
            reinterpret_cast<const void*>(conv_igemm_f32_kernel<false, 128, 1>), reinterpret_cast<const void*>(conv_igemm_f32_kernel<false, 128, 4>),
            reinterpret_cast<const void*>(conv_igemm_f32_kernel<false, 64, 1>), reinterpret_cast<const void*>(conv_igemm_f32_kernel<false, 64, 4>),
            reinterpret_cast<const void*>(conv_igemm_f32_kernel<true, 128, 1>), reinterpret_cast<const void*>(conv_igemm_f32_kernel<true, 128, 4>),
            reinterpret_cast<const void*>(conv_igemm_f32_kernel<true, 64, 1>), reinterpret_cast<const void*>(conv_igemm_f32_kernel<true, 64, 4>)};
        if (const int st = ensure_dyn_lds(kps[which], 2 * ((small ? 64 : 128) + f32m::TN) * f32m::LD * (int)sizeof(float), done[which], "conv_igemm_f32")) return st;
    }
    if (pro) { if (small) METRO_F32M_LAUNCH(true, 64); else METRO_F32M_LAUNCH(true, 128); }
    else { if (small) METRO_F32M_LAUNCH(false, 64); else METRO_F32M_LAUNCH(false, 128); }
#undef METRO_F32M_LAUNCH
    return launch_status("conv_igemm_f32");
}

}  // namespace metro
