// Parity-mode implicit-GEMM convolution: fp32 or fp64 NHWC activations in HBM (template
// TIn / TAct), fp64 (BN-folded) weights, every product and sum carried by
// v_mfma_f64_16x16x4_f64, ONE rounding per output element (none when TAct = double).  Same call sites as conv_igemm_f16.hip (reference resnet_v2.py:123-136,
// 219-220,233-236; resnet_utils.py:82-135), same descriptor, any c_in (the 3-channel stem runs
// here directly with TF's explicit pad-3, resnet_utils.py:125-135).
//
// Why fp64 accumulate: the target is <= 1e-3 mm against the fp64 oracle, i.e. 4.5e-7 of the
// 2200 mm box ~ 4 ulp of fp32.  A k-ordered fp32 fmaf chain (what v_mfma_f32_32x32x2_f32 is,
// bit for bit) over K up to 4608 lands at 1-4e-3 mm; accumulating in fp64 leaves only the
// per-layer storage rounding of the fp32 activations.
//
// MFMA operand roles: A = pixels (rows), B = weights (columns = cout), so the 16 lanes of a
// quarter-wave hold 16 consecutive output channels of one pixel -> 64-byte fp32 row segments.
//   A: lane l holds X[pixel = l&15][k = l>>4]      B: lane l holds W[k = l>>4][cout = l&15]
//   D: reg r of lane l = D[pixel = (l>>4) + 4r][cout = l&15]
#include "metro_common.h"

namespace metro {

typedef double doublex4 __attribute__((ext_vector_type(4)));

// Round 4: the structure the fp32-MFMA kernel (conv_igemm_f32.hip) got in round 3, for the fp64 matrix cores.  The first form
// (64 x 64 tile, one LDS buffer, two barriers per 16-channel step, gathers inside `if (ok)` -- every step paid a dependent memory
// round trip -- and two integer divisions per step) ran the forward at 35.8 TFLOP/s (45 % of the 78.6 TFLOP/s fp64 MFMA peak).  Now:
//   * tile 128 pixels x 128 couts (64 couts for the 64-channel layers) x 16 k, four waves of 64 x 64 (4 x 4 MFMA tiles of 16 x 16:
//     128 accumulator registers), 64 MFMAs of 64 cycles per wave and step;
//   * the operands of step s + 1 are requested before the MFMAs of step s (unconditionally, from clamped addresses, masked
//     afterwards; 16-byte loads of two consecutive channels when c_in is even), converted, pre-activated and written to the OTHER
//     LDS buffer behind them: ONE barrier per step;
//   * the tap / channel of a thread's k column is advanced incrementally.
// Arithmetic unchanged: fp64 products and sums on v_mfma_f64_16x16x4_f64, k ascending in groups of four -- the same bits as before.
namespace f64k {
constexpr int TMP = 128;          // pixels per block
constexpr int BK = 16;
constexpr int LD = BK + 1;        // padded LDS row (doubles) -> (almost) conflict-free ds_read_b64
constexpr int NT = 256;           // 4 waves, 2 x 2
}  // namespace f64k

template <bool PROLOGUE, typename TIn, typename TAct, int TN, int V>
__global__ __launch_bounds__(f64k::NT, 2) void conv_igemm_f64acc_kernel(
    ConvArgs a, const TIn* __restrict__ in, const double* __restrict__ w,
    const double* __restrict__ bias, const double* __restrict__ pro_scale,
    const double* __restrict__ pro_shift, const TAct* __restrict__ residual,
    TAct* __restrict__ out) {
    using namespace f64k;
    constexpr int NJ = TN / 32;                      // 16-cout MFMA tiles per wave (4 or 2)
    constexpr int KC = BK / V;                       // loader columns of V consecutive k
    constexpr int KR = NT / KC;                      // rows a loader pass covers
    constexpr int XR = TMP / KR;                     // pixel rows per loader thread per step
    constexpr int WR = TN / KR;                      // weight rows per loader thread per step
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double (*xs)[TMP * LD] = reinterpret_cast<double (*)[TMP * LD]>(smem_raw);
    double (*ws)[TN * LD] = reinterpret_cast<double (*)[TN * LD]>(smem_raw + 2 * TMP * LD * sizeof(double));

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wave_p = wave >> 1;   // 64-pixel half
    const int wave_c = wave & 1;    // cout half
    const int tiles_c = (a.c_out + TN - 1) / TN;
    const int m0 = (blockIdx.x / tiles_c) * TMP;
    const int n0 = (blockIdx.x % tiles_c) * TN;
    const int k_total = a.kh * a.kw * a.c_in;
    const int hw_out = a.h_out * a.w_out;

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
    TIn xv[XR][V];
    double wv[WR][V], g_sc[V], g_sh[V];
    bool xok[XR], wokv[WR];
    // (channel, tap row, tap column) of this thread's k column, advanced by BK per gather
    int g_c = kcol, g_r = 0, g_s = 0;
    while (g_c >= a.c_in) { g_c -= a.c_in; if (++g_s == a.kw) { g_s = 0; ++g_r; } }
    auto gather = [&](int k0) {
        const int k = k0 + kcol;
        const bool kvalid = k < k_total;
        const int c = kvalid ? g_c : 0;
        const int r = kvalid ? g_r : 0;
        const int sx = kvalid ? g_s : 0;
        g_c += BK;
        while (g_c >= a.c_in) { g_c -= a.c_in; if (++g_s == a.kw) { g_s = 0; ++g_r; } }
        typedef TIn vecI __attribute__((ext_vector_type(V)));
        typedef double vecD __attribute__((ext_vector_type(V)));
        // every load is issued unconditionally from a clamped address and masked afterwards (a load inside `if (ok)` makes hipcc
        // branch around it and wait for it alone); with V = 2 the two k share tap and pixel (c_in even)
        if (PROLOGUE) {
            if constexpr (V == 1) { g_sc[0] = pro_scale[c]; g_sh[0] = pro_shift[c]; }
            else {
                const vecD sv = *reinterpret_cast<const vecD*>(pro_scale + c), hv = *reinterpret_cast<const vecD*>(pro_shift + c);
#pragma unroll
                for (int e = 0; e < V; ++e) { g_sc[e] = sv[e]; g_sh[e] = hv[e]; }
            }
        }
        const size_t wrow = (size_t)(kvalid ? k : 0);
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            const int hi = xh[i] + r * a.dil;
            const int wi = xw[i] + sx * a.dil;
            const bool ok = kvalid && (unsigned)hi < (unsigned)a.h_in && (unsigned)wi < (unsigned)a.w_in;
            const int pix = ok ? xn[i] + hi * a.w_in + wi : 0;
            if constexpr (V == 1) xv[i][0] = in[(size_t)pix * a.in_pix_stride + c];
            else {
                const vecI t = *reinterpret_cast<const vecI*>(in + (size_t)pix * a.in_pix_stride + c);
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
                const vecD t = *reinterpret_cast<const vecD*>(w + (size_t)(wok ? co : 0) * k_total + wrow);
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
                double v = (double)xv[i][e];
                if (PROLOGUE) v = fmax(fma(v, g_sc[e], g_sh[e]), 0.0);     // pre-activation BN + ReLU (resnet_v2.py:119,229) in fp64
                xs[buf][(lrow + KR * i) * LD + kcol + e] = xok[i] ? v : 0.0;
            }
#pragma unroll
        for (int i = 0; i < WR; ++i)
#pragma unroll
            for (int e = 0; e < V; ++e) ws[buf][(lrow + KR * i) * LD + kcol + e] = wokv[i] ? wv[i][e] : 0.0;
    };

    doublex4 acc[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = doublex4{0.0, 0.0, 0.0, 0.0};

    gather(0);
    commit(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < k_total; k0 += BK) {
        gather(k0 + BK);                             // in flight under the MFMAs below (past the end: masked, clamped addresses)
        const double* xl = xs[buf] + (wave_p * 64 + (lane & 15)) * LD + (lane >> 4);
        const double* wl = ws[buf] + (wave_c * (TN / 2) + (lane & 15)) * LD + (lane >> 4);
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
            double af[4], bf[NJ];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = xl[i * 16 * LD + ks * 4];
#pragma unroll
            for (int j = 0; j < NJ; ++j) bf[j] = wl[j * 16 * LD + ks * 4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        commit(buf ^ 1);                             // the other buffer: last read before the previous barrier
        __syncthreads();
        buf ^= 1;
    }

    // epilogue: D[pixel = (lane >> 4) + 4 r][cout = lane & 15] per 16 x 16 tile: 16 lanes store 16 consecutive couts of a pixel.
    // Per 16-pixel row of tiles the 4 x NJ shortcut values are requested first (unconditionally, clamped addresses) and consumed
    // afterwards: a load inside the store loop is waited for on its own -- 64 dependent round trips per lane made the conv3 layers
    // (short K, HBM bound) 1.5-2.2 x slower than with the old 64 x 64 tiles.
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int mrow[4];
        bool mok[4];
        size_t res_pix[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int m = m0 + wave_p * 64 + i * 16 + (lane >> 4) + 4 * rr;
            mok[rr] = m < a.m_total;
            mrow[rr] = mok[rr] ? m : 0;
            res_pix[rr] = 0;
            if (residual != nullptr) {
                const int img = mrow[rr] / hw_out;
                const int rem = mrow[rr] - img * hw_out;
                const int ho = rem / a.w_out;
                const int wo = rem - ho * a.w_out;
                res_pix[rr] = (size_t)(img * a.res_h + ho * a.res_stride + a.res_offset) * a.res_w +
                              (wo * a.res_stride + a.res_offset);
            }
        }
        TAct rv[4][NJ];
        double bv[NJ];
        int cov[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int co = n0 + wave_c * (TN / 2) + j * 16 + (lane & 15);
            cov[j] = co;
            bv[j] = bias[co < a.c_out ? co : 0];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                rv[rr][j] = (TAct)0;
                if (residual != nullptr)          // wave-uniform branch
                    rv[rr][j] = residual[res_pix[rr] * a.c_out + (co < a.c_out ? co : 0)];
            }
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                double v = acc[i][j][rr] + bv[j];
                if (a.relu) v = fmax(v, 0.0);
                v += (double)rv[rr][j];
                if (mok[rr] && cov[j] < a.c_out) out[(size_t)mrow[rr] * a.c_out + cov[j]] = (TAct)v;
            }
    }
}

template <typename TIn, typename TAct>
static int launch_t(const ConvArgs& a, const MetroConvDesc& d, bool pro, const void* in, const double* w, const double* bias,
                    const double* ps, const double* pb, const void* res, void* out, hipStream_t stream) {
    // 64-cout tiles for the 64-channel layers, and wherever 128-cout tiles would leave the 256 CUs with fewer than two blocks each
    // (two co-resident blocks are what overlaps one block's gather / commit / barrier with the other's MFMAs)
    const long tiles128 = (long)((a.c_out + 127) / 128) * ((a.m_total + f64k::TMP - 1) / f64k::TMP);
    const bool small = a.c_out <= 64 || tiles128 < 512;
    const bool vec2 = d.c_in % 2 == 0 && d.in_pix_stride % 2 == 0;      // 16-byte (fp64) / 8-byte (fp32) gathers of two consecutive channels
    if (note_kernel("conv_igemm_f64acc<in%d,act%d,128x%d%s%s>%s", (int)sizeof(TIn) * 8, (int)sizeof(TAct) * 8, small ? 64 : 128, vec2 ? ",v2" : "",
                    pro ? ",pro" : "", res ? "+res" : ""))
        return METRO_OK;
    const int tn = small ? 64 : 128;
    const int tiles_c = (a.c_out + tn - 1) / tn;
    const int tiles_p = (a.m_total + f64k::TMP - 1) / f64k::TMP;
    const int lds = 2 * (f64k::TMP + tn) * f64k::LD * (int)sizeof(double);
    const dim3 grid((unsigned)(tiles_c * tiles_p)), blk(f64k::NT);
#define METRO_F64_LAUNCH(PRO, TNV, VV)                                                                                          \
    do {                                                                                                                         \
        auto kern = conv_igemm_f64acc_kernel<PRO, TIn, TAct, TNV, VV>;                                                           \
        static PerDeviceInt done;                                                                                                \
        if (const int st = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds, done, "conv_igemm_f64acc")) return st;      \
        hipLaunchKernelGGL(kern, grid, blk, lds, stream, a, static_cast<const TIn*>(in), w, bias, ps, pb,                        \
                           static_cast<const TAct*>(res), static_cast<TAct*>(out));                                              \
    } while (0)
    if (pro) {
        if (small) { if (vec2) METRO_F64_LAUNCH(true, 64, 2); else METRO_F64_LAUNCH(true, 64, 1); }
        else { if (vec2) METRO_F64_LAUNCH(true, 128, 2); else METRO_F64_LAUNCH(true, 128, 1); }
    } else {
        if (small) { if (vec2) METRO_F64_LAUNCH(false, 64, 2); else METRO_F64_LAUNCH(false, 64, 1); }
        else { if (vec2) METRO_F64_LAUNCH(false, 128, 2); else METRO_F64_LAUNCH(false, 128, 1); }
    }
#undef METRO_F64_LAUNCH
    return launch_status("conv_igemm_f64acc");
}

int launch_conv_f64acc(const MetroConvDesc& d, const void* in, const double* w, const double* bias,
                       const double* ps, const double* pb, const void* res, void* out,
                       hipStream_t stream) {
    const ConvArgs a = make_conv_args(d);
    const void* r = d.has_residual ? res : nullptr;
    const bool pro = d.has_prologue != 0;
    if (d.in_dtype == METRO_F32 && d.out_dtype == METRO_F32) return launch_t<float, float>(a, d, pro, in, w, bias, ps, pb, r, out, stream);
    if (d.in_dtype == METRO_F32 && d.out_dtype == METRO_F64) return launch_t<float, double>(a, d, pro, in, w, bias, ps, pb, r, out, stream);
    if (d.in_dtype == METRO_F64 && d.out_dtype == METRO_F64) return launch_t<double, double>(a, d, pro, in, w, bias, ps, pb, r, out, stream);
    set_error("conv_f64acc: unsupported in/out dtypes %d/%d", d.in_dtype, d.out_dtype);
    return METRO_ERR_UNSUPPORTED;
}

}  // namespace metro
