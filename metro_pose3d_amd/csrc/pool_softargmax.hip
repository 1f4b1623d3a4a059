// HBM-bound kernels of the path: input preparation, zero-padded max-pool, and the
// softmax-over-volume + expectation soft-argmax with mm decode.
#include "metro_common.h"

namespace metro {

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// fp32 NHWC [n,side,side,3] -> fp16 [n,side+6,side+8,4], zero border (3 top/left, 3/5
// bottom/right) and zero 4th channel.  Materialises the stem's explicit pad-3 (reference
// resnet_utils.py:125-135) and the fp32->fp16 cast (architectures.py:29) once, so the 7x7/2
// stem becomes a pad-free 7x1-tap implicit GEMM whose taps are 8 pixels x 4 channels = 32
// contiguous fp16 (the 8th pixel meets zero weights).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prep_input_f16_kernel(const float* __restrict__ img,
                                                             half4_t* __restrict__ out, int n,
                                                             int side) {
    const int hp = side + 6, wp = side + 8;
    const long total = (long)n * hp * wp;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < total;
         p += (long)gridDim.x * blockDim.x) {
        const int x = (int)(p % wp);
        const long t = p / wp;
        const int y = (int)(t % hp);
        const int im = (int)(t / hp);
        half4_t v = {(half_t)0, (half_t)0, (half_t)0, (half_t)0};
        const int yi = y - 3, xi = x - 3;
        if ((unsigned)yi < (unsigned)side && (unsigned)xi < (unsigned)side) {
            const float* s = img + ((size_t)(im * side + yi) * side + xi) * 3;
            v[0] = (half_t)s[0]; v[1] = (half_t)s[1]; v[2] = (half_t)s[2];
        }
        out[p] = v;
    }
}

int launch_prep_input_f16(const float* images, int n, int side, void* out, hipStream_t stream) {
    if (note_kernel("prep_input_f16")) return METRO_OK;
    const long total = (long)n * (side + 6) * (side + 8);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(prep_input_f16_kernel, dim3(blocks), dim3(256), 0, stream, images,
                       static_cast<half4_t*>(out), n, side);
    return launch_status("prep_input_f16");
}

// ---------------------------------------------------------------------------------------------
// Crop pre-processing (the step BEFORE the path): for every output pixel (x, y) of crop i, source
// coords = H_i * [x, y, 1] (fp32, perspective divide), cv2.remap(INTER_LINEAR, BORDER_CONSTANT 0) of the UINT8 HWC frame, then
// /255 and clip to [-1, 1]: reference src/cameralib.py:406-429 (reproject_image_fast) + src/improc.py:56-61 (normalize01).
// A byte path, reproduced to the byte (oracle/preprocess.py restates the rule from OpenCV's imgwarp.cpp):
//   * coordinates as NumPy's float32 matmul evaluates them: fma(h2, 1, fma(h1, y, rn(h0 * x))), IEEE divide;
//   * OpenCV's fixed point for 8-bit images: s = cvRound(coord * 32) (half to even; NaN / out of int range -> INT_MIN),
//     integer part saturate_cast<short>(s >> 5), 5-bit fractions ax, ay;
//   * 15-bit weights 32 (32 - ay)(32 - ax), 32 (32 - ay) ax, 32 ay (32 - ax), 32 ay ax -- OpenCV's table entry for ax = ay = 0 is
//     (32767, 0, 0, 1) after its sum correction; (32768, 0, 0, 0) gives the same byte for every input since |S11 - S00| < 2^14;
//   * dst = (sum of in-image taps * weights + 2^14) >> 15 (out-of-image taps are the border value 0), a uint8;
//   * normalize01: float(dst) / 255 (IEEE divide), clip.
// One thread per output pixel (3 channels), 12-byte stores.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int cv_round_x86(float v) {
    // cvtss2si: round half to even; NaN and |v| >= 2^31 give the "integer indefinite" INT_MIN
    return (fabsf(v) < 2147483648.f) ? __float2int_rn(v) : (int)0x80000000;
}

__global__ __launch_bounds__(256) void warp_crop_u8_kernel(const unsigned char* __restrict__ img, int h, int w,
                                                           int row_stride, const float* __restrict__ homs,
                                                           float* __restrict__ out, int n, int side) {
    const long total = (long)n * side * side;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
        const int x = (int)(p % side);
        const long t = p / side;
        const int y = (int)(t % side);
        const int i = (int)(t / side);
        const float* H = homs + i * 9;
        const float fx = (float)x, fy = (float)y;
        const float cx = __fmaf_rn(H[2], 1.f, __fmaf_rn(H[1], fy, __fmul_rn(H[0], fx)));
        const float cy = __fmaf_rn(H[5], 1.f, __fmaf_rn(H[4], fy, __fmul_rn(H[3], fx)));
        const float cw = __fmaf_rn(H[8], 1.f, __fmaf_rn(H[7], fy, __fmul_rn(H[6], fx)));
        const float u = __fdiv_rn(cx, cw), v = __fdiv_rn(cy, cw);
        const int sx = cv_round_x86(__fmul_rn(u, 32.f)), sy = cv_round_x86(__fmul_rn(v, 32.f));
        const int ax = sx & 31, ay = sy & 31;
        int x0 = sx >> 5, y0 = sy >> 5;
        x0 = x0 < -32768 ? -32768 : (x0 > 32767 ? 32767 : x0);          // saturate_cast<short>
        y0 = y0 < -32768 ? -32768 : (y0 > 32767 ? 32767 : y0);
        const int wgt[4] = {32 * (32 - ay) * (32 - ax), 32 * (32 - ay) * ax, 32 * ay * (32 - ax), 32 * ay * ax};
        int acc[3] = {0, 0, 0};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int xx = x0 + (k & 1), yy = y0 + (k >> 1);
            if ((unsigned)xx < (unsigned)w && (unsigned)yy < (unsigned)h) {
                const unsigned char* s = img + (size_t)yy * row_stride + (size_t)xx * 3;
                acc[0] += wgt[k] * (int)s[0]; acc[1] += wgt[k] * (int)s[1]; acc[2] += wgt[k] * (int)s[2];
            }
        }
        float* o = out + p * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int byte = (acc[c] + (1 << 14)) >> 15;                  // <= 255: the weights sum to 2^15
            o[c] = fminf(fmaxf(__fdiv_rn((float)byte, 255.f), -1.f), 1.f);
        }
    }
}

int launch_warp_crop_u8(const unsigned char* img, int h, int w, int row_stride, const float* homs, float* out,
                        int n, int side, hipStream_t stream) {
    const long total = (long)n * side * side;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(warp_crop_u8_kernel, dim3(blocks), dim3(256), 0, stream, img, h, w, row_stride, homs, out, n, side);
    return launch_status("warp_crop_u8");
}

// ---------------------------------------------------------------------------------------------
// 3x3 stride-2 max-pool over an input ZERO-padded by (1,1) (reference resnet_utils.py:177-185:
// array_ops.pad then VALID pooling -> the pad value 0 takes part in the max).
// One thread per (output pixel, 16-byte channel chunk).
// ---------------------------------------------------------------------------------------------
template <typename VecT, int VEC>
__global__ __launch_bounds__(256) void maxpool3x3s2_zeropad_kernel(const VecT* __restrict__ in,
                                                                   VecT* __restrict__ out, int n,
                                                                   int h_in, int w_in, int cvec) {
    const int h_out = (h_in + 2 - 3) / 2 + 1, w_out = (w_in + 2 - 3) / 2 + 1;
    const long total = (long)n * h_out * w_out * cvec;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int cv = (int)(idx % cvec);
        long t = idx / cvec;
        const int wo = (int)(t % w_out);
        t /= w_out;
        const int ho = (int)(t % h_out);
        const int im = (int)(t / h_out);
        VecT best;
        bool first = true;
        bool touched_pad = false;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int hi = 2 * ho - 1 + r;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int wi = 2 * wo - 1 + s;
                if ((unsigned)hi < (unsigned)h_in && (unsigned)wi < (unsigned)w_in) {
                    const VecT v = in[((size_t)(im * h_in + hi) * w_in + wi) * cvec + cv];
                    if (first) { best = v; first = false; }
                    else {
#pragma unroll
                        for (int e = 0; e < VEC; ++e) best[e] = v[e] > best[e] ? v[e] : best[e];
                    }
                } else {
                    touched_pad = true;
                }
            }
        }
        if (touched_pad) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) best[e] = best[e] > 0 ? best[e] : 0;
        }
        out[idx] = best;
    }
}

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef double doublex2 __attribute__((ext_vector_type(2)));

int launch_maxpool(const void* in, void* out, int n, int h_in, int w_in, int c, int dtype,
                   hipStream_t stream) {
    const int h_out = (h_in + 2 - 3) / 2 + 1, w_out = (w_in + 2 - 3) / 2 + 1;
    if (dtype != METRO_F16 && dtype != METRO_F32 && dtype != METRO_F64) { set_error("maxpool: unsupported dtype %d", dtype); return METRO_ERR_INVALID_ARG; }
    if (note_kernel("maxpool3x3s2_zeropad<%s>", dtype == METRO_F16 ? "f16" : dtype == METRO_F32 ? "f32" : "f64")) return METRO_OK;
    if (dtype == METRO_F16) {
        if (c % 8) { set_error("maxpool f16: channels %d not a multiple of 8", c); return METRO_ERR_INVALID_ARG; }
        const long total = (long)n * h_out * w_out * (c / 8);
        const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
        hipLaunchKernelGGL((maxpool3x3s2_zeropad_kernel<half8_t, 8>), dim3(blocks), dim3(256), 0, stream,
                           static_cast<const half8_t*>(in), static_cast<half8_t*>(out), n, h_in, w_in, c / 8);
    } else if (dtype == METRO_F32) {
        if (c % 4) { set_error("maxpool f32: channels %d not a multiple of 4", c); return METRO_ERR_INVALID_ARG; }
        const long total = (long)n * h_out * w_out * (c / 4);
        const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
        hipLaunchKernelGGL((maxpool3x3s2_zeropad_kernel<floatx4, 4>), dim3(blocks), dim3(256), 0, stream,
                           static_cast<const floatx4*>(in), static_cast<floatx4*>(out), n, h_in, w_in, c / 4);
    } else if (dtype == METRO_F64) {
        if (c % 2) { set_error("maxpool f64: channels %d not a multiple of 2", c); return METRO_ERR_INVALID_ARG; }
        const long total = (long)n * h_out * w_out * (c / 2);
        const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
        hipLaunchKernelGGL((maxpool3x3s2_zeropad_kernel<doublex2, 2>), dim3(blocks), dim3(256), 0, stream,
                           static_cast<const doublex2*>(in), static_cast<doublex2*>(out), n, h_in, w_in, c / 2);
    } else {
        set_error("maxpool: unsupported dtype %d", dtype);
        return METRO_ERR_INVALID_ARG;
    }
    return launch_status("maxpool3x3s2_zeropad");
}

// ---------------------------------------------------------------------------------------------
// Soft-argmax.  logits fp32 NHWC [n, S, S, C = D*J], channel c = d*J + j (reference
// volumetric.py:230-232).  Per (image, joint): softmax over all S*S*D voxels (tfu.py:466-471),
// then expectation of linspace(0,1,.) coordinates along W (x), H (y), D (z) (tfu.py:474-499 with
// axes [3,2,4], volumetric.py:234).
//
// Kernel 1 (partials): grid (slabs, n).  A slab is a contiguous range of pixels; since NHWC
// keeps a pixel's C channels contiguous, the slab is one contiguous float range read with
// 16-byte loads.  Thread (pp, q) owns the channel quad q (4 consecutive channels) and walks
// pixels pp, pp+PPB, ... keeping an ONLINE softmax state (running max m, sum e, sum e*x,
// sum e*y) per channel: logits are read exactly once.  The block then folds the PPB pixel
// lanes and the D depth channels of each joint through LDS and emits
// (m, S, Sx, Sy, Sz) per (image, slab, joint).
// Kernel 2 (finalize): folds the slabs, divides, decodes to millimetres
// (volumetric.py:288-295,303-306), subtracts the root = last head joint (tfu3d.py:23-25) and
// gathers the exported joint order (main.py:119-127).
// ---------------------------------------------------------------------------------------------
constexpr int SA_NT = 256;

int softargmax_slabs(int n, int side) {
    // Slabs of >= 32 pixels, at most 64 per image.  Deliberately independent of the batch size: the fp32 partial
    // sums are folded slab by slab, so a crop's pose has the same bits whatever batch it arrives in (no op of the
    // path crosses the batch dimension).  Batch 64 at stride 16: 8 slabs x 64 images = 512 blocks.
    (void)n;
    const int pixels = side * side;
    int slabs = pixels / 32;
    if (slabs > 64) slabs = 64;
    if (slabs < 1) slabs = 1;
    return slabs;
}

int64_t softargmax_scratch_bytes(int n, int side, int n_joints_head) {
    return (int64_t)n * softargmax_slabs(n, side) * n_joints_head * 5 * sizeof(double);
}

template <typename AccT>
__device__ __forceinline__ AccT acc_exp(AccT x);
template <> __device__ __forceinline__ float acc_exp<float>(float x) { return __expf(x); }
template <> __device__ __forceinline__ double acc_exp<double>(double x) { return exp(x); }

template <typename AccT, typename LogitT>
__global__ __launch_bounds__(SA_NT) void softargmax_partial_kernel(
    const LogitT* __restrict__ logits, AccT* __restrict__ partials, int side, int depth, int nj,
    int slabs) {
    extern __shared__ __attribute__((aligned(16))) char sa_smem[];
    AccT* red = reinterpret_cast<AccT*>(sa_smem);   // [ppb][C][4]: m, s, sx, sy

    const int C = depth * nj;
    const int quads = C / 4;
    const int ppb = SA_NT / quads;            // pixel lanes per block
    const int tid = threadIdx.x;
    const int pp = tid / quads;
    const int q = tid - pp * quads;
    const int img = blockIdx.y;
    const int slab = blockIdx.x;
    const int pixels = side * side;
    const int p_begin = (int)((long)pixels * slab / slabs);
    const int p_end = (int)((long)pixels * (slab + 1) / slabs);
    const float step_s = 1.0f / (float)(side - 1);   // tf.linspace step in fp32 (tfu.py:481)

    AccT m[4], s[4], sx[4], sy[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { m[e] = (AccT)-INFINITY; s[e] = 0; sx[e] = 0; sy[e] = 0; }

    if (pp < ppb) {
        const LogitT* base = logits + (size_t)img * pixels * C + q * 4;
        typedef LogitT logit4 __attribute__((ext_vector_type(4)));
        // The path is HBM bound and every pixel lane has ONE 16-byte load per pixel: UNR pixels are requested before the first
        // is consumed -- with one load in flight per thread the launch ran at 3.3 TB/s at the configs[4] volume (8 blocks x 238
        // lanes x 16 B = 30 KB in flight per CU).  The online softmax is BRANCH FREE: per channel the running maximum is raised to
        // the maximum of the UNR new logits first (one rescale exp per channel and iteration; exp(0) = 1 exactly when it did not
        // move), pixels past the slab enter as -inf and contribute exp(-inf) = 0; a per-element `if (x > m)` cost a compare, an
        // exec-mask save / restore and a branch per logit.  Pixel coordinates advance incrementally (no division in the loop).
#ifndef METRO_SA_UNR
#define METRO_SA_UNR 2     // measured at the configs[4] volume (285 MB): 2 -> 67.6 us, 4 -> 69.3, 8 -> 88.7 (registers cut the occupancy); 1 (round 3) -> 87
#endif
        constexpr int UNR = METRO_SA_UNR;
        int ph = (p_begin + pp) / side, pw = (p_begin + pp) - ph * side;       // pixel of this lane, advanced by ppb per slot
        for (int p0 = p_begin + pp; p0 < p_end; p0 += UNR * ppb) {
            logit4 v[UNR];
            AccT cx[UNR], cy[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int p = p0 + u * ppb;
                const bool ok = p < p_end;
                v[u] = *reinterpret_cast<const logit4*>(base + (size_t)(ok ? p : p0) * C);
                cx[u] = (AccT)((float)pw * step_s);
                cy[u] = (AccT)((float)ph * step_s);
                if (!ok) { v[u][0] = v[u][1] = v[u][2] = v[u][3] = (LogitT)-INFINITY; }
                pw += ppb;
                while (pw >= side) { pw -= side; ++ph; }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                AccT mx = m[e];
#pragma unroll
                for (int u = 0; u < UNR; ++u) mx = (AccT)v[u][e] > mx ? (AccT)v[u][e] : mx;
                const AccT f = acc_exp<AccT>(m[e] - mx);          // exp(-inf) = 0 on first touch
                AccT s_ = s[e] * f, sx_ = sx[e] * f, sy_ = sy[e] * f;
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const AccT ex = acc_exp<AccT>((AccT)v[u][e] - mx);
                    s_ += ex; sx_ += ex * cx[u]; sy_ += ex * cy[u];
                }
                m[e] = mx; s[e] = s_; sx[e] = sx_; sy[e] = sy_;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            AccT* r = red + ((size_t)pp * C + q * 4 + e) * 4;
            r[0] = m[e]; r[1] = s[e]; r[2] = sx[e]; r[3] = sy[e];
        }
    }
    __syncthreads();

    // fold, stage 1: one thread per CHANNEL folds the pixel lanes (a serial walk of one thread per joint over ppb * depth = 56
    // records was the tail of every block); the result replaces lane 0's record
    for (int c = tid; c < C; c += SA_NT) {               // C > 256 for the 53-joint head
        AccT M = (AccT)-INFINITY;
        for (int l = 0; l < ppb; ++l) {
            const AccT mv = red[((size_t)l * C + c) * 4];
            M = mv > M ? mv : M;
        }
        AccT S = 0, SX = 0, SY = 0;
        for (int l = 0; l < ppb; ++l) {
            const AccT* r = red + ((size_t)l * C + c) * 4;
            if (r[1] > 0) {
                const AccT f = acc_exp<AccT>(r[0] - M);
                S += r[1] * f; SX += r[2] * f; SY += r[3] * f;
            } else if (r[1] != r[1]) S = r[1];           // a NaN sum must reach the finalize launch's non-finite screen
        }
        AccT* r0 = red + (size_t)c * 4;                  // lane 0's record of this channel: read above by this thread only
        r0[0] = M; r0[1] = S; r0[2] = SX; r0[3] = SY;
    }
    __syncthreads();
    // stage 2: one thread per joint folds its `depth` channels (c = d * nj + j, volumetric.py:231)
    if (tid < nj) {
        const int j = tid;
        const float step_d = 1.0f / (float)(depth - 1);
        AccT M = (AccT)-INFINITY;
        for (int d = 0; d < depth; ++d) {
            const AccT mv = red[(size_t)(d * nj + j) * 4];
            M = mv > M ? mv : M;
        }
        AccT S = 0, SX = 0, SY = 0, SZ = 0;
        for (int d = 0; d < depth; ++d) {
            const AccT* r = red + (size_t)(d * nj + j) * 4;
            if (r[1] > 0) {
                const AccT f = acc_exp<AccT>(r[0] - M);
                const AccT cz = (AccT)((float)d * step_d);
                S += r[1] * f; SX += r[2] * f; SY += r[3] * f; SZ += r[1] * f * cz;
            } else if (r[1] != r[1]) S = r[1];
        }
        AccT* o = partials + (((size_t)img * slabs + slab) * nj + j) * 5;
        o[0] = M; o[1] = S; o[2] = SX; o[3] = SY; o[4] = SZ;
    }
}

// One block of four waves per image; wave w folds joints w, w + 4, ...: its lanes stride over the slabs (up to 128 records per
// joint behind the 256-pixel head), maximum and the four rescaled sums folded across the wave with xor shuffles in a fixed
// order -- a serial loop over the slabs by ONE thread per joint cost 20 us on a 64 x 64 heat map.
template <typename AccT>
__device__ __forceinline__ AccT sa_wave_sum(AccT v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <typename AccT>
__global__ __launch_bounds__(1024) void softargmax_finalize_kernel(const AccT* __restrict__ partials,
                                                                  float* __restrict__ poses,
                                                                  SoftArgmaxArgs a, int slabs,
                                                                  float* __restrict__ coords01,
                                                                  int32_t* __restrict__ status) {
    __shared__ AccT mm[METRO_MAX_JOINTS][3];
    // Non-finite screen (status[img] = 1): a record whose sum is NaN, a non-finite maximum or normaliser.  fp16 storage overflows
    // at 65 504: an Inf in the residual stream reaches every logit of its pixel (reference tfu.py:426-440 keeps fp32 variables
    // for the same reason); `r[1] > 0` alone would silently DROP a NaN record and return a finite, wrong pose.
    __shared__ int s_bad;
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    bool bad = false;
    const int img = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nj = a.n_joints_head;
    if (slabs <= 16) {
        // few records per joint (stride-16 / 32 heads: 4 or 1): one thread per joint walks them -- the shuffle folds of the wave
        // form cost more than they save here (8.3 vs 4.6 us at batch 64)
        const int j = threadIdx.x;
        if (j < nj) {
            // every field of every record is loaded unconditionally, eight records' loads in flight: as a chain of dependent loads
            // (a record at a time, the sum fields behind the test of the normaliser) eight records cost 8.5 us
            AccT M = (AccT)-INFINITY;
#pragma unroll 8
            for (int sl = 0; sl < slabs; ++sl) {
                const AccT mv = partials[(((size_t)img * slabs + sl) * nj + j) * 5];
                M = mv > M ? mv : M;
            }
            AccT S = 0, SX = 0, SY = 0, SZ = 0;
#pragma unroll 8
            for (int sl = 0; sl < slabs; ++sl) {
                const AccT* r = partials + (((size_t)img * slabs + sl) * nj + j) * 5;
                const AccT r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3], r4 = r[4];
                if (r1 > 0) {
                    const AccT f = acc_exp<AccT>(r0 - M);
                    S += r1 * f; SX += r2 * f; SY += r3 * f; SZ += r4 * f;
                } else bad = true;
            }
            bad = bad || !(M - M == (AccT)0) || !(S - S == (AccT)0) || !(SX - SX == (AccT)0) || !(SY - SY == (AccT)0) || !(SZ - SZ == (AccT)0);
            const AccT x01 = SX / S, y01 = SY / S, z01 = SZ / S;
            if (coords01 != nullptr) {
                float* c = coords01 + ((size_t)img * nj + j) * 3;
                c[0] = (float)x01; c[1] = (float)y01; c[2] = (float)z01;
            }
            mm[j][0] = (x01 * (AccT)a.lrc + (AccT)a.half_off) * (AccT)a.box_size_mm / (AccT)a.proc_side;
            mm[j][1] = (y01 * (AccT)a.lrc + (AccT)a.half_off) * (AccT)a.box_size_mm / (AccT)a.proc_side;
            mm[j][2] = z01 * (AccT)a.box_size_mm;
        }
    } else
    for (int j = wave; j < nj; j += (int)(blockDim.x >> 6)) {      // many records per joint: a wave per joint (16 waves: one or two rounds)
        AccT M = (AccT)-INFINITY;
        for (int sl = lane; sl < slabs; sl += 64) {
            const AccT mv = partials[(((size_t)img * slabs + sl) * nj + j) * 5];
            M = mv > M ? mv : M;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const AccT other = __shfl_xor(M, o, 64);
            M = other > M ? other : M;
        }
        AccT S = 0, SX = 0, SY = 0, SZ = 0;
#pragma unroll 2
        for (int sl = lane; sl < slabs; sl += 64) {
            const AccT* r = partials + (((size_t)img * slabs + sl) * nj + j) * 5;
            const AccT r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3], r4 = r[4];      // all five before the test: one latency, not two
            if (r1 > 0) {
                const AccT f = acc_exp<AccT>(r0 - M);
                S += r1 * f; SX += r2 * f; SY += r3 * f; SZ += r4 * f;
            } else bad = true;
        }
        S = sa_wave_sum(S); SX = sa_wave_sum(SX); SY = sa_wave_sum(SY); SZ = sa_wave_sum(SZ);
        bad = bad || !(M - M == (AccT)0) || !(S - S == (AccT)0) || !(SX - SX == (AccT)0) || !(SY - SY == (AccT)0) || !(SZ - SZ == (AccT)0);
        if (lane == 0) {
            const AccT x01 = SX / S, y01 = SY / S, z01 = SZ / S;
            if (coords01 != nullptr) {            // net_output_to_heatmap_and_coords output (volumetric.py:234-235)
                float* c = coords01 + ((size_t)img * nj + j) * 3;
                c[0] = (float)x01; c[1] = (float)y01; c[2] = (float)z01;
            }
            // heatmap_to_metric: (c * lrc + half) * box / proc_side ; z * box  (volumetric.py:288-306)
            mm[j][0] = (x01 * (AccT)a.lrc + (AccT)a.half_off) * (AccT)a.box_size_mm / (AccT)a.proc_side;
            mm[j][1] = (y01 * (AccT)a.lrc + (AccT)a.half_off) * (AccT)a.box_size_mm / (AccT)a.proc_side;
            mm[j][2] = z01 * (AccT)a.box_size_mm;
        }
    }
    if (bad) atomicOr(&s_bad, 1);
    __syncthreads();
    if (status != nullptr && threadIdx.x == 0) status[img] = s_bad;
    const int j = threadIdx.x;
    if (poses != nullptr && j < a.n_joints_out) {
        const int src = a.perm[j];
        float* o = poses + ((size_t)img * a.n_joints_out + j) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = (float)(mm[src][c] - mm[nj - 1][c]);   // tfu3d.py:23-25
    }
}

SoftArgmaxArgs make_softargmax_args(const MetroSpec& spec, int n) {
    SoftArgmaxArgs a;
    a.n = n;
    a.side = spec.proc_side / spec.stride;
    a.depth = spec.depth;
    a.n_joints_head = spec.n_joints_head;
    a.n_joints_out = spec.n_joints_out;
    const int last = spec.proc_side - 1;
    a.lrc = last - (last % spec.stride) - 1;                        // volumetric.py:290-291
    a.half_off = spec.centered_stride ? spec.stride / 2 : 0;             // volumetric.py:293-294
    a.box_size_mm = spec.box_size_mm;
    a.proc_side = spec.proc_side;
    for (int i = 0; i < METRO_MAX_JOINTS; ++i) a.perm[i] = i < spec.n_joints_out ? spec.permutation[i] : 0;
    return a;
}

template <typename AccT, typename LogitT>
static int launch_softargmax_t(const void* logits, const SoftArgmaxArgs& a, void* partials,
                               float* poses, float* coords01, hipStream_t stream, int32_t* status) {
    const int C = a.depth * a.n_joints_head;
    const int quads = C / 4;
    const int ppb = SA_NT / quads;
    const int slabs = softargmax_slabs(a.n, a.side);
    const size_t lds = (size_t)ppb * C * 4 * sizeof(AccT);
    if (note_kernel("softargmax_partial<acc%d,logits%d> & softargmax_finalize<acc%d>", (int)sizeof(AccT) * 8, (int)sizeof(LogitT) * 8,
                    (int)sizeof(AccT) * 8))
        return METRO_OK;
    auto kern = softargmax_partial_kernel<AccT, LogitT>;
    if (lds > 64 * 1024) {
        static PerDeviceInt attr_done;
        if (const int st = ensure_dyn_lds(reinterpret_cast<const void*>(kern), 160 * 1024, attr_done, "softargmax_partial")) return st;
    }
    hipLaunchKernelGGL(kern, dim3(slabs, a.n), dim3(SA_NT), lds, stream, static_cast<const LogitT*>(logits),
                       static_cast<AccT*>(partials), a.side, a.depth, a.n_joints_head, slabs);
    int st = launch_status("softargmax_partial");
    if (st) return st;
    hipLaunchKernelGGL(softargmax_finalize_kernel<AccT>, dim3(a.n), dim3(slabs > 16 ? 1024 : 256), 0, stream,
                       static_cast<const AccT*>(partials), poses, a, slabs, coords01, status);
    return launch_status("softargmax_finalize");
}

int launch_softargmax_finalize(const float* partials, const SoftArgmaxArgs& a, int slabs, float* poses_out,
                               hipStream_t stream, float* coords01_out, int32_t* status) {
    if (note_kernel("softargmax_finalize<acc32>")) return METRO_OK;
    hipLaunchKernelGGL(softargmax_finalize_kernel<float>, dim3(a.n), dim3(slabs > 16 ? 1024 : 256), 0, stream, partials, poses_out, a, slabs, coords01_out, status);
    return launch_status("softargmax_finalize");
}

int launch_softargmax(const void* logits, const SoftArgmaxArgs& a, int precise, void* partials,
                      float* poses_out, hipStream_t stream, float* coords01_out, int32_t* status) {
    const int C = a.depth * a.n_joints_head;
    if (C % 4 || C / 4 > SA_NT || a.n_joints_head > METRO_MAX_JOINTS || a.n_joints_out > 64) {
        set_error("softargmax: unsupported head (depth %d, joints %d)", a.depth, a.n_joints_head);
        return METRO_ERR_UNSUPPORTED;
    }
    if (precise == 0) return launch_softargmax_t<float, float>(logits, a, partials, poses_out, coords01_out, stream, status);
    if (precise == 1) return launch_softargmax_t<double, float>(logits, a, partials, poses_out, coords01_out, stream, status);
    if (precise == 2) return launch_softargmax_t<double, double>(logits, a, partials, poses_out, coords01_out, stream, status);
    set_error("softargmax: precise must be 0, 1 or 2 (got %d)", precise);
    return METRO_ERR_INVALID_ARG;
}

}  // namespace metro
