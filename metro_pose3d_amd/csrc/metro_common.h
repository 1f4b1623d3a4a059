// Internal helpers shared by the HIP translation units of libmetro_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "../../include/metro_hip.h"

namespace metro {

// thread-local last-error message (metro_last_error)
void set_error(const char* fmt, ...);
const char* get_error();

#define METRO_CHECK_ARG(cond, ...)                  \
    do {                                            \
        if (!(cond)) {                              \
            ::metro::set_error(__VA_ARGS__);        \
            return METRO_ERR_INVALID_ARG;           \
        }                                           \
    } while (0)

#define METRO_HIP_CHECK(expr)                                                              \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            ::metro::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),      \
                               __FILE__, __LINE__);                                        \
            return METRO_ERR_HIP;                                                          \
        }                                                                                  \
    } while (0)

// Tuning knobs: compile-time constants in the product build.  A build with -DMETRO_TUNING_KNOBS (tools/
// build_dbg_variants.sh, A/B timing runs) reads them from the environment instead; the product never calls getenv.
inline int tuning_knob(const char* name, int dflt) {
#ifdef METRO_TUNING_KNOBS
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

// ---- dispatch notes ----------------------------------------------------------------------------------------
// Which kernel INSTANTIATION a layer runs on depends on its shape and on the batch (tile counts vs 256 CUs).  Every
// leaf launcher calls note_kernel() FIRST -- before any HIP call -- with the id of the instantiation it is about to
// launch.  Off (the default) it is one thread-local load; mode 1 appends the id to a thread-local string
// (metro_last_kernel_id); mode 2 is a DRY RUN: the id is recorded and the launcher returns METRO_OK without touching
// the device (metro_plan_layer_kernel, and the coverage test in tests/test_kernel_coverage.py, work without a GPU).
struct KernelNotes { int mode; char ids[1024]; };   // a string that did not fit ends in " ..." (tests assert against it)
KernelNotes& kernel_notes();
bool note_kernel(const char* fmt, ...) __attribute__((format(printf, 1, 2)));   // true = dry run: skip the launch

// 16-byte activation store of an epilogue.  METRO_NT_STORES (A/B builds only) marks them non-temporal:
//   1 = the streaming kernels whose outputs exceed the L2 (stem, conv_pw64, conv3x3_c64), 2 = every conv kernel.
typedef unsigned int metro_u32x4 __attribute__((ext_vector_type(4)));
template <int LEVEL = 1>
__device__ __forceinline__ void store_out16(void* p, uint4 v) {
#if defined(METRO_NT_STORES)
    if constexpr (METRO_NT_STORES >= LEVEL) {
        metro_u32x4 w = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(w, reinterpret_cast<metro_u32x4*>(p));
        return;
    }
#endif
    *reinterpret_cast<uint4*>(p) = v;
}

// ---- launch-resident weight fragments of the persistent (weight-stationary) kernels -------------------------------------------
// A wave keeps `rows32` x 32 weight rows [.][K] as v_mfma_f32_32x32x16_f16 A fragments: lane (r = lane & 31, h = lane >> 5) holds
// W[row][16 kk + 8 h .. + 7] for every k step kk.  Loading them straight from global memory asks the L2 for 32 rows x 32 bytes per
// wave instruction -- and the L2s answer a near-constant REQUEST rate (~120 G requests/s chip-wide, whatever the size):
// 256 blocks x 8 waves x K/4 instructions x 32 pieces is 2.1 M requests = 16-17 us at K = 512 (8 us at K = 256) before the first
// MFMA of the launch, measured as the batch-independent part of conv_pws / conv_pw64 (round 5).  Staged form: the wave copies 32
// rows x 256 bytes per step with 8 loads of 4 x 256 contiguous bytes (8x fewer, 8x larger requests) into a PRIVATE 8.5 KiB LDS
// scratch (rows padded to 272 bytes: conflict-free 16-byte reads of 32 rows) and reads its fragments back.  No barrier: LDS
// executes a wave's instructions in order; wave_barrier() only keeps hipcc from moving the reads above the writes.
constexpr int W_STAGE_ROW = 272;
constexpr int W_STAGE_BYTES = 32 * W_STAGE_ROW;            // per wave
template <int K, typename Frag>
__device__ __forceinline__ void load_w_frags_staged(const _Float16* __restrict__ rows /* the wave's first row */, Frag* wf /* [K / 16] */,
                                                    char* scratch /* wave-private, W_STAGE_BYTES */, int lane) {
    static_assert(K % 128 == 0 || K == 64, "whole 256-byte row pieces (K = 64: one 128-byte piece)");
    constexpr int PIECE = K >= 128 ? 128 : 64;             // channels per staged step
    constexpr int LPR = PIECE / 8;                         // lanes per row: 16 (or 8)
    constexpr int RPI = 64 / LPR;                          // rows per load instruction: 4 (or 8)
    const int lr = lane / LPR, lc = lane % LPR;
    const int fr = lane & 31, fh = lane >> 5;
#pragma unroll
    for (int c = 0; c < K / PIECE; ++c) {
#pragma unroll
        for (int j = 0; j < 32 / RPI; ++j)
            *reinterpret_cast<uint4*>(scratch + (j * RPI + lr) * W_STAGE_ROW + lc * 16) =
                *reinterpret_cast<const uint4*>(rows + (size_t)(j * RPI + lr) * K + c * PIECE + lc * 8);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k8 = 0; k8 < PIECE / 16; ++k8)
            wf[c * (PIECE / 16) + k8] = *reinterpret_cast<const Frag*>(scratch + fr * W_STAGE_ROW + k8 * 32 + fh * 16);
        __builtin_amdgcn_wave_barrier();
    }
}

// The same for v_mfma_f32_16x16x32_f16 A fragments: 16 weight rows [.][K], lane (r = lane & 15, g = lane >> 4) holds
// W[row][32 ks + 8 g .. + 7] for every k step ks (K / 32 registers of 16 bytes).  Staged 128 channels (256-byte row pieces) at a time.
template <int K, typename Frag>
__device__ __forceinline__ void load_w_frags16_staged(const _Float16* __restrict__ rows, Frag* wf /* [K / 32] */, char* scratch, int lane) {
    static_assert(K % 128 == 0, "whole 256-byte row pieces");
    const int lr = lane >> 4, lc = lane & 15;
#pragma unroll
    for (int c = 0; c < K / 128; ++c) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<uint4*>(scratch + (j * 4 + lr) * W_STAGE_ROW + lc * 16) =
                *reinterpret_cast<const uint4*>(rows + (size_t)(j * 4 + lr) * K + c * 128 + lc * 8);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4)
            wf[c * 4 + k4] = *reinterpret_cast<const Frag*>(scratch + (lane & 15) * W_STAGE_ROW + k4 * 64 + (lane >> 4) * 16);
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- per-device one-time kernel setup -----------------------------------------------------------------------
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) and the occupancy query act on the CURRENT device's copy of a
// kernel: a process that drives several GPUs (inference.py caches one Engine per device) must do them once per
// device, not once per process.  Zero-initialised statics of this type replace `static bool attr_set`.
constexpr int METRO_MAX_DEVICES = 64;
struct PerDeviceInt { int v[METRO_MAX_DEVICES]; };

inline int current_device_slot(int* slot) {
    int dev = 0;
    const hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= METRO_MAX_DEVICES) {
        set_error("hipGetDevice: %s (device %d; at most %d devices per process)", hipGetErrorString(e), dev, METRO_MAX_DEVICES);
        return METRO_ERR_HIP;
    }
    *slot = dev;
    return METRO_OK;
}

// opts a kernel in to `bytes` of dynamic LDS on the current device (once per device)
inline int ensure_dyn_lds(const void* kern, int bytes, PerDeviceInt& done, const char* what) {
    int slot = 0;
    const int st = current_device_slot(&slot);
    if (st) return st;
    if (done.v[slot]) return METRO_OK;
    const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        set_error("hipFuncSetAttribute(%s, %d B of LDS, device %d): %s", what, bytes, slot, hipGetErrorString(e));
        return METRO_ERR_HIP;
    }
    done.v[slot] = 1;
    return METRO_OK;
}

// same, and returns CUs x resident blocks per CU of the kernel on the current device (persistent kernels' grid cap)
inline int ensure_dyn_lds_and_grid_cap(const void* kern, int threads, int bytes, PerDeviceInt& cap, const char* what,
                                       int max_blocks_per_cu, int* grid_cap) {
    int slot = 0;
    int st = current_device_slot(&slot);
    if (st) return st;
    if (cap.v[slot] == 0) {
        const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(%s, %d B of LDS, device %d): %s", what, bytes, slot, hipGetErrorString(e));
            return METRO_ERR_HIP;
        }
        int cus = 0, occ = 0;
        METRO_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, slot));
        if (const int lim = tuning_knob("METRO_CU_LIMIT", 0); lim > 0 && lim < cus) cus = lim;   // A/B builds: partitioned-GPU experiments
        METRO_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, threads, bytes));
        if (occ < 1) occ = 1;
        if (max_blocks_per_cu > 0 && occ > max_blocks_per_cu) occ = max_blocks_per_cu;
        cap.v[slot] = cus * occ;
    }
    *grid_cap = cap.v[slot];
    return METRO_OK;
}

inline int launch_status(const char* what) {
    if (kernel_notes().mode == 2) {     // a leaf launcher that did not call note_kernel() before touching the device
        set_error("internal: %s was launched during a dry run (note_kernel must come first)", what);
        return METRO_ERR_STATE;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("launch of %s failed: %s", what, hipGetErrorString(e));
        return METRO_ERR_HIP;
    }
    return METRO_OK;
}

// device-side view of MetroConvDesc plus derived values
// Two convolutions over the same input fused into one launch: weight/bias rows [0, split) belong
// to the first output (desc.c_out == split + c_out2 rows in total), rows [split, ...) to `out2`
// (NHWC with c_out2 channels, its own ReLU flag).  split must be a multiple of the cout tile.
struct ConvSplit {
    int split = 0;
    int c_out2 = 0;
    int relu2 = 0;
    void* out2 = nullptr;
};

// conv3 of a unit + conv1 of the NEXT unit in one launch (block1: full 256-channel rows per pixel tile):
// after the shortcut add, t1 = relu(W2 * relu(x_out * scale2 + shift2) + bias2) is computed from the
// LDS-resident output tile (reference resnet_v2.py:119,127-128 of unit u+1 on the output of :138 of unit u).
struct ConvFuse2 {
    const void* w2 = nullptr;       // fp16 [c2][c_out], BN-folded
    const float* bias2 = nullptr;   // [c2]
    const void* scale2 = nullptr;   // fp16 [c_out]  (pre-activation BN of unit u+1)
    const void* shift2 = nullptr;
    void* out2 = nullptr;           // fp16 NHWC [.., c2]
    int c2 = 0;
};

// conv1 of the SAME unit in front of a 64 -> 64 3x3 (conv3x3_c64.hip PRE1; block1/unit_1, whose input has 64 channels):
// t1 = relu(W1 * relu(x * scale + shift) + bias1) is computed on the LDS-resident slab (reference resnet_v2.py:119,127-128)
struct ConvPre1 {
    const void* w1 = nullptr;         // fp16 [64][64], BN-folded
    const float* bias1 = nullptr;     // [64]
    const void* pro_scale = nullptr;  // fp16 [64] pre-activation BN of the unit
    const void* pro_shift = nullptr;
    void* t1_dump = nullptr;          // optional: conv1's output fp16 [pixels][64] (layer dumps for the tests; NULL in the product path)
};

// Projection shortcut of the unit computed INSIDE its conv3 launch (conv_pw64.hip PSC; block1/unit_1): shortcut =
// fp16(Wsc * relu(x * scale + shift) + bias_sc) from the unit's 64-channel input x (reference resnet_v2.py:119,122-125),
// added to fp16(conv3 + bias) in fp16 like the reference graph (:138) -- the shortcut tensor never exists in HBM
struct ConvProjSc {
    const void* x = nullptr;          // fp16 [pixels][64]: the unit's raw input
    const void* w_sc = nullptr;       // fp16 [256][64]
    const float* bias_sc = nullptr;   // [256]
    const void* pro_scale = nullptr;  // fp16 [64]
    const void* pro_shift = nullptr;
};

// block1 without its 256-channel residual stream in HBM (conv_pw64.hip REB / OUTM, round 5).  The launch conv3(u) + conv1(u+1):
//   t2_prev != NULL  the unit's identity shortcut x_{u-1} is REBUILT in the launch from the previous unit's conv2 output and conv3
//                    parameters + the projection shortcut of ConvProjSc (x_{u-1} = fp16(W3_prev . t2_prev + b) + fp16(Wsc . pre(x0) + bsc)),
//                    reference resnet_v2.py:119-125,134-138 of unit u-1, instead of being read as a 512-byte-per-pixel tensor;
//   out_mode         0 = the launch's sum is stored in full, 1 = not at all (it only feeds the next unit's conv1 in the launch),
//                    2 = only the pixels (sub_off + 2 i, sub_off + 2 j) the next, strided unit's shortcut reads (resnet_v2.py:113-121;
//                    resnet_utils.py:64-79), as a compact [n, h_sub, w_sub, 256] tensor.
struct ConvRebuild {
    const void* t2_prev = nullptr;     // fp16 [pixels][64]
    const void* w3_prev = nullptr;     // fp16 [256][64]
    const float* bias3_prev = nullptr; // [256]
    int out_mode = 0;
    void* out_sub = nullptr;
    int sub_off = 0, h_sub = 0, w_sub = 0;
    int classic = 0;                   // 1 = the single-role kernel of conv_pw64.hip even where conv_b1.hip's producer / consumer form
                                       // would be dispatched (metro_forward_upto stopping at the layer: an independent second form)
};

struct ConvArgs {
    int split, c_out2, relu2;   // see ConvSplit (0 = plain convolution)
    int n, h_in, w_in, c_in, in_pix_stride;
    int h_out, w_out, c_out;
    int kh, kw, stride, dil, pad_top, pad_left;
    int res_h, res_w, res_stride, res_offset;
    int m_total;   // n * h_out * w_out
    int relu;
};

inline ConvArgs make_conv_args(const MetroConvDesc& d) {
    ConvArgs a;
    a.split = 0; a.c_out2 = 0; a.relu2 = 0;
    a.n = d.n; a.h_in = d.h_in; a.w_in = d.w_in; a.c_in = d.c_in; a.in_pix_stride = d.in_pix_stride;
    a.h_out = d.h_out; a.w_out = d.w_out; a.c_out = d.c_out;
    a.kh = d.kh; a.kw = d.kw; a.stride = d.stride; a.dil = d.dilation;
    a.pad_top = d.pad_top; a.pad_left = d.pad_left;
    a.res_h = d.res_h; a.res_w = d.res_w; a.res_stride = d.res_stride; a.res_offset = d.res_offset;
    a.m_total = d.n * d.h_out * d.w_out;
    a.relu = d.relu;
    return a;
}

int validate_conv_desc(const MetroConvDesc* d);

// kernel launchers implemented in the .hip files
// fp16 convolution dispatcher: tap-reuse slab kernel for 3x3 stride-1 layers, the LDS-DMA ring kernel otherwise
int launch_conv_f16(const MetroConvDesc& d, const void* in, const void* w, const float* bias,
                    const void* pro_scale, const void* pro_shift, const void* residual, void* out,
                    hipStream_t stream);
// LDS-DMA ring kernel (coalesced epilogue)
bool conv_f16_dma_supported(const MetroConvDesc& d);
int launch_conv_f16_dma(const MetroConvDesc& d, const void* in, const void* w, const float* bias,
                        const void* pro_scale, const void* pro_shift, const void* residual, void* out,
                        hipStream_t stream, const ConvSplit* split = nullptr, const ConvFuse2* fuse2 = nullptr,
                        const ConvProjSc* psc = nullptr, const ConvRebuild* rebuild = nullptr);
bool conv_f16_fuse2_supported(const MetroConvDesc& d, int c2);
// 256 x 256 x 64 GEMM, four waves of 128 x 128, register-staged operands (conv_gemm4w.hip): the pre-activated deep-K 1x1 layers
// (conv1, projection shortcut, shortcut + conv1 pair of blocks 3-4) with at least one tile per CU
bool conv_gemm4w_shape_ok(const MetroConvDesc& d, const ConvSplit* split);      // what the kernel can run
bool conv_gemm4w_supported(const MetroConvDesc& d, const ConvSplit* split);     // ... and when the dispatcher prefers it
int launch_conv_gemm4w(const MetroConvDesc& d, const void* in, const void* w, const float* bias, const void* pro_scale,
                       const void* pro_shift, const void* residual, void* out, hipStream_t stream, const ConvSplit* split);
// persistent pipelined kernel for block1's 64-channel 1x1 convolutions (conv_pw64.hip); mode: 0 plain,
// 1 projection shortcut + conv1 pair (c_out = 256 + 64 concatenated rows), 2 conv3 + the next unit's conv1, 3 = 2 with the projection
// shortcut computed in the launch, 4 = 3 with the residual rebuilt / the sum kept on chip (ConvRebuild)
bool conv_pw64_supported(const MetroConvDesc& d, int mode);
int launch_conv_pw64(const MetroConvDesc& d, const void* in, const void* w, const float* bias, const void* pro_scale,
                     const void* pro_shift, const void* residual, void* out, hipStream_t stream,
                     const ConvSplit* split, const ConvFuse2* fuse2, const ConvProjSc* psc = nullptr,
                     const ConvRebuild* rebuild = nullptr);
// producer / consumer form of the conv3 + next conv1 launches of block1 whose sum stays on chip (conv_b1.hip)
bool conv_b1_chain_preferred();
void conv_b1_set_form(int classic);      // thread-local test switch: 1 = never dispatch it (nor conv_pws.hip's kernel)
bool classic_forms_forced();
// conv3 + shortcut of blocks 3-4 with the two waves of a SIMD half a tile apart (conv_pws.hip); same bits as conv_pw64's <k256|k512,wm8,res>
bool conv_pws_supported(const MetroConvDesc& d);
int launch_conv_pws(const MetroConvDesc& d, const void* in, const void* w, const float* bias, const void* res, void* out, hipStream_t stream);
int launch_conv_b1_chain(const MetroConvDesc& d, const void* in, const void* w, const float* bias, void* out, hipStream_t stream,
                         const ConvFuse2& f2, const ConvProjSc& psc, const ConvRebuild& rb);
// stem 7x7/2 conv + zero-padded 3x3/2 max-pool in one persistent kernel (stem_pool_f16.hip); input is the
// bordered 4-channel fp16 image of launch_prep_input_f16, weights packed [64][7][8][4]
bool stem_pool_f16_supported(int side, int base_width);
int launch_stem_pool_f16(const void* prepped, const void* w, const float* bias, void* out, int n, int side,
                         hipStream_t stream);
// same, reading the fp32 NHWC3 crops directly (prep_input_f16 fused away)
bool stem_pool_f32in_supported(int side, int base_width);
int launch_stem_pool_f32in(const float* images, const void* w, const float* bias, void* out, int n, int side,
                           hipStream_t stream);
// persistent weight-resident 3x3 for the 64 -> 64 channel layers (conv3x3_c64.hip)
bool conv3x3_c64_supported(const MetroConvDesc& d);
int launch_conv3x3_c64(const MetroConvDesc& d, const void* in, const void* w, const float* bias, void* out, hipStream_t stream,
                       const ConvPre1* pre1 = nullptr);
// 3x3 stride-1 convs with tap reuse from an LDS-resident activation slab
bool conv3x3_slab_supported(const MetroConvDesc& d);
int launch_conv3x3_slab(const MetroConvDesc& d, const void* in, const void* w, const float* bias, void* out,
                        hipStream_t stream);
int launch_conv_f64acc(const MetroConvDesc& d, const void* in, const double* w, const double* bias,
                       const double* pro_scale, const double* pro_shift, const void* residual,
                       void* out, hipStream_t stream);
// fp32 conv on the fp32 matrix cores (conv_igemm_f32.hip): METRO_PREC_F32M
int launch_conv_f32m(const MetroConvDesc& d, const void* in, const float* w, const float* bias, const float* pro_scale,
                     const float* pro_shift, const void* residual, void* out, hipStream_t stream);
int launch_prep_input_f16(const float* images, int n, int side, void* out, hipStream_t stream);
int launch_warp_crop_u8(const unsigned char* img, int h, int w, int row_stride, const float* homs, float* out,
                        int n, int side, hipStream_t stream);
int launch_eval_metrics(const float* pred, const float* truth, const unsigned char* valid, int n, int nj,
                        float threshold, float* dist, float* dist_pa, double* sums, hipStream_t stream);
int launch_maxpool(const void* in, void* out, int n, int h_in, int w_in, int c, int dtype,
                   hipStream_t stream);

struct SoftArgmaxArgs {
    int n, side, depth, n_joints_head, n_joints_out;
    int lrc, half_off;           // decode constants (reference volumetric.py:288-295)
    float box_size_mm;
    int proc_side;
    int perm[METRO_MAX_JOINTS];
};
int softargmax_slabs(int n, int side);
int64_t softargmax_scratch_bytes(int n, int side, int n_joints_head);
int launch_softargmax(const void* logits, const SoftArgmaxArgs& a, int precise, void* partials,
                      float* poses_out, hipStream_t stream, float* coords01_out = nullptr, int32_t* status = nullptr);
// `status` (optional, int32 [n]): 1 where an image's soft-argmax statistics were not finite (fp16 overflow upstream), else 0
// finalize only (slabs folded, mm decode, root-relative, permutation) on fp32 partials written by another kernel
int launch_softargmax_finalize(const float* partials, const SoftArgmaxArgs& a, int slabs, float* poses_out,
                               hipStream_t stream, float* coords01_out = nullptr, int32_t* status = nullptr);
// the volumetric head in one launch (head_f16.hip): postnorm prologue + logits GEMM + per-joint softmax statistics
bool head_f16_supported(int c_in, int c_head, int n_joints, int depth, int side);
int head_f16_slabs(int side);               // records per image the partials slot must hold
int head_f16_records(int n, int c_in, int c_head, int side);      // records per image a launch at batch n writes
int launch_head_f16(const void* x, const void* w, const float* bias, const void* pro_scale, const void* pro_shift,
                    int n, int c_in, int c_head, int n_joints, int depth, int side, float* partials, float* logits_out,
                    hipStream_t stream);
// alternative decode heads (heads.hip): root_z != NULL selects true-root-depth, else the bone-length solve
int launch_backproject(const float* coords01, const float* inv_k, const double* targets, int per_pose_targets,
                       const float* root_z, const int* edges, int n, int nj, int ne, const MetroSpec& spec,
                       int root_relative, int permute, float* out, float* z_out, hipStream_t stream);
int launch_heatmap_to_25d(const float* coords01, float* out, int n, const MetroSpec& spec, hipStream_t stream);
int launch_to_orig_cam(const float* x, const float* rot, const int* mirror, float* out, int n, int nj, hipStream_t stream);
SoftArgmaxArgs make_softargmax_args(const MetroSpec& spec, int n);

}  // namespace metro
