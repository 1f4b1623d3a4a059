/*
 * metro_hip.h -- C ABI of libmetro_hip.so: the MI355X (gfx950) implementation of the MeTRo
 * inference hot path (ResNet-v2 backbone -> 1x1 volumetric head -> soft-argmax -> mm decode).
 *
 * The reference (isarandi/metro-pose3d) is pure Python/TensorFlow and has NO native interface:
 * every FLOP of this path runs inside `sess.run` of a frozen GraphDef (reference
 * inference.py:25-27,31-43).  This header is therefore the boundary a binding would attach to
 * in place of `tf.import_graph_def` + `Session.run`; each entry point names the reference
 * code whose work it replaces.  Plain pointers and sizes only; no torch / TF types.
 *
 * Conventions:
 *   - every function returns 0 on success or a negative MetroStatus; metro_last_error()
 *     returns a thread-local message for the last failure on the calling thread;
 *   - all `void* d_*` pointers are DEVICE pointers owned by the caller; the library never
 *     allocates or frees device memory and never synchronises the stream;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *   - a plan is bound to the device current at creation and is not thread-safe.
 */
#ifndef METRO_HIP_H
#define METRO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define METRO_ABI_VERSION 8

typedef enum MetroStatus {
    METRO_OK = 0,
    METRO_ERR_INVALID_ARG = -1,
    METRO_ERR_UNSUPPORTED = -2,
    METRO_ERR_HIP = -3,
    METRO_ERR_STATE = -4,
    METRO_ERR_NONFINITE = -5   /* metro_forward_status: non-finite activations reached the soft-argmax */
} MetroStatus;

/* Arithmetic mode of a plan.  Images are always fp32 in, poses fp32 out.
 * F16: activations/weights fp16, fp32 MFMA accumulation, fp32 soft-argmax -- the reference's
 *      default compute dtype (reference src/options.py:73, src/tfu.py:426-440).  Throughput mode.
 * F32: activations fp32 in HBM, every contraction accumulated with v_mfma_f64_16x16x4_f64
 *      from fp64-folded weights, fp64 soft-argmax; one rounding to fp32 per layer output.
 *      Sits at the fp32 storage noise floor (~1e-3 mm vs exact arithmetic).
 * F64: as F32 but activations and logits are stored as fp64 too: the parity mode, measured
 *      against the fp64 oracle (bar <= 1e-3 mm; lands orders of magnitude below).
 * F32M: the arithmetic of the reference's fp32 graph (--dtype=float32, reference src/options.py:73): fp32 activations, fp32
 *      (BN-folded) weights, every contraction on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation in ascending
 *      k), fp64 soft-argmax on the fp32 logits.  The parity mode at fp32 speed: sits AT the noise floor two correct fp32
 *      implementations have between them (1-5e-3 mm), not under the 1e-3 mm bar. */
typedef enum MetroPrecision { METRO_PREC_F16 = 0, METRO_PREC_F32 = 1, METRO_PREC_F64 = 2, METRO_PREC_F32M = 3 } MetroPrecision;

typedef enum MetroDType { METRO_F16 = 0, METRO_F32 = 1, METRO_F64 = 2 } MetroDType;

#define METRO_MAX_JOINTS 64

/* The constants a frozen graph of the reference bakes in from its flags
 * (reference src/options.py:41,73,96,109-119; src/main.py:106-128). */
typedef struct MetroSpec {
    int32_t arch;                 /* 50 | 101: resnet_v2_50 / resnet_v2_101 (resnet_v2.py:272-312) */
    int32_t stride;               /* 4 | 8 | 16 | 32: --stride-test (options.py:96)                */
    int32_t n_joints_head;        /* J_head: head emits depth * J_head channels (volumetric.py:158) */
    int32_t depth;                /* D = 8 (options.py:113)                                         */
    int32_t centered_stride;      /* 1 (options.py:118)                                             */
    int32_t proc_side;            /* 256 (options.py:41)                                            */
    float   box_size_mm;          /* 2200 (options.py:119)                                          */
    int32_t base_width;           /* 64 for ResNet-50/101; smaller values = toy specs for tests     */
    int32_t precision;            /* MetroPrecision                                                 */
    int32_t n_joints_out;         /* Jout: rows of `output` (main.py:127)                           */
    int32_t permutation[METRO_MAX_JOINTS]; /* output row i = head joint permutation[i] (main.py:119-125) */
} MetroSpec;

typedef struct MetroPlan MetroPlan;

typedef enum MetroParamKind {
    METRO_PARAM_CONV_W = 0,   /* conv kernel, packed [c_out][kh][kw_pad][c_in_pad], BN-folded if bn_var != "" */
    METRO_PARAM_BIAS = 1,     /* per-c_out bias: conv biases, or beta - mean*scale of the folded BN           */
    METRO_PARAM_PRO_SCALE = 2,/* pre-activation BN as prologue: gamma / sqrt(var + 1e-5), per c_in            */
    METRO_PARAM_PRO_SHIFT = 3 /* beta - mean * scale, per c_in                                                */
} MetroParamKind;

/* One tensor of the plan's parameter blob.  The caller fills the blob (host side, any language)
 * from a TF-slim style variable dictionary and uploads it; see INTEGRATION.md. */
typedef struct MetroParamInfo {
    char    name[96];      /* plan-local name, e.g. "block3/unit_2/conv2/W"                      */
    char    conv_var[160]; /* slim scope of the source conv ("" for prologue tensors)            */
    char    bn_var[160];   /* slim scope of the BatchNorm folded in / used as prologue, or ""    */
    int32_t kind;          /* MetroParamKind                                                     */
    int32_t dtype;         /* MetroDType of the packed tensor                                    */
    int32_t c_out, kh, kw, c_in;   /* logical conv dims (c_out = length for 1-D tensors)        */
    int32_t kw_pad, c_in_pad;      /* packed dims; padding is zero-filled                        */
    int64_t offset;        /* byte offset in the blob (256-byte aligned)                         */
    int64_t bytes;
} MetroParamInfo;

#define METRO_FUSED_CONV1_IN_FRONT 1      /* '<unit>/conv1+conv2': conv1 (1x1 on the pre-activated unit input, folded BN + ReLU,
                                           * reference resnet_v2.py:119,127-128) runs on the 3x3 layer's LDS-resident input slab;
                                           * the layer's input tensor is the unit's RAW input (c_in = its channels)          */
#define METRO_FUSED_PROJECTION_SHORTCUT 2 /* conv3 launch that computes the unit's projection shortcut (resnet_v2.py:122-125)
                                           * from the unit's raw input instead of reading a shortcut tensor: has_residual = 0 */
/* block1 without its 256-channel residual stream in HBM (round 5; reference resnet_v2.py:119-138 of block1's units): */
#define METRO_FUSED_OUT_ON_CHIP 4         /* the launch's primary output (the unit's sum) is NOT written by metro_forward -- it only
                                           * feeds the next unit's conv1 inside the launch; metro_forward_upto(last_layer = this
                                           * layer) runs the storing form of the same kernel and writes it at out_offset        */
#define METRO_FUSED_REBUILT_SHORTCUT 8    /* conv3 launch whose identity shortcut x_{u-1} is rebuilt in the launch from the
                                           * 64-channel tensors it is a function of (the previous unit's conv2 output and conv3
                                           * weights + the block's projection shortcut of the pooled stem output) instead of being
                                           * read: has_residual / res_stride / res_offset still state the reference's shortcut   */
#define METRO_FUSED_COMPACT_SHORTCUT 16   /* conv3 of a strided unit whose sub-sampled shortcut (resnet_v2.py:113-121) is read from
                                           * the compact [n, h_out, w_out, c] copy the previous launch wrote at its out_sub_offset;
                                           * has_residual / res_stride / res_offset still state the reference's gather            */

typedef struct MetroLayerInfo {
    char    name[96];
    int32_t kind;          /* 0 input-prep, 1 conv, 2 max-pool, 3 soft-argmax partial, 4 finalize */
    int32_t h_in, w_in, c_in, h_out, w_out, c_out, kh, kw, stride, dilation, pad_top, pad_left;
    int32_t has_prologue, relu, has_residual, res_stride, res_offset;
    int32_t out_dtype;     /* MetroDType of the layer output in the workspace                    */
    int64_t out_offset;    /* byte offset of the output tensor in the workspace.  The 'logits' layer of an f16 plan
                            * whose head runs as one launch (head_f16) NEVER writes this slot during metro_forward:
                            * the logits stay on chip; only metro_forward_upto(last_layer = logits) fills it       */
    int64_t out_bytes_per_image;
    double  flops_per_image; /* 2*MACs (convs only), SURVEY.md section 8d accounting              */
    int64_t out2_offset;   /* fused launches with a second output tensor (shortcut+conv1 pairs,  */
    int32_t out2_channels; /*   conv3+next conv1): its workspace offset / channels; -1 / 0 = none.
                            * METRO_FUSED_CONV1_IN_FRONT layers: conv1's output, which lives in LDS during
                            * metro_forward and is written here ONLY by metro_forward_upto(last_layer = this layer) */
    int32_t fused_flags;   /* METRO_FUSED_*: other layers of the unit computed inside this launch        */
    /* algorithmic HBM bytes of this launch, every tensor it touches counted once (bench.py: the minimum the measured
     * rocprofv3 FETCH_SIZE/WRITE_SIZE traffic is compared with): activations read + written per image (input, outputs,
     * shortcut), and the parameter tensors it reads (once per launch, batch independent). */
    int64_t algo_act_bytes_per_image;
    int64_t algo_param_bytes;
    int64_t out_sub_offset; /* >= 0: the launch also (or only: METRO_FUSED_OUT_ON_CHIP) writes pixels (out_sub_off + 2 i,
                             * out_sub_off + 2 j) of its primary output as a compact [n, out_sub_side, out_sub_side, c_out]
                             * tensor here -- what the next, strided unit's shortcut reads; -1 = none                       */
    int32_t out_sub_side, out_sub_off;
} MetroLayerInfo;

/* ---- plan life cycle: replaces tf.import_graph_def of the frozen graph
 *      (reference inference.py:31-43) and the graph construction of
 *      volumetric.build_inference_model (reference src/model/volumetric.py:152-216). ---- */
int  metro_plan_create(const MetroSpec* spec, int32_t max_batch, MetroPlan** out_plan);
int  metro_plan_destroy(MetroPlan* plan);
int64_t metro_plan_workspace_bytes(const MetroPlan* plan);
int64_t metro_plan_param_bytes(const MetroPlan* plan);
int32_t metro_plan_num_params(const MetroPlan* plan);
int  metro_plan_param_info(const MetroPlan* plan, int32_t index, MetroParamInfo* out);
int32_t metro_plan_num_layers(const MetroPlan* plan);
int  metro_plan_layer_info(const MetroPlan* plan, int32_t index, MetroLayerInfo* out);
double metro_plan_flops_per_image(const MetroPlan* plan);
/* Which kernel instantiation layer `index` runs on at batch n (1 <= n <= max_batch): the choice depends on the layer's
 * shape AND on the batch (tile counts against the 256 CUs), so parity established at one batch does not transfer to another
 * unless the id is the same.  Writes a NUL-terminated id such as "conv3x3_f16_slab<128x256,rows384,bufs2,tps1,kc64,ws3>" or
 * "conv_igemm_f16_dma<128x128,bk64,s4,pro>+pair" ("a & b" when the layer launches two kernels).  A dry run of the layer's
 * dispatch code: nothing is launched and no device is needed.  tests/test_kernel_coverage.py requires every id the
 * BASELINE configurations dispatch at their per-GPU batch to be the id of a single-kernel test that compares with the oracle. */
int  metro_plan_layer_kernel(const MetroPlan* plan, int32_t index, int32_t n, char* buf, int32_t buf_len);
/* Test instrumentation for the single-kernel entry points below (thread-local): mode 0 off (default); 1 every launch on this
 * thread appends its kernel id to the string metro_last_kernel_id() returns; 2 DRY RUN -- entry points record the id and
 * return METRO_OK without launching.  Setting a mode clears the string. */
int  metro_kernel_notes(int32_t mode);
const char* metro_last_kernel_id(void);
/* Binds the uploaded parameter blob (device pointer, metro_plan_param_bytes() long). */
int  metro_plan_bind_params(MetroPlan* plan, const void* d_param_blob);

/* ---- the hot path: replaces sess.run(poses_tensor) (reference inference.py:25-27), i.e.
 *      architectures.resnet (architectures.py:24-35) -> net_output_to_heatmap_and_coords
 *      (volumetric.py:227-235) -> heatmap_to_metric (volumetric.py:303-306) -> root_relative
 *      (tfu3d.py:23-25) -> tf.gather(permutation) (main.py:127).
 *      d_images_nhwc: fp32 [n,256,256,3] in [0,1];  d_poses_out: fp32 [n,Jout,3] in mm. ---- */
int  metro_forward(MetroPlan* plan, const float* d_images_nhwc, int32_t n, float* d_poses_out,
                   void* d_workspace, void* stream);
/* Small batches are launch-latency bound (45 dependent launches for ResNet-50 stride 16): forwards with
 * n <= max_batch_for_graphs are captured once per (n, buffers, stream) into a hipGraph and replayed.
 * 0 (default) = always plain launches.  The capture happens on the second call with a given key. */
int  metro_plan_set_graph_max_batch(MetroPlan* plan, int32_t max_batch_for_graphs);
/* Non-finite screen of the LAST metro_forward(n) on this workspace.  The reference keeps fp32 variables under fp16 compute
 * (reference src/tfu.py:426-440) and TensorFlow hands NaN poses back silently when an activation overflows fp16 (65 504); here
 * the finalize launch writes one int32 per image into the workspace (1 = a soft-argmax record, maximum or normaliser of that
 * image was not finite; a NaN record is never silently dropped) and this call copies the n words to the host -- it
 * SYNCHRONISES the stream, the only entry point of the path that does.  *n_nonfinite_out = number of flagged images; returns
 * METRO_ERR_NONFINITE (metro_last_error() names the remedy: precision f32m / f64) when it is not zero. */
int  metro_forward_status(const MetroPlan* plan, const void* d_workspace, int32_t n, void* stream, int32_t* n_nonfinite_out);
/* Byte offset, inside the workspace, of the int32[max_batch] non-finite words metro_forward_status reads: a caller that chains
 * several forwards (more crops than the plan's max_batch) can fold them on the device after each one and synchronise ONCE.  The
 * words are those of the LAST forward on this workspace: valid until the next metro_forward on it, on the same stream only. */
int64_t metro_plan_status_offset(const MetroPlan* plan);
/* Same, stopping after layer `last_layer` (inclusive) so tests can read that layer's output at
 * MetroLayerInfo.out_offset in the workspace.  d_poses_out may be NULL if the finalize layer
 * is not reached. */
int  metro_forward_upto(MetroPlan* plan, const float* d_images_nhwc, int32_t n,
                        float* d_poses_out, void* d_workspace, void* stream,
                        int32_t last_layer);

/* ---- per-layer timing on the launch stream (HIP events around every launch; this is what
 *      bench.py's roofline object is computed from).  ms_out[i] accumulates layer i's time. ---- */
int  metro_forward_timed(MetroPlan* plan, const float* d_images_nhwc, int32_t n,
                         float* d_poses_out, void* d_workspace, void* stream,
                         float* ms_out /* [num_layers] host */);

/* ---- single-kernel entry points (parity tests call these through ctypes) ---- */

/* Implicit-GEMM convolution over NHWC (replaces slim.conv2d / conv2d_same call sites:
 * reference resnet_v2.py:123-136,219-220,233-236; resnet_utils.py:82-135).  Generic over
 * kernel size, stride, dilation and asymmetric TF padding; optional per-input-channel
 * scale/shift+ReLU prologue (pre-activation BN, resnet_v2.py:119,229) and
 * bias / ReLU / strided-shifted residual epilogue (resnet_v2.py:113-121,138). */
typedef struct MetroConvDesc {
    int32_t n, h_in, w_in, c_in;
    int32_t in_pix_stride;      /* elements between consecutive input pixels (>= c_in)      */
    int32_t h_out, w_out, c_out;
    int32_t kh, kw, stride, dilation;
    int32_t pad_top, pad_left;  /* input row = ho*stride - pad_top + r*dilation             */
    int32_t has_prologue;       /* requires kh == kw == 1 and no padding                    */
    int32_t relu;
    int32_t has_residual;
    int32_t res_h, res_w;       /* spatial dims of the residual tensor [n,res_h,res_w,c_out] */
    int32_t res_stride, res_offset; /* residual pixel = (ho*res_stride+res_offset, wo*...)  */
    int32_t out_dtype;          /* MetroDType of output AND residual: F16/F32 (fast kernel), F32/F64 (precise kernel) */
    int32_t in_dtype;           /* MetroDType of the input: F16 (fast kernel), F32/F64 (precise kernel) */
} MetroConvDesc;

/* fp16 operands, fp32 MFMA accumulate (v_mfma_f32_32x32x16_f16).  Weights [c_out][kh*kw*c_in]
 * fp16, bias fp32[c_out], prologue scale/shift fp16[c_in], residual fp16. c_in % 8 == 0. */
int  metro_conv_f16(const MetroConvDesc* d, const void* d_in, const void* d_w, const float* d_bias,
                    const void* d_pro_scale, const void* d_pro_shift, const void* d_residual,
                    void* d_out, void* stream);
/* Two 1x1 convolutions of a bottleneck unit on the SAME pre-activated input in one launch: the projection
 * shortcut (rows [0, split) of d_w / d_bias, no ReLU, -> d_out with `split` channels) and conv1 with its folded
 * BN + ReLU (rows [split, d->c_out) -> d_out2 with d->c_out - split channels).  Replaces the two
 * layers_lib.conv2d calls of reference src/model/resnet_v2.py:122-128.  d->has_prologue = 1, no residual. */
int  metro_conv_f16_pair(const MetroConvDesc* d, const void* d_in, const void* d_w, const float* d_bias,
                         const void* d_pro_scale, const void* d_pro_shift, void* d_out, int32_t split,
                         void* d_out2, void* stream);
/* conv3 (+bias, + shortcut) of unit u and conv1 (pre-activation BN+ReLU of unit u+1, folded BN + ReLU) of
 * unit u+1 in one launch (block1 shapes: c_in = c2 = 64, c_out = 256):
 *   d_out  = conv(d_in) + bias + residual                          (resnet_v2.py:134-138 of unit u)
 *   d_out2 = relu(W2 * relu(d_out * scale2 + shift2) + bias2)      (resnet_v2.py:119,127-128 of unit u+1)
 * W2 fp16 [c2][c_out], bias2 fp32 [c2], scale2/shift2 fp16 [c_out], d_out2 fp16 [.., c2]. */
int  metro_conv_f16_next(const MetroConvDesc* d, const void* d_in, const void* d_w, const float* d_bias,
                         const void* d_residual, void* d_out, const void* d_w2, const float* d_bias2,
                         const void* d_scale2, const void* d_shift2, void* d_out2, int32_t c2, void* stream);
/* block1/unit_1 in two launches (the unit's input has only 64 channels, so recomputing beats storing):
 * metro_conv_f16_conv1_conv2 -- conv1 (1x1, 64 -> 64, folded BN + ReLU) on relu(x * pro_scale + pro_shift), then conv2 (3x3, SAME,
 *   folded BN + ReLU; d = ITS descriptor, relu = 1) in one launch; conv1's output lives in LDS only (reference
 *   resnet_v2.py:119,127-132).  d_x fp16 [n,h,w,64], d_w1 fp16 [64][64], d_w2 fp16 [64][3][3][64], d_out fp16 [n,h,w,64].
 * metro_conv_f16_next_proj -- metro_conv_f16_next with the unit's PROJECTION shortcut computed in the launch:
 *   d_out = fp16(conv3(d_in) + bias) + fp16(Wsc * relu(d_x * pro_scale + pro_shift) + bias_sc)   (resnet_v2.py:119,122-125,134-138)
 *   d_out2 = relu(W2 * relu(d_out * scale2 + shift2) + bias2)                                    (unit u+1, :119,127-128)
 *   d->has_residual = 0; d_x fp16 [.., 64] the unit's raw input, d_w_sc fp16 [256][64], d_bias_sc fp32 [256].
 *   d_out == NULL: the sum is NOT stored (it only feeds the second GEMM): the form metro_forward runs for block1/unit_1 (round 5).
 * metro_conv_f16_next_rebuild -- the launch of block1/unit_2 (round 5): the unit's identity shortcut x_1 is REBUILT from the tensors
 *   it is a function of instead of being read as a 512-byte-per-pixel tensor:
 *   x_1    = fp16(W3_prev * d_t2_prev + bias3_prev) + fp16(Wsc * relu(d_x * pro_scale + pro_shift) + bias_sc)      (unit 1, :119-138)
 *   d_out  = fp16(conv3(d_in) + bias) + x_1                                                                        (unit 2, :120-121,134-138)
 *   d_out2 = relu(W2 * relu(d_out * scale2 + shift2) + bias2)                                                      (unit 3, :119,127-128)
 *   with the MFMAs, k order and fp16 roundings of the launches that would have stored x_1: the same bits.  Exactly one of d_out
 *   (the whole sum, fp16 [n,h,w,256]) and d_out_sub (pixels (sub_off + 2i, sub_off + 2j) of it as a compact
 *   [n, ceil((h - sub_off) / 2), ceil((w - sub_off) / 2), 256] tensor: what a strided unit 3's shortcut reads, resnet_v2.py:113-121)
 *   must be non-NULL.  w a power of two >= 16, h * w % 64 == 0. */
int  metro_conv_f16_conv1_conv2(const MetroConvDesc* d, const void* d_x, const void* d_w1, const float* d_bias1, const void* d_pro_scale,
                                const void* d_pro_shift, const void* d_w2, const float* d_bias2, void* d_out, void* stream);
int  metro_conv_f16_next_proj(const MetroConvDesc* d, const void* d_in, const void* d_w, const float* d_bias, const void* d_x,
                              const void* d_w_sc, const float* d_bias_sc, const void* d_pro_scale, const void* d_pro_shift, void* d_out,
                              const void* d_w2, const float* d_bias2, const void* d_scale2, const void* d_shift2, void* d_out2,
                              int32_t c2, void* stream);
int  metro_conv_f16_next_rebuild(const MetroConvDesc* d, const void* d_in, const void* d_w, const float* d_bias, const void* d_x,
                                 const void* d_w_sc, const float* d_bias_sc, const void* d_pro_scale, const void* d_pro_shift,
                                 const void* d_t2_prev, const void* d_w3_prev, const float* d_bias3_prev, void* d_out, void* d_out_sub,
                                 int32_t sub_off, const void* d_w2, const float* d_bias2, const void* d_scale2, const void* d_shift2,
                                 void* d_out2, int32_t c2, void* stream);
/* Test switch (thread-local) for the two entry points above when their sum stays on chip or is rebuilt: metro_forward runs them on
 * the producer / consumer kernel of conv_b1.hip ("conv_b1_chain<...>"); 1 = run the classic single-role kernel of conv_pw64.hip
 * instead (what metro_forward_upto stopping at such a layer runs): two independent forms that must give the same bits.
 * It also switches metro_conv_f16 between conv_pws.hip's skewed kernel (default; conv3 + shortcut of blocks 3-4) and
 * conv_pw64.hip's lock-step one (1), and metro_conv_f16_pair on block2's shapes (256 -> 512 + 128) between the weight-resident
 * kernel conv_pw64<k256,wm8,cb512,pro,pair> (default, round 5) and the ring kernel it replaced (1): again the same bits.
 * Round 6: and the dilated 3x3 layers whose halo exceeds the tap-reuse kernel's slab (rate 4 / 8 at stride 4 and 8) between that
 * kernel in sub-grid pixel order ("conv3x3_f16_slab<...>+subgrid", default) and the generic ring kernel (1): two fp32 summation
 * orders of the same products (chunk-major / tap-major), equal up to rounding flips of the fp16 result. */
int  metro_conv_b1_form(int32_t classic);
/* The same contract as metro_conv_f16 / metro_conv_f16_pair on the 256 x 256 x 64 GEMM kernel with four waves of 128 x 128
 * (conv_gemm4w.hip: register-staged operands, one barrier per K tile), which metro_forward picks for the pre-activated deep-K
 * 1x1 layers with at least one tile per CU (conv1, projection shortcut, shortcut+conv1 pair of blocks 3-4: reference
 * resnet_v2.py:122-128; >= 256 tiles of 256 x 256, K >= 1024 or >= 1024 tiles).
 * 1x1, stride 1, c_in % 128 == 0, c_out % 256 == 0, n*h*w % 256 == 0.  split > 0: fused pair (rows [0,split) ->
 * d_out, rows [split, c_out) with ReLU -> d_out2, (c_out - split) % 256 == 0); split == 0: plain layer, d_out2 ignored.
 * Same K order and one fp32 accumulator per output as every other fp16 conv kernel here: bit-identical to them.
 * Round 6: the entry picks the tile itself, as metro_forward does -- whole 256 x 256 tiles; HALF tiles (256 cout x 128 pixels) for
 * a layer with fewer than 256 whole but >= 224 half tiles and K >= 1024 (block4's conv1 at 64 crops); QUARTER tiles (256 x 64) for
 * K >= 2048 below that (32 crops); a pair whose first output fills whole rounds of 256 CUs while its second one has < 256 whole
 * tiles as whole + half tiles in one grid (block4's pair at 64 crops).  metro_last_kernel_id names the forms
 * ("conv_gemm4w<256x128,pro>", "...<256x256,pro>+pair & ...<256x128,pro>+pair").  Same bits whatever the tile.
 * (Two earlier forms of this GEMM that metro_forward never dispatches -- conv_gemm8p, conv_gemm4d -- are built into
 * libmetro_experimental.so: metro_pose3d_amd/csrc/experimental/metro_experimental.h.) */
int  metro_conv_f16_gemm4w(const MetroConvDesc* d, const void* d_in, const void* d_w, const float* d_bias,
                           const void* d_pro_scale, const void* d_pro_shift, const void* d_residual, void* d_out,
                           int32_t split, void* d_out2, void* stream);
/* Stem 7x7/2 convolution (+bias) and zero-padded 3x3/2 max-pool in one launch (reference resnet_v2.py:219-224,
 * resnet_utils.py:138-185).  d_prepped = metro_prep_input_f16 output [n,side+6,side+8,4] fp16, d_w packed
 * [64][7][8][4] fp16, d_out fp16 [n,side/4,side/4,64].  side % 32 == 0. */
int  metro_stem_pool_f16(const void* d_prepped, const void* d_w, const float* d_bias, void* d_out, int32_t n,
                         int32_t side, void* stream);
/* Same, reading the fp32 NHWC crops [n,side,side,3] directly: the fp32->fp16 cast (reference
 * src/model/architectures.py:29) and the stem's zero border happen on the way into LDS. */
int  metro_stem_pool_f32in(const float* d_images, const void* d_w, const float* d_bias, void* d_out, int32_t n,
                           int32_t side, void* stream);
/* fp32 or fp64 activations (in_dtype / out_dtype), fp64 weights/bias/prologue,
 * v_mfma_f64_16x16x4_f64 accumulate, one rounding to out_dtype. */
int  metro_conv_f64acc(const MetroConvDesc* d, const void* d_in, const double* d_w,
                       const double* d_bias, const double* d_pro_scale,
                       const double* d_pro_shift, const void* d_residual, void* d_out,
                       void* stream);

/* fp32 activations, fp32 weights / bias / prologue, v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation): the conv
 * kernel of the METRO_PREC_F32M mode.  in_dtype = out_dtype = METRO_F32; any c_in. */
int  metro_conv_f32m(const MetroConvDesc* d, const void* d_in, const float* d_w, const float* d_bias, const float* d_pro_scale,
                     const float* d_pro_shift, const void* d_residual, void* d_out, void* stream);

/* fp32 NHWC [n,side,side,3] -> zero-bordered fp16 [n,side+6,side+8,4] (the stem's explicit
 * pad-3 of reference resnet_utils.py:125-135 materialised once; channel 3 is zero). */
int  metro_prep_input_f16(const float* d_images, int32_t n, int32_t side, void* d_out, void* stream);

/* Crop pre-processing, the step before the path: n crops of one uint8
 * HWC RGB frame [h, w, 3] (row_stride bytes per row), crop i sampled through the 3x3 homography
 * d_homographies[i] (row-major fp32, maps OUTPUT pixel (x, y, 1) to SOURCE pixel coordinates) with
 * OpenCV's 8-bit remap rule (coordinates rounded to 1/32 px, 15-bit weights, result rounded to uint8, constant-0 border: the
 * reference warps the uint8 frame), then /255 and clip -- reference src/cameralib.py:406-429 (reproject_image_fast ->
 * cv2.remap INTER_LINEAR) + src/improc.py:56-61 (normalize01).  Every output value is k/255 for the byte k cv2 produces.
 * h, w <= 32767 (OpenCV holds the integer coordinate in a short).
 * d_out: fp32 NHWC [n, side, side, 3], exactly the input contract of metro_forward. */
int  metro_warp_crop_u8(const uint8_t* d_image, int32_t h, int32_t w, int32_t row_stride,
                        const float* d_homographies, int32_t n, int32_t side, float* d_out, void* stream);

/* 3x3 stride-2 max-pool over a ZERO-padded (1,1) input (reference resnet_utils.py:177-185).
 * dtype METRO_F16 / METRO_F32 / METRO_F64; c % 8 == 0 (f16), c % 4 == 0 (f32), c % 2 == 0 (f64). */
int  metro_maxpool3x3s2_zeropad(const void* d_in, void* d_out, int32_t n, int32_t h_in,
                                int32_t w_in, int32_t c, int32_t dtype, void* stream);

/* Soft-argmax over the (H,W,D) volume per joint from fp32 NHWC logits [n,side,side,depth*J]
 * (channel = d*J + j), then mm decode, root-relative and joint permutation (reference
 * volumetric.py:227-235,288-306; tfu.py:466-499; tfu3d.py:23-25; main.py:127).
 * d_partials: scratch of metro_softargmax_scratch_bytes().  precise = 0: fp32 logits, fp32
 * accumulators; 1: fp32 logits, fp64 accumulators; 2: fp64 logits, fp64 accumulators. */
int64_t metro_softargmax_scratch_bytes(int32_t n, int32_t side, int32_t n_joints_head);
int  metro_softargmax(const void* d_logits, int32_t n, const MetroSpec* spec, int32_t precise,
                      void* d_partials, float* d_poses_out, void* stream);

/* The volumetric head in ONE launch, as metro_forward runs it in f16 mode for heads of <= 160 channels (head_f16.hip):
 * postnorm BN + ReLU on the raw residual stream (reference resnet_v2.py:229), the 1x1 logits convolution + bias (:233-236,
 * fp32 accumulators, architectures.py:34), the per-joint softmax statistics of every 32-pixel slab from the on-chip logits
 * tile (volumetric.py:227-235, tfu.py:466-499), then the slab fold / mm decode / root-relative / gather of metro_softargmax.
 * The kernel (tile of 256 / 128 / 64 pixels, K-parts per wave group) is chosen from the batch: results are reproducible for a
 * given n, not bit-identical across n (the K-parts are added in a fixed order that depends on the tile).
 * d_x fp16 [n, side, side, c_in]; d_w fp16 [depth * J][c_in]; d_bias fp32; d_pro_scale / d_pro_shift fp16 [c_in];
 * d_partials: metro_head_f16_scratch_bytes(); d_logits_out: optional fp32 NHWC logits dump (NULL in the product path). */
int64_t metro_head_f16_scratch_bytes(int32_t n, int32_t side, int32_t n_joints_head);
int  metro_head_f16(const void* d_x, const void* d_w, const float* d_bias, const void* d_pro_scale, const void* d_pro_shift,
                    int32_t n, int32_t c_in, const MetroSpec* spec, void* d_partials, float* d_logits_out, float* d_poses_out,
                    void* stream);

/* Evaluation metrics, the step after the path (reference
 * src/main.py:339-359): per (pose, joint) root-relative distance in mm before and after rigid
 * alignment with scale (Procrustes without reflection: src/util3d.py:139-159,
 * src/eval/procrustes.py:6-107), and per-joint sums over the valid entries of
 * {count, dist, dist_aligned, max(0, 1 - dist/threshold), dist <= threshold} -> d_sums[J][5] (fp64).
 * d_pred / d_true: fp32 [n, J, 3] (root = last joint); d_valid: uint8 [n, J]. */
int  metro_eval_metrics(const float* d_pred, const float* d_true, const uint8_t* d_valid, int32_t n,
                        int32_t n_joints, float threshold_mm, float* d_dist, float* d_dist_aligned,
                        double* d_sums, void* stream);

/* ---- alternative decode heads, the step AFTER the path ---- */
/* Soft-argmax coordinates in [0,1], head joint order, (x,y,z): the `coords3d` that
 * net_output_to_heatmap_and_coords returns (reference src/model/volumetric.py:227-235), i.e. metro_softargmax
 * without heatmap_to_metric / root_relative / gather.  d_coords01_out fp32 [n, n_joints_head, 3]. */
int  metro_softargmax01(const void* d_logits, int32_t n, const MetroSpec* spec, int32_t precise, void* d_scratch,
                        float* d_coords01_out, void* stream);
/* `--scale-recovery=bone-lengths` / `bone-lengths-true` (volumetric.py:171-191): heatmap_to_image (:288-295),
 * rays = inv_intrinsics . [u,v,1] (:221-222), delta_z = (z - z_root) * box_size, per-pose z offset by the
 * reference's scipy Levenberg-Marquardt solve (src/model/bone_length_based_backproj.py:38-62; MINPACK lmder
 * restated for one unknown, fp64), back_project (:284-285).  d_bone_lengths fp64 [n_edges] (dataset means) or
 * [n, n_edges] when per_pose_lengths != 0; d_edges int32 [n_edges, 2] head joint indices.  root_relative != 0
 * subtracts the last head joint (tfu3d.py:23-25); permute != 0 gathers spec->permutation (main.py:119-127).
 * d_coords3d_out fp32 [n, J, 3]; d_z_offset_out fp32 [n] or NULL. */
int  metro_backproject_bone_lengths(const float* d_coords01, const float* d_inv_intrinsics, const double* d_bone_lengths,
                                    int32_t per_pose_lengths, const int32_t* d_edges, int32_t n_edges, int32_t n,
                                    const MetroSpec* spec, int32_t root_relative, int32_t permute,
                                    float* d_coords3d_out, float* d_z_offset_out, void* stream);
/* `--scale-recovery=true-root-depth` (volumetric.py:192-199): the same with a given root depth [n] fp32. */
int  metro_backproject_root_depth(const float* d_coords01, const float* d_inv_intrinsics, const float* d_root_z,
                                  int32_t n, const MetroSpec* spec, int32_t root_relative, int32_t permute,
                                  float* d_coords3d_out, void* stream);
/* heatmap_to_25d (volumetric.py:298-300): (x, y) in crop pixels via heatmap_to_image, z * box_size_mm; head order. */
int  metro_heatmap_to_25d(const float* d_coords01, int32_t n, const MetroSpec* spec, float* d_out, void* stream);
/* to_orig_cam (volumetric.py:277-281): x' = R x per joint, joints swapped with their mirror joint when
 * det(R) <= 0.  d_rot fp32 [n,9] row-major, d_mirror int32 [n_joints]. */
int  metro_to_orig_cam(const float* d_coords, const float* d_rot, const int32_t* d_mirror, float* d_out, int32_t n,
                       int32_t n_joints, void* stream);

const char* metro_last_error(void);
int32_t metro_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* METRO_HIP_H */
