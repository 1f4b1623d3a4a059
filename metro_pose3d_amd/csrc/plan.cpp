// Planner + executor + C ABI of libmetro_hip.so.
//
// metro_plan_create restates, in C++, the graph that the reference builds in Python at export
// time (reference src/main.py:106-128 -> src/model/volumetric.py:152-216 ->
// src/model/architectures.py:24-35 -> src/model/resnet_v2.py:142-312 ->
// src/model/resnet_utils.py:263-350) as a flat list of kernel launches over a pre-planned
// workspace.  The test-side oracle (oracle/spec.py) restates the same control flow
// independently in Python; tests/test_abi_and_plan.py compares the two layer by layer.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "metro_common.h"

namespace metro {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* get_error() { return g_err; }

static thread_local KernelNotes g_notes = {0, ""};
KernelNotes& kernel_notes() { return g_notes; }
bool note_kernel(const char* fmt, ...) {
    if (g_notes.mode == 0) return false;
    // Mode 1 accumulates until metro_kernel_notes() is called again; an id that does not fit the 1 KiB string is replaced by a
    // trailing " ..." so a truncated string can never pass for an id (a whole metro_forward of ~46 launches does not fit: mode 1
    // is meant for ONE single-kernel entry-point call at a time).
    const size_t cap = sizeof(g_notes.ids) - 5;           // room for " ..."
    const size_t used = strlen(g_notes.ids);
    if (used >= 4 && strcmp(g_notes.ids + used - 4, " ...") == 0) return g_notes.mode == 2;
    char one[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(one, sizeof(one), fmt, ap);
    va_end(ap);
    const size_t need = strlen(one) + (used ? 3 : 0);
    if (used + need > cap) { memcpy(g_notes.ids + used, " ...", 5); return g_notes.mode == 2; }
    if (used) memcpy(g_notes.ids + used, " & ", 3);       // several kernels: a & b
    memcpy(g_notes.ids + used + (used ? 3 : 0), one, strlen(one) + 1);
    return g_notes.mode == 2;
}

int validate_conv_desc(const MetroConvDesc* d) {
    METRO_CHECK_ARG(d != nullptr, "conv desc is NULL");
    METRO_CHECK_ARG(d->n > 0 && d->h_in > 0 && d->w_in > 0 && d->c_in > 0 && d->h_out > 0 &&
                        d->w_out > 0 && d->c_out > 0,
                    "conv desc: non-positive dimension");
    METRO_CHECK_ARG(d->kh > 0 && d->kw > 0 && d->stride > 0 && d->dilation > 0, "conv desc: bad kernel geometry");
    METRO_CHECK_ARG(d->in_pix_stride > 0, "conv desc: in_pix_stride must be positive");
    METRO_CHECK_ARG((long)d->n * d->h_out * d->w_out < (1L << 31), "conv desc: too many output pixels");
    METRO_CHECK_ARG((long)d->n * d->h_in * d->w_in < (1L << 31), "conv desc: too many input pixels");
    if (d->has_prologue) {
        METRO_CHECK_ARG(d->kh == 1 && d->kw == 1, "conv desc: prologue requires a 1x1 kernel");
        const long last_h = (long)(d->h_out - 1) * d->stride - d->pad_top;
        const long last_w = (long)(d->w_out - 1) * d->stride - d->pad_left;
        METRO_CHECK_ARG(d->pad_top <= 0 && d->pad_left <= 0 && last_h < d->h_in && last_w < d->w_in,
                        "conv desc: prologue requires every tap to be in bounds (no padding)");
    }
    if (d->has_residual) {
        METRO_CHECK_ARG(d->res_stride > 0 && d->res_offset >= 0, "conv desc: bad residual gather");
        METRO_CHECK_ARG((d->h_out - 1) * d->res_stride + d->res_offset < d->res_h &&
                            (d->w_out - 1) * d->res_stride + d->res_offset < d->res_w,
                        "conv desc: residual gather out of bounds");
    }
    return METRO_OK;
}

}  // namespace metro

using namespace metro;

// ---------------------------------------------------------------------------------------------
namespace {

enum LayerKind { LK_PREP = 0, LK_CONV = 1, LK_POOL = 2, LK_SOFTARGMAX = 3 };
enum Slot { S_IMAGES = -2, S_NONE = -1, S_PREP = 0, S_STEM, S_X0, S_X1, S_T1, S_T2, S_T2B, S_SC, S_LOGITS, S_PART, S_STATUS, S_COUNT };

struct Layer {
    MetroLayerInfo info;
    int kind;
    MetroConvDesc cd;     // cd.n is filled per call
    int in_slot, out_slot, res_slot;
    int p_w, p_bias, p_scale, p_shift;
    // fused pair (projection shortcut + conv1 of the same unit, same pre-activated input):
    int split, c_out2, relu2, out2_slot;
    // conv3 of unit u + conv1 of unit u+1 (ConvFuse2): parameter indices of the second GEMM, -1 = none
    int f2_w, f2_bias, f2_scale, f2_shift, f2_c2;
    // block1/unit_1 in two launches (round 3): conv1 fused in FRONT of conv2 (conv3x3_c64 PRE1: parameter indices of conv1 and of the
    // unit's pre-activation, -1 = none) and the projection shortcut computed INSIDE the conv3 launch (conv_pw64 PSC: its
    // parameters and the slot of the unit's input)
    int p1_w, p1_bias, p1_scale, p1_shift;
    int psc_w, psc_bias, psc_scale, psc_shift, psc_slot;
    // block1 without its 256-channel residual stream in HBM (round 5, conv_pw64 REB / OUTM): parameters and slot of the PREVIOUS
    // unit's conv3 (the shortcut is rebuilt from them + psc_*), what metro_forward does with the launch's sum (0 store, 1 keep on
    // chip, 2 sub-sampled compact copy into sub_slot only) and the geometry of that copy
    int reb_w, reb_bias, reb_slot;
    int out_mode, sub_slot, sub_off, sub_side;
    int stem_pool;        // stem conv + max-pool in one launch (the layer's output is the pooled tensor)
    int head_c_in;        // soft-argmax layer of a fused head: input channels of the logits GEMM (which head kernel ran)
    int head_fused;       // logits layer: GEMM + per-joint softmax statistics in one launch (head_f16.hip); the
                          // soft-argmax layer behind it then only finalizes
};

}  // namespace

// A captured forward: valid for exactly this (batch, buffers, stream) tuple.
struct GraphEntry {
    int n;
    const void* images;
    const void* poses;
    const void* ws;
    hipStream_t stream;
    hipGraphExec_t exec;
    int eager_runs;          // the first call for a key runs eagerly (lazy one-time setup must not be captured)
};

struct MetroPlan {
    std::vector<GraphEntry> graphs;
    hipStream_t cap_stream = nullptr;   // private stream used only to CAPTURE (the legacy null stream cannot capture)
    int graph_max_batch = 0;   // forwards with n <= this replay a captured hipGraph (0 = always eager)
    MetroSpec spec;
    int max_batch;
    bool fast;
    int act_dtype;                       // MetroDType of activations in the workspace
    int act_bytes;                       // bytes per activation element in the workspace
    std::vector<MetroParamInfo> params;
    std::vector<Layer> layers;
    int64_t slot_bytes_per_image[S_COUNT];
    int64_t slot_offset[S_COUNT];
    int64_t workspace_bytes;
    int64_t param_bytes;
    const char* d_params;
    double flops_per_image;
};

namespace {

int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

struct Builder {
    MetroPlan* p;
    std::string root;

    int add_param(const std::string& name, int kind, const std::string& conv_var,
                  const std::string& bn_var, int dtype, int c_out, int kh, int kw, int c_in,
                  int kw_pad, int c_in_pad) {
        MetroParamInfo pi;
        memset(&pi, 0, sizeof(pi));
        snprintf(pi.name, sizeof(pi.name), "%s", name.c_str());
        snprintf(pi.conv_var, sizeof(pi.conv_var), "%s", conv_var.c_str());
        snprintf(pi.bn_var, sizeof(pi.bn_var), "%s", bn_var.c_str());
        pi.kind = kind; pi.dtype = dtype;
        pi.c_out = c_out; pi.kh = kh; pi.kw = kw; pi.c_in = c_in; pi.kw_pad = kw_pad; pi.c_in_pad = c_in_pad;
        const int64_t es = dtype == METRO_F16 ? 2 : dtype == METRO_F32 ? 4 : 8;
        const int64_t elems = kind == METRO_PARAM_CONV_W ? (int64_t)c_out * kh * kw_pad * c_in_pad : c_out;
        pi.bytes = elems * es;
        pi.offset = p->param_bytes;
        p->param_bytes = align_up(p->param_bytes + pi.bytes, 256);
        p->params.push_back(pi);
        return (int)p->params.size() - 1;
    }

    void need(int slot, int64_t bytes_per_image) {
        if (slot >= 0) p->slot_bytes_per_image[slot] = std::max(p->slot_bytes_per_image[slot], bytes_per_image);
    }

    // Adds one convolution.  `scope` is the slim scope below root, e.g.
    // "block1/unit_1/bottleneck_v2/conv1"; bn_fold = scope of the BN folded into it ("" = the
    // conv has its own biases); prologue_bn = scope of the pre-activation BN applied to its input.
    void add_conv(const std::string& lname, const std::string& scope, const std::string& bn_fold,
                  const std::string& prologue_bn, int in_slot, int out_slot, int res_slot,
                  int side_in, int c_in, int side_out, int c_out, int k, int stride, int dil,
                  int pad_beg, bool relu, int res_side, int res_stride, int res_offset,
                  int out_dtype, int in_dtype) {
        Layer L;
        memset(&L, 0, sizeof(L));
        L.f2_w = L.f2_bias = L.f2_scale = L.f2_shift = -1; L.p1_w = L.p1_bias = L.p1_scale = L.p1_shift = -1; L.psc_w = L.psc_bias = L.psc_scale = L.psc_shift = -1; L.psc_slot = S_NONE; L.reb_w = L.reb_bias = -1; L.reb_slot = L.sub_slot = S_NONE;
        L.kind = LK_CONV;
        const bool fast = p->fast;
        const bool f32m = p->spec.precision == METRO_PREC_F32M;
        const int wdt = fast ? METRO_F16 : f32m ? METRO_F32 : METRO_F64;
        const int bdt = fast || f32m ? METRO_F32 : METRO_F64;
        MetroConvDesc& cd = L.cd;
        cd.n = 0;
        cd.h_in = cd.w_in = side_in; cd.c_in = c_in; cd.in_pix_stride = c_in;
        cd.h_out = cd.w_out = side_out; cd.c_out = c_out;
        cd.kh = cd.kw = k; cd.stride = stride; cd.dilation = dil;
        cd.pad_top = cd.pad_left = pad_beg;
        cd.has_prologue = !prologue_bn.empty();
        cd.relu = relu;
        cd.has_residual = res_slot != S_NONE;
        cd.res_h = cd.res_w = res_side; cd.res_stride = res_stride; cd.res_offset = res_offset;
        cd.out_dtype = out_dtype;
        cd.in_dtype = in_dtype;
        const std::string conv_var = root + "/" + scope;
        const std::string bn_var = bn_fold.empty() ? "" : root + "/" + bn_fold;
        L.p_w = add_param(lname + "/W", METRO_PARAM_CONV_W, conv_var, bn_var, wdt, c_out, k, k, c_in, k, c_in);
        L.p_bias = add_param(lname + "/bias", METRO_PARAM_BIAS, conv_var, bn_var, bdt, c_out, 1, 1, 1, 1, 1);
        L.p_scale = L.p_shift = -1;
        if (cd.has_prologue) {
            const std::string pv = root + "/" + prologue_bn;
            L.p_scale = add_param(lname + "/pro_scale", METRO_PARAM_PRO_SCALE, "", pv, wdt, c_in, 1, 1, 1, 1, 1);
            L.p_shift = add_param(lname + "/pro_shift", METRO_PARAM_PRO_SHIFT, "", pv, wdt, c_in, 1, 1, 1, 1, 1);
        }
        L.in_slot = in_slot; L.out_slot = out_slot; L.res_slot = res_slot; L.out2_slot = S_NONE;
        const int64_t out_es = out_dtype == METRO_F16 ? 2 : out_dtype == METRO_F32 ? 4 : 8;
        need(out_slot, (int64_t)side_out * side_out * c_out * out_es);
        fill_info(L, lname, (double)2.0 * side_out * side_out * c_out * k * k * c_in);
        p->layers.push_back(L);
    }

    // Projection shortcut (c_sc outputs, bias, no ReLU) and conv1 (cb outputs, folded BN + ReLU) of a
    // unit read the same pre-activated tensor: one launch over concatenated weight rows
    // (reference resnet_v2.py:122-128).  Parameter tensors keep their own names and are laid out
    // back to back so the kernel sees one [c_sc + cb][c_in] matrix.
    int add_shortcut_conv1_pair(const std::string& un, const std::string& sc, int in_slot, int side, int c_in,
                                int c_sc, int cb, int adt) {
        Layer L;
        memset(&L, 0, sizeof(L));
        L.f2_w = L.f2_bias = L.f2_scale = L.f2_shift = -1; L.p1_w = L.p1_bias = L.p1_scale = L.p1_shift = -1; L.psc_w = L.psc_bias = L.psc_scale = L.psc_shift = -1; L.psc_slot = S_NONE; L.reb_w = L.reb_bias = -1; L.reb_slot = L.sub_slot = S_NONE;
        L.kind = LK_CONV;
        MetroConvDesc& cd = L.cd;
        cd.h_in = cd.w_in = side; cd.c_in = c_in; cd.in_pix_stride = c_in;
        cd.h_out = cd.w_out = side; cd.c_out = c_sc + cb;
        cd.kh = cd.kw = 1; cd.stride = 1; cd.dilation = 1;
        cd.has_prologue = 1; cd.relu = 0; cd.has_residual = 0; cd.res_stride = 1;
        cd.out_dtype = adt; cd.in_dtype = adt;
        const std::string pre = root + "/" + sc + "/preact";
        L.p_w = add_param(un + "/shortcut/W", METRO_PARAM_CONV_W, root + "/" + sc + "/shortcut", "", METRO_F16, c_sc, 1, 1, c_in, 1, c_in);
        const int w2 = add_param(un + "/conv1/W", METRO_PARAM_CONV_W, root + "/" + sc + "/conv1", root + "/" + sc + "/conv1/BatchNorm",
                                 METRO_F16, cb, 1, 1, c_in, 1, c_in);
        L.p_bias = add_param(un + "/shortcut/bias", METRO_PARAM_BIAS, root + "/" + sc + "/shortcut", "", METRO_F32, c_sc, 1, 1, 1, 1, 1);
        const int b2 = add_param(un + "/conv1/bias", METRO_PARAM_BIAS, root + "/" + sc + "/conv1", root + "/" + sc + "/conv1/BatchNorm",
                                 METRO_F32, cb, 1, 1, 1, 1, 1);
        // contiguity (sizes are multiples of the 256-byte blob alignment for c_sc % 256 == 0, c_in % 64 == 0)
        if (p->params[w2].offset != p->params[L.p_w].offset + p->params[L.p_w].bytes ||
            p->params[b2].offset != p->params[L.p_bias].offset + p->params[L.p_bias].bytes) {
            set_error("internal: fused pair parameters of %s are not contiguous in the blob", un.c_str());
            return METRO_ERR_STATE;       // the kernel reads conv1's rows at w + c_sc * c_in: never launch on a broken layout
        }
        L.p_scale = add_param(un + "/shortcut/pro_scale", METRO_PARAM_PRO_SCALE, "", pre, METRO_F16, c_in, 1, 1, 1, 1, 1);
        L.p_shift = add_param(un + "/shortcut/pro_shift", METRO_PARAM_PRO_SHIFT, "", pre, METRO_F16, c_in, 1, 1, 1, 1, 1);
        L.in_slot = in_slot; L.out_slot = S_SC; L.res_slot = S_NONE;
        L.split = c_sc; L.c_out2 = cb; L.relu2 = 1; L.out2_slot = S_T1;
        need(S_SC, (int64_t)side * side * c_sc * 2);
        need(S_T1, (int64_t)side * side * cb * 2);
        fill_info(L, un + "/shortcut+conv1", 2.0 * side * side * (double)(c_sc + cb) * c_in);
        L.info.c_out = c_sc;      // the primary output tensor (S_SC) has c_sc channels
        p->layers.push_back(L);
        return METRO_OK;
    }

    // Appends conv1 of the NEXT unit (1x1, cb outputs, folded BN + ReLU, pre-activation prologue of that
    // unit) to the conv3 layer just added: the kernel runs it as a second GEMM on the LDS-resident output
    // tile (reference resnet_v2.py:119,127-128 of unit u+1).  Parameter names stay those of the next unit.
    void fuse_next_conv1(const std::string& un_next, const std::string& sc_next, int side, int c_in, int cb) {
        Layer& L = p->layers.back();
        const std::string conv_var = root + "/" + sc_next + "/conv1";
        const std::string bn_var = conv_var + "/BatchNorm";
        const std::string pv = root + "/" + sc_next + "/preact";
        const std::string ln = un_next + "/conv1";
        L.f2_w = add_param(ln + "/W", METRO_PARAM_CONV_W, conv_var, bn_var, METRO_F16, cb, 1, 1, c_in, 1, c_in);
        L.f2_bias = add_param(ln + "/bias", METRO_PARAM_BIAS, conv_var, bn_var, METRO_F32, cb, 1, 1, 1, 1, 1);
        L.f2_scale = add_param(ln + "/pro_scale", METRO_PARAM_PRO_SCALE, "", pv, METRO_F16, c_in, 1, 1, 1, 1, 1);
        L.f2_shift = add_param(ln + "/pro_shift", METRO_PARAM_PRO_SHIFT, "", pv, METRO_F16, c_in, 1, 1, 1, 1, 1);
        L.f2_c2 = cb; L.out2_slot = S_T1;
        need(S_T1, (int64_t)side * side * cb * 2);
        const double flops = 2.0 * side * side * (double)cb * c_in;
        const std::string name = std::string(L.info.name) + "+" + un_next.substr(un_next.find('/') + 1) + "/conv1";
        snprintf(L.info.name, sizeof(L.info.name), "%s", name.c_str());
        L.info.flops_per_image += flops;
        p->flops_per_image += flops;
    }

    // conv1 of the unit (1x1, cb outputs, folded BN + ReLU, on the pre-activated unit input) fused in FRONT of the conv2 layer
    // just added, whose input becomes the unit's raw input (reference resnet_v2.py:119,127-132).
    void fuse_conv1_in_front(const std::string& un, const std::string& sc, int in_slot, int side, int c_in, int cb) {
        Layer& L = p->layers.back();
        const std::string conv_var = root + "/" + sc + "/conv1";
        const std::string bn_var = conv_var + "/BatchNorm";
        const std::string pv = root + "/" + sc + "/preact";
        L.p1_w = add_param(un + "/conv1/W", METRO_PARAM_CONV_W, conv_var, bn_var, METRO_F16, cb, 1, 1, c_in, 1, c_in);
        L.p1_bias = add_param(un + "/conv1/bias", METRO_PARAM_BIAS, conv_var, bn_var, METRO_F32, cb, 1, 1, 1, 1, 1);
        L.p1_scale = add_param(un + "/conv1/pro_scale", METRO_PARAM_PRO_SCALE, "", pv, METRO_F16, c_in, 1, 1, 1, 1, 1);
        L.p1_shift = add_param(un + "/conv1/pro_shift", METRO_PARAM_PRO_SHIFT, "", pv, METRO_F16, c_in, 1, 1, 1, 1, 1);
        L.in_slot = in_slot;
        const double flops = 2.0 * side * side * (double)cb * c_in;
        snprintf(L.info.name, sizeof(L.info.name), "%s/conv1+conv2", un.c_str());
        L.info.flops_per_image += flops;
        L.info.fused_flags |= METRO_FUSED_CONV1_IN_FRONT;
        p->flops_per_image += flops;
    }

    // The unit's projection shortcut (1x1, c_out outputs, bias, on the pre-activated unit input) computed inside the conv3
    // layer just added instead of being read from a tensor (reference resnet_v2.py:119,122-125,138).
    void fuse_projection_shortcut(const std::string& un, const std::string& sc, int x_slot, int side, int c_in, int c_out) {
        Layer& L = p->layers.back();
        const std::string conv_var = root + "/" + sc + "/shortcut";
        const std::string pv = root + "/" + sc + "/preact";
        L.psc_w = add_param(un + "/shortcut/W", METRO_PARAM_CONV_W, conv_var, "", METRO_F16, c_out, 1, 1, c_in, 1, c_in);
        L.psc_bias = add_param(un + "/shortcut/bias", METRO_PARAM_BIAS, conv_var, "", METRO_F32, c_out, 1, 1, 1, 1, 1);
        L.psc_scale = add_param(un + "/shortcut/pro_scale", METRO_PARAM_PRO_SCALE, "", pv, METRO_F16, c_in, 1, 1, 1, 1, 1);
        L.psc_shift = add_param(un + "/shortcut/pro_shift", METRO_PARAM_PRO_SHIFT, "", pv, METRO_F16, c_in, 1, 1, 1, 1, 1);
        L.psc_slot = x_slot;
        const double flops = 2.0 * side * side * (double)c_out * c_in;
        L.info.flops_per_image += flops;
        L.info.fused_flags |= METRO_FUSED_PROJECTION_SHORTCUT;
        p->flops_per_image += flops;
    }

    void fill_info(Layer& L, const std::string& lname, double flops) {
        MetroLayerInfo& I = L.info;
        snprintf(I.name, sizeof(I.name), "%s", lname.c_str());
        I.kind = L.kind;
        const MetroConvDesc& cd = L.cd;
        I.h_in = cd.h_in; I.w_in = cd.w_in; I.c_in = cd.c_in; I.h_out = cd.h_out; I.w_out = cd.w_out;
        I.c_out = cd.c_out; I.kh = cd.kh; I.kw = cd.kw; I.stride = cd.stride; I.dilation = cd.dilation;
        I.pad_top = cd.pad_top; I.pad_left = cd.pad_left;
        I.has_prologue = cd.has_prologue; I.relu = cd.relu; I.has_residual = cd.has_residual;
        I.res_stride = cd.res_stride; I.res_offset = cd.res_offset;
        I.out_dtype = cd.out_dtype;
        I.flops_per_image = flops;
        p->flops_per_image += flops;
    }
};

// TF 'SAME' padding: out = ceil(in/s); total = max((out-1)*s + k_eff - in, 0); beg = total/2
int tf_same_pad_beg(int in, int k_eff, int s) {
    const int out = (in + s - 1) / s;
    const int total = std::max((out - 1) * s + k_eff - in, 0);
    return total / 2;
}

int build_plan(MetroPlan* p) {
    const MetroSpec& sp = p->spec;
    Builder B{p, std::string("MainPart/resnet_v2_") + std::to_string(sp.arch)};
    const bool fast = p->fast;
    const int adt = p->act_dtype;
    const int aes = p->act_bytes;
    const int ldt = sp.precision == METRO_PREC_F64 ? METRO_F64 : METRO_F32;   // logits dtype
    const int side = sp.proc_side;
    const int bw = sp.base_width;

    // ---- root block: conv1 7x7/2 with explicit pad 3 (+bias, no BN, no ReLU), pool1 ----------
    // reference resnet_v2.py:219-224, resnet_utils.py:125-135,177-185
    const int s2 = (side + 6 - 7) / 2 + 1;   // 128
    bool fused_stem_pool = false;
    // the fused stem+pool kernel can read the fp32 crops directly (cast + border on the way into LDS)
    const bool raw_stem = fast && stem_pool_f32in_supported(side, bw);
    if (fast) {
        Layer L;
        memset(&L, 0, sizeof(L));
        L.f2_w = L.f2_bias = L.f2_scale = L.f2_shift = -1; L.p1_w = L.p1_bias = L.p1_scale = L.p1_shift = -1; L.psc_w = L.psc_bias = L.psc_scale = L.psc_shift = -1; L.psc_slot = S_NONE; L.reb_w = L.reb_bias = -1; L.reb_slot = L.sub_slot = S_NONE;
        L.kind = LK_PREP;
        L.cd.h_in = L.cd.w_in = side; L.cd.c_in = 3;
        L.cd.h_out = side + 6; L.cd.w_out = side + 8; L.cd.c_out = 4; L.cd.out_dtype = METRO_F16;
        L.in_slot = S_IMAGES; L.out_slot = S_PREP; L.res_slot = S_NONE;
        L.p_w = L.p_bias = L.p_scale = L.p_shift = -1;
        if (!raw_stem) {
            B.need(S_PREP, (int64_t)(side + 6) * (side + 8) * 4 * 2);
            B.fill_info(L, "prep_input", 0.0);
            p->layers.push_back(L);
        }

        // stem as a pad-free 7x1-tap conv over the bordered 4-channel image: each tap = 8
        // pixels x 4 channels = 32 contiguous fp16; weights packed [c_out][7][8][4] (zeros in
        // the 8th pixel and the 4th channel).
        Layer S;
        memset(&S, 0, sizeof(S));
        S.f2_w = S.f2_bias = S.f2_scale = S.f2_shift = -1; S.p1_w = S.p1_bias = S.p1_scale = S.p1_shift = -1; S.psc_w = S.psc_bias = S.psc_scale = S.psc_shift = -1; S.psc_slot = S_NONE; S.reb_w = S.reb_bias = -1; S.reb_slot = S.sub_slot = S_NONE;
        S.kind = LK_CONV;
        MetroConvDesc& cd = S.cd;
        cd.h_in = side + 6; cd.w_in = side + 8; cd.c_in = 32; cd.in_pix_stride = 4;
        cd.h_out = cd.w_out = s2; cd.c_out = bw;
        cd.kh = 7; cd.kw = 1; cd.stride = 2; cd.dilation = 1; cd.pad_top = cd.pad_left = 0;
        cd.out_dtype = adt; cd.in_dtype = METRO_F16;
        const std::string cv = B.root + "/conv1";
        S.p_w = B.add_param("conv1/W", METRO_PARAM_CONV_W, cv, "", METRO_F16, bw, 7, 7, 3, 8, 4);
        S.p_bias = B.add_param("conv1/bias", METRO_PARAM_BIAS, cv, "", METRO_F32, bw, 1, 1, 1, 1, 1);
        S.p_scale = S.p_shift = -1;
        S.in_slot = S_PREP; S.out_slot = S_STEM; S.res_slot = S_NONE;
        if (stem_pool_f16_supported(side, bw)) {
            // reference resnet_v2.py:219-224: the pooled tensor is the only thing block1 reads
            fused_stem_pool = true;
            S.stem_pool = raw_stem ? 2 : 1;
            if (raw_stem) S.in_slot = S_IMAGES;
            S.out_slot = S_X0;
            const int s4f = (s2 + 2 - 3) / 2 + 1;
            B.need(S_X0, (int64_t)s4f * s4f * bw * aes);
            B.fill_info(S, "conv1+pool1", 2.0 * s2 * s2 * bw * 7 * 7 * 3);
            S.info.h_out = S.info.w_out = s4f;
            S.cd.h_out = S.cd.w_out = s4f;        // shape of the stored tensor (forward_upto, out_bytes_per_image)
        } else {
            B.need(S_STEM, (int64_t)s2 * s2 * bw * aes);
            B.fill_info(S, "conv1", 2.0 * s2 * s2 * bw * 7 * 7 * 3);
        }
        p->layers.push_back(S);
    } else {
        B.add_conv("conv1", "conv1", "", "", S_IMAGES, S_STEM, S_NONE, side, 3, s2, bw, 7, 2, 1, 3,
                   false, 0, 1, 0, adt, METRO_F32);
    }
    const int s4 = (s2 + 2 - 3) / 2 + 1;     // 64
    if (!fused_stem_pool) {
        Layer L;
        memset(&L, 0, sizeof(L));
        L.f2_w = L.f2_bias = L.f2_scale = L.f2_shift = -1; L.p1_w = L.p1_bias = L.p1_scale = L.p1_shift = -1; L.psc_w = L.psc_bias = L.psc_scale = L.psc_shift = -1; L.psc_slot = S_NONE; L.reb_w = L.reb_bias = -1; L.reb_slot = L.sub_slot = S_NONE;
        L.kind = LK_POOL;
        L.cd.h_in = L.cd.w_in = s2; L.cd.c_in = bw; L.cd.h_out = L.cd.w_out = s4; L.cd.c_out = bw;
        L.cd.kh = L.cd.kw = 3; L.cd.stride = 2; L.cd.dilation = 1; L.cd.pad_top = L.cd.pad_left = 1;
        L.cd.out_dtype = adt;
        L.in_slot = S_STEM; L.out_slot = S_X0; L.res_slot = S_NONE;
        L.p_w = L.p_bias = L.p_scale = L.p_shift = -1;
        B.need(S_X0, (int64_t)s4 * s4 * bw * aes);
        B.fill_info(L, "pool1", 0.0);
        p->layers.push_back(L);
    }

    // ---- block table (reference resnet_v2.py:272-312) ---------------------------------------
    bool centered[3] = {false, false, false};
    if (sp.centered_stride) {
        if (sp.arch == 50) {
            const int i_last = (int)std::lround(std::log2((double)sp.stride)) - 3;   // :279-281
            if (i_last >= 0 && i_last < 3) centered[i_last] = true;
        } else {
            int i_last = (int)std::log2((double)sp.stride) - 3;                      // :301-302
            if (i_last < 0) i_last += 3;   // Python c[-1]
            if (i_last >= 0 && i_last < 3) centered[i_last] = true;
        }
    }
    const int n_units[4] = {3, 4, sp.arch == 50 ? 6 : 23, 3};
    const int base[4] = {bw, 2 * bw, 4 * bw, 8 * bw};
    const int block_stride[4] = {2, 2, 2, 1};

    // ---- stack_blocks_dense (reference resnet_utils.py:307-348) ------------------------------
    const double output_stride = sp.stride / 4.0;    // resnet_v2.py:215 (float division)
    int current_stride = 1, rate = 1;
    int cur_side = s4, cur_c = bw, cur = S_X0;
    bool conv1_done = false;      // conv1 of this unit already ran inside the previous unit's conv3 launch
    // block1 without its 256-channel residual stream in HBM (round 5): 0 = off, 1 = unit 1 planned (x_1 stays on chip), 2 = unit 2
    // planned (x_1 rebuilt in its conv3 launch; x_2 on chip / sub-sampled / stored).  chain_l1 = layer index of unit 1's conv3
    // launch, chain_x0 = slot of the block's input, cur_compact = `cur` holds the sub-sampled compact copy of the unit input.
    int chain = 0, chain_l1 = -1, chain_x0 = S_NONE;
    bool cur_compact = false;
    for (int b = 0; b < 4; ++b) {
        for (int u = 1; u <= n_units[b]; ++u) {
            const int unit_stride = u == n_units[b] ? block_stride[b] : 1;   // resnet_v2.py:260-269
            const bool unit_centered = u == n_units[b] && b < 3 && centered[b];
            int s, r;
            if ((double)current_stride == output_stride) {
                s = 1; r = rate; rate *= unit_stride;                        // :325-327
            } else {
                s = unit_stride; r = 1; current_stride *= unit_stride;       // :329-333
                if ((double)current_stride > output_stride) { set_error("The target output_stride cannot be reached."); return METRO_ERR_INVALID_ARG; }
            }
            const int cb = base[b], cout = 4 * base[b];
            const int side_out = s == 2 ? (cur_side + 1) / 2 : cur_side;
            const std::string un = "block" + std::to_string(b + 1) + "/unit_" + std::to_string(u);
            const std::string sc = un + "/bottleneck_v2";
            const int shift = (unit_centered && s == 2) ? 1 : 0;             // resnet_v2.py:113-115
            const int nxt = cur == S_X0 ? S_X1 : S_X0;
            const bool project = cur_c != cout;                              // resnet_v2.py:120-125
            // measured on MI355X (batch 64): pays when conv1 fills whole 128-cout tiles and the pair is not huge
            // (block2/block3 of ResNet-50/101: -9 / -7 us); block1 (cb = 64: a half-empty tile) and block4 lose
            // block1 (cb = 64: a half-empty tile in the tiled kernel) pairs only in the persistent kernel
            bool pw_pair = false;
            if (fast && project && s == 1 && cur_c == 64 && cout == 256 && cb == 64) {
                MetroConvDesc probe;
                memset(&probe, 0, sizeof(probe));
                probe.n = 1; probe.h_in = probe.w_in = probe.h_out = probe.w_out = cur_side;
                probe.c_in = probe.in_pix_stride = cur_c; probe.c_out = cout + cb;
                probe.kh = probe.kw = 1; probe.stride = 1; probe.dilation = 1; probe.has_prologue = 1;
                probe.out_dtype = probe.in_dtype = METRO_F16; probe.res_stride = 1;
                pw_pair = conv_pw64_supported(probe, 1);
            }
            // block1/unit_1 (64-channel input): conv1 in front of conv2 inside the weight-resident 3x3 kernel, the projection
            // shortcut inside the conv3 (+ next conv1) launch: two launches, no shortcut / t1 tensors.  The choice must hold for
            // EVERY batch the plan may run (the layer list is fixed): probed at n = 1.
            bool unit_fused = false;
            if (fast && project && s == 1 && r == 1 && cur_c == 64 && cb == 64 && cout == 256 && u < n_units[b] && !conv1_done &&
                tuning_knob("METRO_UNIT1_FUSED", 1)) {
                MetroConvDesc c2, c3;
                memset(&c2, 0, sizeof(c2));
                c2.n = 1; c2.h_in = c2.w_in = c2.h_out = c2.w_out = cur_side; c2.c_in = c2.in_pix_stride = c2.c_out = 64;
                c2.kh = c2.kw = 3; c2.stride = 1; c2.dilation = 1; c2.pad_top = c2.pad_left = 1; c2.relu = 1;
                c2.out_dtype = c2.in_dtype = METRO_F16; c2.res_stride = 1;
                c3 = c2;
                c3.kh = c3.kw = 1; c3.pad_top = c3.pad_left = 0; c3.relu = 0; c3.c_out = 256;
                unit_fused = conv3x3_c64_supported(c2) && conv_pw64_supported(c3, 3);
            }
            // ... and none of the block's 256-channel sums in HBM (conv_pw64 REB / OUTM): three units, the second one plain
            bool chain_start = false;
            if (unit_fused && u == 1 && n_units[b] == 3 && tuning_knob("METRO_B1_REBUILD", 1)) {
                MetroConvDesc c3;
                memset(&c3, 0, sizeof(c3));
                c3.n = 1; c3.h_in = c3.w_in = c3.h_out = c3.w_out = cur_side; c3.c_in = c3.in_pix_stride = 64; c3.c_out = 256;
                c3.kh = c3.kw = 1; c3.stride = 1; c3.dilation = 1; c3.out_dtype = c3.in_dtype = METRO_F16; c3.res_stride = 1;
                chain_start = conv_pw64_supported(c3, 4);
            }
            const bool chain_mid = chain == 1 && u == 2 && conv1_done && !project && s == 1 && r == 1;
            if (chain == 1 && !chain_mid) { set_error("internal: block1 rebuild chain planned for a unit 2 that is not plain"); return METRO_ERR_STATE; }
            const int nxt_slot = chain_mid ? S_X1 : nxt;      // unit 2 of the chain: S_X0 still holds x0, which its launch reads
            // (round 5: block4's pair too -- 1024 -> 2048 + 512 in one conv_gemm4w launch: the 134 MB input read once, one launch
            // fewer; same-box A/B batch 256 -0.4 %, batch 64 0; 1024 = the round-4 plan)
            static const int pair_max_cout = tuning_knob("METRO_PAIR_MAX_COUT", 2048);
            const bool fuse_pair = fast && project && s == 1 && cout % 256 == 0 && cur_c % 64 == 0 && !unit_fused &&
                                   ((cb % 128 == 0 && cout <= pair_max_cout) || pw_pair);
            if (unit_fused) {
                // conv1 runs inside the conv2 launch below
            } else if (conv1_done) {
                conv1_done = false;       // S_T1 already holds relu(bn(conv1(preact(x))))
            } else if (fuse_pair) {
                const int pst = B.add_shortcut_conv1_pair(un, sc, cur, cur_side, cur_c, cout, cb, adt);
                if (pst != METRO_OK) return pst;
            } else {
                if (project) {
                    // conv1x1(shift(preact), stride s) + bias: input pixel = shift + s*ho
                    B.add_conv(un + "/shortcut", sc + "/shortcut", "", sc + "/preact", cur, S_SC, S_NONE,
                               cur_side, cur_c, side_out, cout, 1, s, 1, -shift, false, 0, 1, 0, adt, adt);
                }
                // conv1: 1x1 on preact, BN+ReLU folded (resnet_v2.py:127-128)
                B.add_conv(un + "/conv1", sc + "/conv1", sc + "/conv1/BatchNorm", sc + "/preact", cur, S_T1,
                           S_NONE, cur_side, cur_c, cur_side, cb, 1, 1, 1, 0, true, 0, 1, 0, adt, adt);
            }
            // conv2: conv2d_same 3x3 (resnet_utils.py:82-135)
            const int k_eff = 3 + 2 * (r - 1);
            const int pad_beg = (s == 1 || unit_centered) ? tf_same_pad_beg(cur_side, k_eff, s)
                                                          : (k_eff - 1) / 2;
            const int t2_slot = chain_start ? S_T2B : S_T2;   // unit 1 of the chain: its conv2 output is read again by unit 2's launch
            B.add_conv(un + "/conv2", sc + "/conv2", sc + "/conv2/BatchNorm", "", S_T1, t2_slot, S_NONE,
                       cur_side, cb, side_out, cb, 3, s, r, pad_beg, true, 0, 1, 0, adt, adt);
            if (unit_fused) B.fuse_conv1_in_front(un, sc, cur, cur_side, cur_c, cb);
            // conv3 + bias + shortcut (resnet_v2.py:134-138)
            if (unit_fused) {
                B.add_conv(un + "/conv3", sc + "/conv3", "", "", t2_slot, nxt, S_NONE, side_out, cb, side_out, cout, 1, 1, 1, 0, false,
                           0, 1, 0, adt, adt);
                B.fuse_projection_shortcut(un, sc, cur, cur_side, cur_c, cout);
                if (chain_start) {
                    Layer& L3 = p->layers.back();
                    L3.out_mode = 1;                                   // x_1 only feeds unit 2's conv1, inside this launch
                    L3.info.fused_flags |= METRO_FUSED_OUT_ON_CHIP;
                    chain = 1; chain_l1 = (int)p->layers.size() - 1; chain_x0 = cur;
                }
            } else if (chain_mid) {
                // identity shortcut x_1 (resnet_v2.py:120-121 with stride 1) rebuilt in the launch: no residual tensor is read
                B.add_conv(un + "/conv3", sc + "/conv3", "", "", S_T2, nxt_slot, S_NONE, side_out, cb, side_out, cout, 1, 1, 1, 0, false,
                           0, 1, 0, adt, adt);
                Layer& L3 = p->layers.back();
                const Layer& L1 = p->layers[chain_l1];
                L3.info.has_residual = 1; L3.info.res_stride = 1; L3.info.res_offset = 0;    // the reference's shortcut, as for any unit
                L3.info.fused_flags |= METRO_FUSED_REBUILT_SHORTCUT;
                L3.reb_w = L1.p_w; L3.reb_bias = L1.p_bias; L3.reb_slot = L1.in_slot;
                L3.psc_w = L1.psc_w; L3.psc_bias = L1.psc_bias; L3.psc_scale = L1.psc_scale; L3.psc_shift = L1.psc_shift; L3.psc_slot = chain_x0;
                // what the last unit reads of this sum: every pixel (it runs at stride 1: stride-4 nets), or every second one
                const bool next_strided = (double)current_stride != output_stride;      // resnet_utils.py:325-333 for unit 3
                if (next_strided) {
                    L3.out_mode = 2; L3.sub_slot = S_X1; L3.sub_off = (b < 3 && centered[b]) ? 1 : 0; L3.sub_side = (cur_side + 1) / 2;
                    L3.out_slot = S_SC;                                // metro_forward_upto stopping here writes the whole sum
                    L3.info.fused_flags |= METRO_FUSED_OUT_ON_CHIP;
                    B.need(S_SC, (int64_t)side_out * side_out * cout * aes);
                    cur_compact = true;
                }
                chain = 2;
            } else if (project)
                B.add_conv(un + "/conv3", sc + "/conv3", "", "", S_T2, nxt, S_SC, side_out, cb, side_out,
                           cout, 1, 1, 1, 0, false, side_out, 1, 0, adt, adt);
            else {
                B.add_conv(un + "/conv3", sc + "/conv3", "", "", S_T2, nxt, cur, side_out, cb, side_out,
                           cout, 1, 1, 1, 0, false, cur_side, s, shift, adt, adt);
                if (cur_compact) {
                    // the previous launch wrote exactly the pixels this unit's sub-sampled shortcut reads, compactly: the launch
                    // adds them pixel for pixel (the info keeps the reference's gather: stride s, offset shift)
                    if (s != 2) { set_error("internal: compact shortcut planned for a unit that is not strided"); return METRO_ERR_STATE; }
                    Layer& L3 = p->layers.back();
                    L3.cd.res_h = L3.cd.res_w = side_out; L3.cd.res_stride = 1; L3.cd.res_offset = 0;
                    L3.info.fused_flags |= METRO_FUSED_COMPACT_SHORTCUT;
                    cur_compact = false;
                }
                if (chain == 2) chain = 0;
            }
            // block1 (full 256-channel rows per pixel tile): conv1 of the next unit rides in this launch
            if (fast && u < n_units[b] && s == 1) {
                MetroConvDesc probe = p->layers.back().cd;
                probe.n = 1;
                // block2 (128 -> 512 on 32-wide maps): the persistent kernel with all 512 channels of a pixel tile in one block
                const bool pw_next = probe.c_in == 128 && cb == 128 && conv_pw64_supported(probe, 2);
                // chain_mid: unit 2 of block1's rebuild chain has no residual tensor to read -- its identity shortcut exists only
                // inside the fused conv3 + next-conv1 launch (conv_pw64 REB), so the fusion is not optional there (METRO_FUSE2=0
                // used to drop the shortcut silently: ADVICE r5)
                if (conv_f16_fuse2_supported(probe, cb) || unit_fused || chain_mid || pw_next) {
                    const std::string un2 = "block" + std::to_string(b + 1) + "/unit_" + std::to_string(u + 1);
                    B.fuse_next_conv1(un2, un2 + "/bottleneck_v2", side_out, cout, cb);
                    conv1_done = true;
                }
            }
            cur = nxt_slot; cur_side = side_out; cur_c = cout;
        }
        if (chain != 0 || cur_compact) { set_error("internal: block%d ended inside a rebuild chain", b + 1); return METRO_ERR_STATE; }
    }
    if ((double)current_stride != output_stride) { set_error("The target output_stride cannot be reached."); return METRO_ERR_INVALID_ARG; }
    if (cur_side != sp.proc_side / sp.stride) { set_error("internal: output side %d != %d", cur_side, sp.proc_side / sp.stride); return METRO_ERR_STATE; }

    // ---- postnorm (prologue) + logits 1x1 (+bias), fp32 out (resnet_v2.py:229-236, architectures.py:34)
    const int c_head = sp.depth * sp.n_joints_head;
    B.add_conv("logits", "logits", "", "postnorm", cur, S_LOGITS, S_NONE, cur_side, cur_c, cur_side,
               c_head, 1, 1, 1, 0, false, 0, 1, 0, ldt, adt);
    // fp16 mode: the logits stay on chip (volumetric.py:227-235 starts in the GEMM's epilogue)
    const bool head_fused = fast && head_f16_supported(cur_c, c_head, sp.n_joints_head, sp.depth, cur_side);
    p->layers.back().head_fused = head_fused ? 1 : 0;

    // ---- soft-argmax + decode ---------------------------------------------------------------
    {
        Layer L;
        memset(&L, 0, sizeof(L));
        L.f2_w = L.f2_bias = L.f2_scale = L.f2_shift = -1; L.p1_w = L.p1_bias = L.p1_scale = L.p1_shift = -1; L.psc_w = L.psc_bias = L.psc_scale = L.psc_shift = -1; L.psc_slot = S_NONE; L.reb_w = L.reb_bias = -1; L.reb_slot = L.sub_slot = S_NONE;
        L.kind = LK_SOFTARGMAX;
        L.cd.h_in = L.cd.w_in = cur_side; L.cd.c_in = c_head; L.cd.h_out = 1; L.cd.w_out = sp.n_joints_out;
        L.cd.c_out = 3; L.cd.out_dtype = METRO_F32;
        L.in_slot = S_LOGITS; L.out_slot = S_NONE; L.res_slot = S_NONE;
        L.p_w = L.p_bias = L.p_scale = L.p_shift = -1;
        L.head_fused = head_fused ? 1 : 0;
        L.head_c_in = cur_c;
        B.fill_info(L, "softargmax", 0.0);
        p->layers.push_back(L);
    }

    // ---- workspace layout ------------------------------------------------------------------
    int64_t off = 0;
    for (int s = 0; s < S_COUNT; ++s) {
        p->slot_offset[s] = off;
        int64_t bytes = p->slot_bytes_per_image[s] * p->max_batch;
        if (s == S_PART) {
            const int hs = sp.proc_side / sp.stride;
            bytes = softargmax_scratch_bytes(p->max_batch, hs, sp.n_joints_head);
            // the one-launch head writes one (m, S, Sx, Sy, Sz) fp32 record per (image, 64-pixel slab, joint): more slabs
            // than the two-launch path's <= 64 once the heat map has > 4096 pixels (side >= 96, e.g. proc_side 384 at stride 4)
            if (head_fused)
                bytes = std::max(bytes, (int64_t)p->max_batch * head_f16_slabs(hs) * sp.n_joints_head * 5 * 4);
        }
        if (s == S_STATUS) bytes = (int64_t)p->max_batch * 4;      // int32 per image: the finalize launch's non-finite screen
        off = align_up(off + bytes, 256);
    }
    p->workspace_bytes = off;
    for (Layer& L : p->layers) {
        L.info.out_offset = L.out_slot >= 0 ? p->slot_offset[L.out_slot] : -1;
        L.info.out_sub_offset = L.out_mode == 2 ? p->slot_offset[L.sub_slot] : -1;
        L.info.out_sub_side = L.out_mode == 2 ? L.sub_side : 0;
        L.info.out_sub_off = L.out_mode == 2 ? L.sub_off : 0;
        const int64_t es = L.cd.out_dtype == METRO_F16 ? 2 : L.cd.out_dtype == METRO_F32 ? 4 : 8;
        L.info.out_bytes_per_image = (int64_t)L.cd.h_out * L.cd.w_out * (L.split > 0 ? L.split : L.cd.c_out) * es;
        const bool two = L.kind == LK_CONV && (L.split > 0 || L.f2_w >= 0);
        L.info.out2_offset = two ? p->slot_offset[L.out2_slot] : -1;
        L.info.out2_channels = two ? (L.split > 0 ? L.c_out2 : L.f2_c2) : 0;
        if (L.p1_w >= 0) {      // conv1+conv2: conv1's output is a DUMP-ONLY second tensor (metro_forward_upto stopping here), not traffic
            L.info.out2_offset = p->slot_offset[S_T1];
            L.info.out2_channels = p->params[L.p1_w].c_out;
        }
        // algorithmic bytes: every tensor the launch touches, once
        const int64_t in_es = L.in_slot == S_IMAGES ? 4 : (L.kind == LK_SOFTARGMAX ? (sp.precision == METRO_PREC_F64 ? 8 : 4)
                                                           : (L.kind == LK_CONV && !fast ? aes : (L.kind == LK_CONV ? 2 : aes)));
        int64_t act = 0;
        if (L.in_slot == S_IMAGES) act += (int64_t)sp.proc_side * sp.proc_side * 3 * 4;
        else if (L.kind == LK_CONV && L.cd.in_pix_stride != L.cd.c_in) act += (int64_t)L.cd.h_in * L.cd.w_in * L.cd.in_pix_stride * in_es;
        else act += (int64_t)L.cd.h_in * L.cd.w_in * L.cd.c_in * in_es;
        if (L.kind == LK_SOFTARGMAX) act += (int64_t)sp.n_joints_out * 3 * 4;
        else if (L.out_mode == 0) act += L.info.out_bytes_per_image;
        else if (L.out_mode == 2) act += (int64_t)L.sub_side * L.sub_side * L.cd.c_out * es;     // the sub-sampled copy only
        if (L.reb_w >= 0) act += (int64_t)L.cd.h_out * L.cd.w_out * p->params[L.reb_w].c_in * es;  // the previous unit's conv2 output
        if (two) act += (int64_t)L.cd.h_out * L.cd.w_out * L.info.out2_channels * es;
        if (L.kind == LK_CONV && L.cd.has_residual) act += (int64_t)L.cd.h_out * L.cd.w_out * L.cd.c_out * es;
        if (L.psc_w >= 0) act += (int64_t)L.cd.h_out * L.cd.w_out * p->params[L.psc_w].c_in * es;      // the unit input, read for the shortcut
        if (L.head_fused) {       // logits never reach HBM: the launch writes / the finalize reads the per-slab statistics
            const int64_t part = (int64_t)head_f16_slabs(sp.proc_side / sp.stride) * sp.n_joints_head * 5 * 4;
            if (L.kind == LK_CONV) act = (int64_t)L.cd.h_in * L.cd.w_in * L.cd.c_in * 2 + part;
            else act = part + (int64_t)sp.n_joints_out * 3 * 4;
        }
        L.info.algo_act_bytes_per_image = act;
        int64_t pb = 0;
        for (int idx : {L.p_w, L.p_bias, L.p_scale, L.p_shift, L.f2_w, L.f2_bias, L.f2_scale, L.f2_shift, L.p1_w, L.p1_bias, L.p1_scale,
                        L.p1_shift, L.psc_w, L.psc_bias, L.psc_scale, L.psc_shift, L.reb_w, L.reb_bias})
            if (idx >= 0) pb += p->params[idx].bytes;
        if (L.split > 0) pb += p->params[L.p_w + 1].bytes + p->params[L.p_bias + 1].bytes;   // conv1 rows of a fused pair
        L.info.algo_param_bytes = pb;
    }
    return METRO_OK;
}

// One layer of the plan at batch n.  `dump` (metro_forward_upto stopping at this layer): launches whose intermediate tensors live on
// chip also write them out -- the fp32 logits of the one-launch head, conv1's output of a conv1+conv2 launch.
int launch_layer(const MetroPlan* p, const char* d_params, int li, const float* images, int n, float* poses, char* ws, hipStream_t stream, bool dump) {
    const Layer& L = p->layers[li];
    auto slot_ptr = [&](int slot) -> void* {
        if (slot == S_IMAGES) return const_cast<float*>(images);
        if (slot < 0) return nullptr;
        return ws + p->slot_offset[slot];
    };
    auto prm = [&](int idx) -> const void* { return idx < 0 ? nullptr : d_params + p->params[idx].offset; };
    switch (L.kind) {
        case LK_PREP:
            return launch_prep_input_f16(images, n, p->spec.proc_side, slot_ptr(L.out_slot), stream);
        case LK_POOL:
            return launch_maxpool(slot_ptr(L.in_slot), slot_ptr(L.out_slot), n, L.cd.h_in, L.cd.w_in, L.cd.c_in, p->act_dtype, stream);
        case LK_CONV: {
            MetroConvDesc cd = L.cd;
            cd.n = n;
            if (p->fast && L.head_fused) {
                float* logits_dump = dump ? static_cast<float*>(slot_ptr(L.out_slot)) : nullptr;
                return launch_head_f16(slot_ptr(L.in_slot), prm(L.p_w), static_cast<const float*>(prm(L.p_bias)), prm(L.p_scale),
                                       prm(L.p_shift), n, L.cd.c_in, L.cd.c_out, p->spec.n_joints_head, p->spec.depth, L.cd.h_in,
                                       static_cast<float*>(slot_ptr(S_PART)), logits_dump, stream);
            }
            if (p->fast && L.stem_pool == 2)
                return launch_stem_pool_f32in(images, prm(L.p_w), static_cast<const float*>(prm(L.p_bias)), slot_ptr(L.out_slot), n,
                                              p->spec.proc_side, stream);
            if (p->fast && L.stem_pool)
                return launch_stem_pool_f16(slot_ptr(L.in_slot), prm(L.p_w), static_cast<const float*>(prm(L.p_bias)),
                                            slot_ptr(L.out_slot), n, p->spec.proc_side, stream);
            if (p->fast && L.split > 0) {
                ConvSplit sp;
                sp.split = L.split; sp.c_out2 = L.c_out2; sp.relu2 = L.relu2; sp.out2 = slot_ptr(L.out2_slot);
                return launch_conv_f16_dma(cd, slot_ptr(L.in_slot), prm(L.p_w), static_cast<const float*>(prm(L.p_bias)),
                                           prm(L.p_scale), prm(L.p_shift), nullptr, slot_ptr(L.out_slot), stream, &sp);
            }
            if (p->fast && L.f2_w >= 0) {
                ConvFuse2 f2;
                f2.w2 = prm(L.f2_w); f2.bias2 = static_cast<const float*>(prm(L.f2_bias));
                f2.scale2 = prm(L.f2_scale); f2.shift2 = prm(L.f2_shift);
                f2.out2 = slot_ptr(L.out2_slot); f2.c2 = L.f2_c2;
                ConvProjSc ps;
                if (L.psc_w >= 0) {
                    ps.x = slot_ptr(L.psc_slot); ps.w_sc = prm(L.psc_w); ps.bias_sc = static_cast<const float*>(prm(L.psc_bias));
                    ps.pro_scale = prm(L.psc_scale); ps.pro_shift = prm(L.psc_shift);
                }
                // block1 without its residual stream in HBM: metro_forward_upto stopping here (dump) stores the sum in full
                ConvRebuild rb;
                const bool has_rb = L.reb_w >= 0 || L.out_mode != 0;
                if (L.reb_w >= 0) {
                    rb.t2_prev = slot_ptr(L.reb_slot); rb.w3_prev = prm(L.reb_w); rb.bias3_prev = static_cast<const float*>(prm(L.reb_bias));
                }
                rb.out_mode = dump ? 0 : L.out_mode;
                rb.classic = dump ? 1 : 0;
                if (rb.out_mode == 2) { rb.out_sub = slot_ptr(L.sub_slot); rb.sub_off = L.sub_off; rb.h_sub = rb.w_sub = L.sub_side; }
                return launch_conv_f16_dma(cd, slot_ptr(L.in_slot), prm(L.p_w), static_cast<const float*>(prm(L.p_bias)),
                                           nullptr, nullptr, slot_ptr(L.res_slot), slot_ptr(L.out_slot), stream, nullptr, &f2,
                                           L.psc_w >= 0 ? &ps : nullptr, has_rb ? &rb : nullptr);
            }
            if (L.reb_w >= 0 || L.out_mode != 0) {
                // only the fused conv3 + next-conv1 launch above reads the rebuild data; any other path would drop the shortcut
                set_error("internal: layer %d carries block1 rebuild data (reb_w %d, out_mode %d) but no fused next-conv1 launch", li, L.reb_w, L.out_mode);
                return METRO_ERR_STATE;
            }
            if (p->fast && L.p1_w >= 0) {
                ConvPre1 p1;
                p1.w1 = prm(L.p1_w); p1.bias1 = static_cast<const float*>(prm(L.p1_bias));
                p1.pro_scale = prm(L.p1_scale); p1.pro_shift = prm(L.p1_shift);
                p1.t1_dump = dump ? slot_ptr(S_T1) : nullptr;        // conv1's output exists in LDS only; layer dumps get a copy
                return launch_conv3x3_c64(cd, slot_ptr(L.in_slot), prm(L.p_w), static_cast<const float*>(prm(L.p_bias)),
                                          slot_ptr(L.out_slot), stream, &p1);
            }
            if (p->fast)
                return launch_conv_f16(cd, slot_ptr(L.in_slot), prm(L.p_w), static_cast<const float*>(prm(L.p_bias)),
                                       prm(L.p_scale), prm(L.p_shift), slot_ptr(L.res_slot), slot_ptr(L.out_slot), stream);
            if (p->spec.precision == METRO_PREC_F32M)
                return launch_conv_f32m(cd, slot_ptr(L.in_slot), static_cast<const float*>(prm(L.p_w)), static_cast<const float*>(prm(L.p_bias)),
                                        static_cast<const float*>(prm(L.p_scale)), static_cast<const float*>(prm(L.p_shift)),
                                        slot_ptr(L.res_slot), slot_ptr(L.out_slot), stream);
            return launch_conv_f64acc(cd, slot_ptr(L.in_slot), static_cast<const double*>(prm(L.p_w)),
                                      static_cast<const double*>(prm(L.p_bias)), static_cast<const double*>(prm(L.p_scale)),
                                      static_cast<const double*>(prm(L.p_shift)), slot_ptr(L.res_slot), slot_ptr(L.out_slot), stream);
        }
        case LK_SOFTARGMAX: {
            if (poses == nullptr) { set_error("metro_forward: poses_out is NULL"); return METRO_ERR_INVALID_ARG; }
            const SoftArgmaxArgs a = make_softargmax_args(p->spec, n);
            if (L.head_fused)
                return launch_softargmax_finalize(static_cast<const float*>(slot_ptr(S_PART)), a,
                                                  head_f16_records(n, L.head_c_in, a.depth * a.n_joints_head, a.side), poses, stream, nullptr,
                                                  static_cast<int32_t*>(slot_ptr(S_STATUS)));
            // precise: 0 fp32 / fp32, 1 fp32 logits + fp64 accumulators (F32 and F32M modes), 2 fp64 / fp64
            return launch_softargmax(slot_ptr(L.in_slot), a, p->spec.precision == METRO_PREC_F32M ? 1 : p->spec.precision, slot_ptr(S_PART), poses, stream,
                                     nullptr, static_cast<int32_t*>(slot_ptr(S_STATUS)));
        }
    }
    set_error("internal: layer %d has unknown kind %d", li, L.kind);
    return METRO_ERR_STATE;
}

int run_layers(MetroPlan* p, const float* images, int n, float* poses, void* ws_, hipStream_t stream,
               int last_layer, float* ms_out) {
    METRO_CHECK_ARG(p != nullptr, "plan is NULL");
    METRO_CHECK_ARG(n > 0 && n <= p->max_batch, "batch %d outside [1, %d]", n, p->max_batch);
    METRO_CHECK_ARG(images != nullptr && ws_ != nullptr, "NULL images/workspace pointer");
    if (p->d_params == nullptr) { set_error("metro_forward: parameters not bound (metro_plan_bind_params)"); return METRO_ERR_STATE; }
    char* ws = static_cast<char*>(ws_);
    const int nl = (int)p->layers.size();
    if (last_layer < 0 || last_layer >= nl) last_layer = nl - 1;

    std::vector<hipEvent_t> ev;
    if (ms_out) {
        ev.resize(2 * (last_layer + 1));
        for (auto& e : ev) METRO_HIP_CHECK(hipEventCreate(&e));
    }
    int st = METRO_OK;
    for (int li = 0; li <= last_layer && st == METRO_OK; ++li) {
        if (ms_out) METRO_HIP_CHECK(hipEventRecord(ev[2 * li], stream));
        st = launch_layer(p, p->d_params, li, images, n, poses, ws, stream, li == last_layer && li + 1 < nl);
        if (ms_out) METRO_HIP_CHECK(hipEventRecord(ev[2 * li + 1], stream));
    }
    if (ms_out) {
        if (st == METRO_OK) {
            METRO_HIP_CHECK(hipEventSynchronize(ev.back()));
            for (int li = 0; li <= last_layer; ++li) {
                float ms = 0.f;
                METRO_HIP_CHECK(hipEventElapsedTime(&ms, ev[2 * li], ev[2 * li + 1]));
                ms_out[li] += ms;
            }
        }
        for (auto& e : ev) (void)hipEventDestroy(e);
    }
    return st;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
extern "C" {

int metro_plan_create(const MetroSpec* spec, int32_t max_batch, MetroPlan** out_plan) {
    METRO_CHECK_ARG(spec != nullptr && out_plan != nullptr, "metro_plan_create: NULL argument");
    *out_plan = nullptr;
    METRO_CHECK_ARG(spec->arch == 50 || spec->arch == 101, "unsupported arch %d (50|101)", spec->arch);
    METRO_CHECK_ARG(spec->stride == 4 || spec->stride == 8 || spec->stride == 16 || spec->stride == 32,
                    "unsupported stride %d (4|8|16|32)", spec->stride);
    METRO_CHECK_ARG(spec->proc_side > 0 && spec->proc_side % 32 == 0, "proc_side %d must be a positive multiple of 32", spec->proc_side);
    METRO_CHECK_ARG(spec->depth >= 2 && spec->depth <= 64, "depth %d out of range", spec->depth);
    METRO_CHECK_ARG(spec->n_joints_head >= 1 && spec->n_joints_head <= METRO_MAX_JOINTS, "n_joints_head %d out of range", spec->n_joints_head);
    METRO_CHECK_ARG(spec->n_joints_out >= 1 && spec->n_joints_out <= METRO_MAX_JOINTS, "n_joints_out %d out of range", spec->n_joints_out);
    for (int i = 0; i < spec->n_joints_out; ++i)
        METRO_CHECK_ARG(spec->permutation[i] >= 0 && spec->permutation[i] < spec->n_joints_head,
                        "permutation[%d] = %d outside the head's %d joints", i, spec->permutation[i], spec->n_joints_head);
    METRO_CHECK_ARG((spec->depth * spec->n_joints_head) % 4 == 0, "depth*n_joints_head must be a multiple of 4");
    METRO_CHECK_ARG(spec->precision == METRO_PREC_F16 || spec->precision == METRO_PREC_F32 || spec->precision == METRO_PREC_F64 ||
                        spec->precision == METRO_PREC_F32M, "unknown precision %d", spec->precision);
    METRO_CHECK_ARG(spec->base_width >= 8 && spec->base_width % 8 == 0, "base_width %d must be a positive multiple of 8", spec->base_width);
    METRO_CHECK_ARG(max_batch >= 1 && max_batch <= 4096, "max_batch %d out of range", max_batch);
    METRO_CHECK_ARG(spec->box_size_mm > 0.f, "box_size_mm must be positive");

    MetroPlan* p = new MetroPlan();
    p->spec = *spec;
    p->max_batch = max_batch;
    p->fast = spec->precision == METRO_PREC_F16;
    const bool f32_store = spec->precision == METRO_PREC_F32 || spec->precision == METRO_PREC_F32M;
    p->act_dtype = p->fast ? METRO_F16 : f32_store ? METRO_F32 : METRO_F64;
    p->act_bytes = p->fast ? 2 : f32_store ? 4 : 8;
    for (int s = 0; s < S_COUNT; ++s) { p->slot_bytes_per_image[s] = 0; p->slot_offset[s] = 0; }
    p->workspace_bytes = 0; p->param_bytes = 0; p->d_params = nullptr; p->flops_per_image = 0.0;
    const int st = build_plan(p);
    if (st != METRO_OK) { delete p; return st; }
    *out_plan = p;
    return METRO_OK;
}

int metro_plan_destroy(MetroPlan* plan) {
    if (plan)
        for (GraphEntry& g : plan->graphs)
            if (g.exec) (void)hipGraphExecDestroy(g.exec);
    if (plan && plan->cap_stream) (void)hipStreamDestroy(plan->cap_stream);
    delete plan;
    return METRO_OK;
}
int64_t metro_plan_workspace_bytes(const MetroPlan* plan) { return plan ? plan->workspace_bytes : -1; }
int64_t metro_plan_param_bytes(const MetroPlan* plan) { return plan ? plan->param_bytes : -1; }
int32_t metro_plan_num_params(const MetroPlan* plan) { return plan ? (int32_t)plan->params.size() : -1; }
int32_t metro_plan_num_layers(const MetroPlan* plan) { return plan ? (int32_t)plan->layers.size() : -1; }
double metro_plan_flops_per_image(const MetroPlan* plan) { return plan ? plan->flops_per_image : -1.0; }

int metro_plan_param_info(const MetroPlan* plan, int32_t index, MetroParamInfo* out) {
    METRO_CHECK_ARG(plan && out && index >= 0 && index < (int)plan->params.size(), "metro_plan_param_info: bad argument");
    *out = plan->params[index];
    return METRO_OK;
}

int metro_plan_layer_info(const MetroPlan* plan, int32_t index, MetroLayerInfo* out) {
    METRO_CHECK_ARG(plan && out && index >= 0 && index < (int)plan->layers.size(), "metro_plan_layer_info: bad argument");
    *out = plan->layers[index].info;
    return METRO_OK;
}

int metro_plan_layer_kernel(const MetroPlan* plan, int32_t index, int32_t n, char* buf, int32_t buf_len) {
    METRO_CHECK_ARG(plan && buf && buf_len > 1 && index >= 0 && index < (int)plan->layers.size(), "metro_plan_layer_kernel: bad argument");
    METRO_CHECK_ARG(n > 0 && n <= plan->max_batch, "metro_plan_layer_kernel: batch %d outside [1, %d]", n, plan->max_batch);
    // dry run of the layer's dispatch: the leaf launcher records the instantiation it would launch and returns
    KernelNotes& kn = kernel_notes();
    const KernelNotes saved = kn;
    kn.mode = 2; kn.ids[0] = 0;
    static const char fake = 0;                       // pointers are never dereferenced in a dry run (launch_status asserts it)
    float dummy_poses = 0.f;
    const int st = launch_layer(plan, &fake, index, reinterpret_cast<const float*>(&fake), n, &dummy_poses,
                                const_cast<char*>(&fake), nullptr, false);
    snprintf(buf, (size_t)buf_len, "%s", kn.ids);
    kn = saved;
    return st;
}

int metro_kernel_notes(int32_t mode) {
    METRO_CHECK_ARG(mode >= 0 && mode <= 2, "metro_kernel_notes: mode must be 0 (off), 1 (record) or 2 (dry run)");
    KernelNotes& kn = kernel_notes();
    kn.mode = mode; kn.ids[0] = 0;
    return METRO_OK;
}
const char* metro_last_kernel_id(void) { return kernel_notes().ids; }

int metro_plan_bind_params(MetroPlan* plan, const void* d_param_blob) {
    METRO_CHECK_ARG(plan && d_param_blob, "metro_plan_bind_params: NULL argument");
    METRO_CHECK_ARG(((uintptr_t)d_param_blob & 255) == 0, "parameter blob must be 256-byte aligned");
    plan->d_params = static_cast<const char*>(d_param_blob);
    // captured forwards bake the OLD blob's pointers into their kernel arguments: drop them
    for (GraphEntry& g : plan->graphs)
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
    plan->graphs.clear();
    return METRO_OK;
}

int metro_plan_set_graph_max_batch(MetroPlan* plan, int32_t max_batch_for_graphs) {
    METRO_CHECK_ARG(plan != nullptr && max_batch_for_graphs >= 0, "metro_plan_set_graph_max_batch: bad argument");
    plan->graph_max_batch = max_batch_for_graphs;
    return METRO_OK;
}

int metro_forward(MetroPlan* plan, const float* d_images_nhwc, int32_t n, float* d_poses_out,
                  void* d_workspace, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (plan == nullptr || n > plan->graph_max_batch || n < 1)
        return run_layers(plan, d_images_nhwc, n, d_poses_out, d_workspace, stream, -1, nullptr);
    // small batches are launch-latency bound (57 dependent launches): replay a captured hipGraph
    GraphEntry* hit = nullptr;
    for (GraphEntry& g : plan->graphs)
        if (g.n == n && g.images == d_images_nhwc && g.poses == d_poses_out && g.ws == d_workspace)
            hit = &g;
    if (hit == nullptr) {
        if (plan->graphs.size() >= 16) {            // bounded cache: drop the oldest capture
            if (plan->graphs.front().exec) (void)hipGraphExecDestroy(plan->graphs.front().exec);
            plan->graphs.erase(plan->graphs.begin());
        }
        plan->graphs.push_back(GraphEntry{n, d_images_nhwc, d_poses_out, d_workspace, stream, nullptr, 0});
        hit = &plan->graphs.back();
    }
    if (hit->exec == nullptr) {
        if (hit->eager_runs == 0) {                  // first sight of this key: plain launches (sets kernel attributes)
            hit->eager_runs = 1;
            return run_layers(plan, d_images_nhwc, n, d_poses_out, d_workspace, stream, -1, nullptr);
        }
        hipGraph_t graph = nullptr;
        if (plan->cap_stream == nullptr) METRO_HIP_CHECK(hipStreamCreateWithFlags(&plan->cap_stream, hipStreamNonBlocking));
        METRO_HIP_CHECK(hipStreamBeginCapture(plan->cap_stream, hipStreamCaptureModeThreadLocal));
        const int st = run_layers(plan, d_images_nhwc, n, d_poses_out, d_workspace, plan->cap_stream, -1, nullptr);
        const hipError_t e = hipStreamEndCapture(plan->cap_stream, &graph);
        if (st != METRO_OK) { if (graph) (void)hipGraphDestroy(graph); return st; }
        if (e != hipSuccess) { set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); return METRO_ERR_HIP; }
        const hipError_t ei = hipGraphInstantiate(&hit->exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (ei != hipSuccess) { hit->exec = nullptr; set_error("hipGraphInstantiate: %s", hipGetErrorString(ei)); return METRO_ERR_HIP; }
    }
    METRO_HIP_CHECK(hipGraphLaunch(hit->exec, stream));
    return METRO_OK;
}

int metro_forward_status(const MetroPlan* plan, const void* d_workspace, int32_t n, void* stream_, int32_t* n_nonfinite_out) {
    METRO_CHECK_ARG(plan && d_workspace && n_nonfinite_out, "metro_forward_status: NULL argument");
    METRO_CHECK_ARG(n > 0 && n <= plan->max_batch, "metro_forward_status: batch %d outside [1, %d]", n, plan->max_batch);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    std::vector<int32_t> host((size_t)n);
    METRO_HIP_CHECK(hipMemcpyAsync(host.data(), static_cast<const char*>(d_workspace) + plan->slot_offset[S_STATUS], (size_t)n * 4,
                                   hipMemcpyDeviceToHost, stream));
    METRO_HIP_CHECK(hipStreamSynchronize(stream));
    int32_t bad = 0;
    for (int32_t v : host) bad += v != 0;
    *n_nonfinite_out = bad;
    if (bad) {
        set_error("%d of %d crops reached the soft-argmax with non-finite statistics%s", bad, n,
                  plan->spec.precision == METRO_PREC_F16 ? " (fp16 storage overflows at 65504: run this model with precision f32m or f64)" : "");
        return METRO_ERR_NONFINITE;
    }
    return METRO_OK;
}

int64_t metro_plan_status_offset(const MetroPlan* plan) { return plan ? plan->slot_offset[S_STATUS] : -1; }

int metro_forward_upto(MetroPlan* plan, const float* d_images_nhwc, int32_t n, float* d_poses_out,
                       void* d_workspace, void* stream, int32_t last_layer) {
    return run_layers(plan, d_images_nhwc, n, d_poses_out, d_workspace, static_cast<hipStream_t>(stream), last_layer, nullptr);
}

int metro_forward_timed(MetroPlan* plan, const float* d_images_nhwc, int32_t n, float* d_poses_out,
                        void* d_workspace, void* stream, float* ms_out) {
    METRO_CHECK_ARG(ms_out != nullptr, "metro_forward_timed: ms_out is NULL");
    return run_layers(plan, d_images_nhwc, n, d_poses_out, d_workspace, static_cast<hipStream_t>(stream), -1, ms_out);
}

int metro_conv_f16(const MetroConvDesc* d, const void* d_in, const void* d_w, const float* d_bias,
                   const void* d_pro_scale, const void* d_pro_shift, const void* d_residual,
                   void* d_out, void* stream) {
    int st = validate_conv_desc(d);
    if (st) return st;
    METRO_CHECK_ARG(d->c_in % 8 == 0 && d->in_pix_stride % 4 == 0, "conv_f16: c_in must be a multiple of 8 (got %d) and in_pix_stride of 4", d->c_in);
    METRO_CHECK_ARG(d->c_out % 4 == 0, "conv_f16: c_out must be a multiple of 4 (got %d)", d->c_out);
    METRO_CHECK_ARG(d->out_dtype == METRO_F16 || d->out_dtype == METRO_F32, "conv_f16: out_dtype must be F16 or F32");
    METRO_CHECK_ARG(d->in_dtype == METRO_F16, "conv_f16: in_dtype must be F16");
    METRO_CHECK_ARG(d_in && d_w && d_bias && d_out, "conv_f16: NULL tensor pointer");
    METRO_CHECK_ARG(!d->has_prologue || (d_pro_scale && d_pro_shift), "conv_f16: prologue tensors missing");
    METRO_CHECK_ARG(!d->has_residual || d_residual, "conv_f16: residual tensor missing");
    return launch_conv_f16(*d, d_in, d_w, d_bias, d_pro_scale, d_pro_shift, d_residual, d_out,
                           static_cast<hipStream_t>(stream));
}

int metro_conv_f16_pair(const MetroConvDesc* d, const void* d_in, const void* d_w, const float* d_bias,
                        const void* d_pro_scale, const void* d_pro_shift, void* d_out, int32_t split,
                        void* d_out2, void* stream) {
    int st = validate_conv_desc(d);
    if (st) return st;
    METRO_CHECK_ARG(d->in_dtype == METRO_F16 && d->out_dtype == METRO_F16, "conv_f16_pair: fp16 tensors only");
    METRO_CHECK_ARG(d->kh == 1 && d->kw == 1 && d->stride == 1 && d->has_prologue && !d->has_residual && !d->relu,
                    "conv_f16_pair: 1x1 stride-1 convolution with prologue, without residual/ReLU on the first output");
    METRO_CHECK_ARG(d_in && d_w && d_bias && d_out && d_out2 && d_pro_scale && d_pro_shift, "conv_f16_pair: NULL tensor pointer");
    METRO_CHECK_ARG(split > 0 && split < d->c_out && split % 256 == 0 && (d->c_out - split) % 8 == 0 && d->c_in % 64 == 0,
                    "conv_f16_pair: split %d of c_out %d (split %% 256, rest %% 8, c_in %% 64 required)", split, d->c_out);
    ConvSplit sp;
    sp.split = split; sp.c_out2 = d->c_out - split; sp.relu2 = 1; sp.out2 = d_out2;
    return launch_conv_f16_dma(*d, d_in, d_w, d_bias, d_pro_scale, d_pro_shift, nullptr, d_out,
                               static_cast<hipStream_t>(stream), &sp);
}

int metro_conv_f16_next(const MetroConvDesc* d, const void* d_in, const void* d_w, const float* d_bias,
                        const void* d_residual, void* d_out, const void* d_w2, const float* d_bias2,
                        const void* d_scale2, const void* d_shift2, void* d_out2, int32_t c2, void* stream) {
    int st = validate_conv_desc(d);
    if (st) return st;
    METRO_CHECK_ARG(d_in && d_w && d_bias && d_out && d_w2 && d_bias2 && d_scale2 && d_shift2 && d_out2,
                    "conv_f16_next: NULL tensor pointer");
    METRO_CHECK_ARG(!d->has_residual || d_residual, "conv_f16_next: residual tensor missing");
    METRO_CHECK_ARG(conv_f16_fuse2_supported(*d, c2) || (conv_pw64_supported(*d, 2) && c2 == (d->c_in == 128 ? 128 : 64)),
                    "conv_f16_next: built for 1x1 stride-1 64 -> 256 with c2 = 64 (block1) and 128 -> 512 with c2 = 128 (block2), fp16");
    ConvFuse2 f2;
    f2.w2 = d_w2; f2.bias2 = d_bias2; f2.scale2 = d_scale2; f2.shift2 = d_shift2; f2.out2 = d_out2; f2.c2 = c2;
    return launch_conv_f16_dma(*d, d_in, d_w, d_bias, nullptr, nullptr, d_residual, d_out,
                               static_cast<hipStream_t>(stream), nullptr, &f2);
}

int metro_conv_f16_conv1_conv2(const MetroConvDesc* d, const void* d_x, const void* d_w1, const float* d_bias1, const void* d_pro_scale,
                               const void* d_pro_shift, const void* d_w2, const float* d_bias2, void* d_out, void* stream) {
    int st = validate_conv_desc(d);
    if (st) return st;
    METRO_CHECK_ARG(d_x && d_w1 && d_bias1 && d_pro_scale && d_pro_shift && d_w2 && d_bias2 && d_out, "conv_f16_conv1_conv2: NULL tensor pointer");
    METRO_CHECK_ARG(conv3x3_c64_supported(*d), "conv_f16_conv1_conv2: built for 3x3 stride-1 SAME 64 -> 64 on maps of <= 64 columns that tile into "
                    "128-pixel row pairs (block1 of the 256-pixel nets)");
    ConvPre1 p1;
    p1.w1 = d_w1; p1.bias1 = d_bias1; p1.pro_scale = d_pro_scale; p1.pro_shift = d_pro_shift;
    return launch_conv3x3_c64(*d, d_x, d_w2, d_bias2, d_out, static_cast<hipStream_t>(stream), &p1);
}

int metro_conv_f16_next_proj(const MetroConvDesc* d, const void* d_in, const void* d_w, const float* d_bias, const void* d_x,
                             const void* d_w_sc, const float* d_bias_sc, const void* d_pro_scale, const void* d_pro_shift, void* d_out,
                             const void* d_w2, const float* d_bias2, const void* d_scale2, const void* d_shift2, void* d_out2,
                             int32_t c2, void* stream) {
    int st = validate_conv_desc(d);
    if (st) return st;
    METRO_CHECK_ARG(d_in && d_w && d_bias && d_x && d_w_sc && d_bias_sc && d_pro_scale && d_pro_shift && d_w2 && d_bias2 &&
                        d_scale2 && d_shift2 && d_out2, "conv_f16_next_proj: NULL tensor pointer");
    METRO_CHECK_ARG(conv_pw64_supported(*d, 3) && c2 == 64, "conv_f16_next_proj: built for 1x1 stride-1 64 -> 256 without prologue / residual, "
                    "c2 = 64 (block1/unit_1), fp16");
    ConvFuse2 f2;
    f2.w2 = d_w2; f2.bias2 = d_bias2; f2.scale2 = d_scale2; f2.shift2 = d_shift2; f2.out2 = d_out2; f2.c2 = c2;
    ConvProjSc ps;
    ps.x = d_x; ps.w_sc = d_w_sc; ps.bias_sc = d_bias_sc; ps.pro_scale = d_pro_scale; ps.pro_shift = d_pro_shift;
    ConvRebuild rb;
    rb.out_mode = d_out == nullptr ? 1 : 0;          // d_out == NULL: the sum stays on chip (it only feeds the second GEMM)
    return launch_conv_f16_dma(*d, d_in, d_w, d_bias, nullptr, nullptr, nullptr, d_out, static_cast<hipStream_t>(stream), nullptr, &f2, &ps,
                               d_out == nullptr ? &rb : nullptr);
}

int metro_conv_f16_next_rebuild(const MetroConvDesc* d, const void* d_in, const void* d_w, const float* d_bias, const void* d_x,
                                const void* d_w_sc, const float* d_bias_sc, const void* d_pro_scale, const void* d_pro_shift,
                                const void* d_t2_prev, const void* d_w3_prev, const float* d_bias3_prev, void* d_out, void* d_out_sub,
                                int32_t sub_off, const void* d_w2, const float* d_bias2, const void* d_scale2, const void* d_shift2,
                                void* d_out2, int32_t c2, void* stream) {
    int st = validate_conv_desc(d);
    if (st) return st;
    METRO_CHECK_ARG(d_in && d_w && d_bias && d_x && d_w_sc && d_bias_sc && d_pro_scale && d_pro_shift && d_t2_prev && d_w3_prev && d_bias3_prev &&
                        d_w2 && d_bias2 && d_scale2 && d_shift2 && d_out2, "conv_f16_next_rebuild: NULL tensor pointer");
    METRO_CHECK_ARG((d_out != nullptr) != (d_out_sub != nullptr), "conv_f16_next_rebuild: exactly one of d_out (the whole sum) and d_out_sub (its "
                    "sub-sampled compact copy) must be given");
    METRO_CHECK_ARG(sub_off == 0 || sub_off == 1, "conv_f16_next_rebuild: sub_off %d must be 0 or 1", sub_off);
    METRO_CHECK_ARG(conv_pw64_supported(*d, 4) && c2 == 64, "conv_f16_next_rebuild: built for 1x1 stride-1 64 -> 256 without prologue / residual on maps "
                    "whose width is a power of two >= 16 and whose pixel count is a multiple of 64, c2 = 64 (block1/unit_2), fp16");
    ConvFuse2 f2;
    f2.w2 = d_w2; f2.bias2 = d_bias2; f2.scale2 = d_scale2; f2.shift2 = d_shift2; f2.out2 = d_out2; f2.c2 = c2;
    ConvProjSc ps;
    ps.x = d_x; ps.w_sc = d_w_sc; ps.bias_sc = d_bias_sc; ps.pro_scale = d_pro_scale; ps.pro_shift = d_pro_shift;
    ConvRebuild rb;
    rb.t2_prev = d_t2_prev; rb.w3_prev = d_w3_prev; rb.bias3_prev = d_bias3_prev;
    if (d_out_sub != nullptr) {
        rb.out_mode = 2; rb.out_sub = d_out_sub; rb.sub_off = sub_off;
        rb.h_sub = (d->h_out - sub_off + 1) / 2; rb.w_sub = (d->w_out - sub_off + 1) / 2;
    }
    return launch_conv_f16_dma(*d, d_in, d_w, d_bias, nullptr, nullptr, nullptr, d_out, static_cast<hipStream_t>(stream), nullptr, &f2, &ps, &rb);
}

int metro_conv_b1_form(int32_t classic) {
    METRO_CHECK_ARG(classic == 0 || classic == 1, "metro_conv_b1_form: 0 (default dispatch) or 1 (classic single-role kernel)");
    conv_b1_set_form(classic);
    return METRO_OK;
}

int metro_conv_f16_gemm4w(const MetroConvDesc* d, const void* d_in, const void* d_w, const float* d_bias,
                          const void* d_pro_scale, const void* d_pro_shift, const void* d_residual, void* d_out,
                          int32_t split, void* d_out2, void* stream) {
    int st = validate_conv_desc(d);
    if (st) return st;
    METRO_CHECK_ARG(d_in && d_w && d_bias && d_out, "conv_f16_gemm4w: NULL tensor pointer");
    METRO_CHECK_ARG(!d->has_prologue || (d_pro_scale && d_pro_shift), "conv_f16_gemm4w: prologue tensors missing");
    METRO_CHECK_ARG(!d->has_residual || d_residual, "conv_f16_gemm4w: residual tensor missing");
    METRO_CHECK_ARG(split >= 0 && split < d->c_out && (split == 0 || d_out2), "conv_f16_gemm4w: bad split %d / missing second output", split);
    ConvSplit sp;
    sp.split = split; sp.c_out2 = d->c_out - split; sp.relu2 = 1; sp.out2 = d_out2;
    return launch_conv_gemm4w(*d, d_in, d_w, d_bias, d_pro_scale, d_pro_shift, d_residual, d_out,
                              static_cast<hipStream_t>(stream), split > 0 ? &sp : nullptr);
}

int metro_stem_pool_f16(const void* d_prepped, const void* d_w, const float* d_bias, void* d_out, int32_t n,
                        int32_t side, void* stream) {
    METRO_CHECK_ARG(d_prepped && d_w && d_bias && d_out, "stem_pool_f16: NULL tensor pointer");
    METRO_CHECK_ARG(n > 0, "stem_pool_f16: n = %d", n);
    METRO_CHECK_ARG(stem_pool_f16_supported(side, 64), "stem_pool_f16: side %d must be a multiple of 32 (and METRO_STEM_POOL != 0)", side);
    return launch_stem_pool_f16(d_prepped, d_w, d_bias, d_out, n, side, static_cast<hipStream_t>(stream));
}

int metro_stem_pool_f32in(const float* d_images, const void* d_w, const float* d_bias, void* d_out, int32_t n,
                          int32_t side, void* stream) {
    METRO_CHECK_ARG(d_images && d_w && d_bias && d_out, "stem_pool_f32in: NULL tensor pointer");
    METRO_CHECK_ARG(n > 0, "stem_pool_f32in: n = %d", n);
    METRO_CHECK_ARG(stem_pool_f32in_supported(side, 64), "stem_pool_f32in: side %d must be a multiple of 32 (and METRO_STEM_POOL / METRO_STEM_RAW != 0)", side);
    return launch_stem_pool_f32in(d_images, d_w, d_bias, d_out, n, side, static_cast<hipStream_t>(stream));
}

int metro_conv_f64acc(const MetroConvDesc* d, const void* d_in, const double* d_w, const double* d_bias,
                      const double* d_pro_scale, const double* d_pro_shift, const void* d_residual,
                      void* d_out, void* stream) {
    int st = validate_conv_desc(d);
    if (st) return st;
    METRO_CHECK_ARG((d->in_dtype == METRO_F32 || d->in_dtype == METRO_F64) && (d->out_dtype == METRO_F32 || d->out_dtype == METRO_F64) &&
                        !(d->in_dtype == METRO_F64 && d->out_dtype == METRO_F32),
                    "conv_f64acc: in/out dtypes must be F32/F32, F32/F64 or F64/F64");
    METRO_CHECK_ARG(d_in && d_w && d_bias && d_out, "conv_f64acc: NULL tensor pointer");
    METRO_CHECK_ARG(!d->has_prologue || (d_pro_scale && d_pro_shift), "conv_f64acc: prologue tensors missing");
    METRO_CHECK_ARG(!d->has_residual || d_residual, "conv_f64acc: residual tensor missing");
    return launch_conv_f64acc(*d, d_in, d_w, d_bias, d_pro_scale, d_pro_shift, d_residual, d_out,
                              static_cast<hipStream_t>(stream));
}

int metro_conv_f32m(const MetroConvDesc* d, const void* d_in, const float* d_w, const float* d_bias, const float* d_pro_scale,
                    const float* d_pro_shift, const void* d_residual, void* d_out, void* stream) {
    int st = validate_conv_desc(d);
    if (st) return st;
    METRO_CHECK_ARG(d->in_dtype == METRO_F32 && d->out_dtype == METRO_F32, "conv_f32m: in/out dtypes must be F32");
    METRO_CHECK_ARG(d_in && d_w && d_bias && d_out, "conv_f32m: NULL tensor pointer");
    METRO_CHECK_ARG(!d->has_prologue || (d_pro_scale && d_pro_shift), "conv_f32m: prologue tensors missing");
    METRO_CHECK_ARG(!d->has_residual || d_residual, "conv_f32m: residual tensor missing");
    return launch_conv_f32m(*d, d_in, d_w, d_bias, d_pro_scale, d_pro_shift, d_residual, d_out, static_cast<hipStream_t>(stream));
}

int metro_prep_input_f16(const float* d_images, int32_t n, int32_t side, void* d_out, void* stream) {
    METRO_CHECK_ARG(d_images && d_out && n > 0 && side > 0, "prep_input_f16: bad argument");
    return launch_prep_input_f16(d_images, n, side, d_out, static_cast<hipStream_t>(stream));
}

int metro_warp_crop_u8(const uint8_t* d_image, int32_t h, int32_t w, int32_t row_stride, const float* d_homographies,
                       int32_t n, int32_t side, float* d_out, void* stream) {
    METRO_CHECK_ARG(d_image && d_homographies && d_out, "warp_crop_u8: NULL pointer");
    METRO_CHECK_ARG(h > 0 && w > 0 && n > 0 && side > 0 && row_stride >= 3 * w, "warp_crop_u8: bad geometry (h %d w %d stride %d n %d side %d)", h, w, row_stride, n, side);
    METRO_CHECK_ARG(h <= 32767 && w <= 32767, "warp_crop_u8: frames larger than 32767 pixels a side are outside cv2.remap's short coordinates (h %d w %d)", h, w);
    return launch_warp_crop_u8(d_image, h, w, row_stride, d_homographies, d_out, n, side, static_cast<hipStream_t>(stream));
}

int metro_eval_metrics(const float* d_pred, const float* d_true, const uint8_t* d_valid, int32_t n, int32_t n_joints,
                       float threshold_mm, float* d_dist, float* d_dist_aligned, double* d_sums, void* stream) {
    METRO_CHECK_ARG(d_pred && d_true && d_valid && d_dist && d_dist_aligned && d_sums, "eval_metrics: NULL pointer");
    METRO_CHECK_ARG(n > 0 && n_joints >= 3 && n_joints <= 1024 && threshold_mm > 0.f, "eval_metrics: bad sizes (n %d, joints %d)", n, n_joints);
    return launch_eval_metrics(d_pred, d_true, d_valid, n, n_joints, threshold_mm, d_dist, d_dist_aligned, d_sums,
                               static_cast<hipStream_t>(stream));
}

int metro_maxpool3x3s2_zeropad(const void* d_in, void* d_out, int32_t n, int32_t h_in, int32_t w_in,
                               int32_t c, int32_t dtype, void* stream) {
    METRO_CHECK_ARG(d_in && d_out && n > 0 && h_in > 0 && w_in > 0 && c > 0, "maxpool: bad argument");
    return launch_maxpool(d_in, d_out, n, h_in, w_in, c, dtype, static_cast<hipStream_t>(stream));
}

int64_t metro_softargmax_scratch_bytes(int32_t n, int32_t side, int32_t n_joints_head) {
    if (n <= 0 || side <= 1 || n_joints_head <= 0) return -1;
    return softargmax_scratch_bytes(n, side, n_joints_head);
}

int metro_softargmax(const void* d_logits, int32_t n, const MetroSpec* spec, int32_t precise,
                     void* d_partials, float* d_poses_out, void* stream) {
    METRO_CHECK_ARG(d_logits && spec && d_partials && d_poses_out && n > 0, "softargmax: bad argument");
    METRO_CHECK_ARG(spec->n_joints_head >= 1 && spec->n_joints_head <= METRO_MAX_JOINTS &&
                        spec->n_joints_out >= 1 && spec->n_joints_out <= METRO_MAX_JOINTS,
                    "softargmax: joint counts out of range");
    METRO_CHECK_ARG(spec->proc_side / spec->stride >= 2, "softargmax: heat-map side must be >= 2");
    const SoftArgmaxArgs a = make_softargmax_args(*spec, n);
    return launch_softargmax(d_logits, a, precise, d_partials, d_poses_out, static_cast<hipStream_t>(stream));
}

int64_t metro_head_f16_scratch_bytes(int32_t n, int32_t side, int32_t n_joints_head) {
    if (n <= 0 || side <= 1 || n_joints_head <= 0) return -1;
    return (int64_t)n * head_f16_slabs(side) * n_joints_head * 5 * (int64_t)sizeof(float);
}

int metro_head_f16(const void* d_x, const void* d_w, const float* d_bias, const void* d_pro_scale, const void* d_pro_shift,
                   int32_t n, int32_t c_in, const MetroSpec* spec, void* d_partials, float* d_logits_out, float* d_poses_out,
                   void* stream) {
    METRO_CHECK_ARG(d_x && d_w && d_bias && d_pro_scale && d_pro_shift && spec && d_partials && d_poses_out && n > 0,
                    "head_f16: bad argument");
    METRO_CHECK_ARG(spec->n_joints_head >= 1 && spec->n_joints_head <= METRO_MAX_JOINTS && spec->n_joints_out >= 1 &&
                        spec->n_joints_out <= METRO_MAX_JOINTS, "head_f16: joint counts out of range");
    METRO_CHECK_ARG((spec->depth * spec->n_joints_head) % 4 == 0, "head_f16: depth*n_joints_head must be a multiple of 4");
    const int side = spec->proc_side / spec->stride;
    const SoftArgmaxArgs a = make_softargmax_args(*spec, n);
    int st = launch_head_f16(d_x, d_w, d_bias, d_pro_scale, d_pro_shift, n, c_in, spec->depth * spec->n_joints_head,
                             spec->n_joints_head, spec->depth, side, static_cast<float*>(d_partials), d_logits_out,
                             static_cast<hipStream_t>(stream));
    if (st) return st;
    return launch_softargmax_finalize(static_cast<const float*>(d_partials), a,
                                      head_f16_records(n, c_in, spec->depth * spec->n_joints_head, side), d_poses_out,
                                      static_cast<hipStream_t>(stream));
}

int metro_softargmax01(const void* d_logits, int32_t n, const MetroSpec* spec, int32_t precise, void* d_partials,
                       float* d_coords01_out, void* stream) {
    METRO_CHECK_ARG(d_logits && spec && d_partials && d_coords01_out && n > 0, "softargmax01: bad argument");
    METRO_CHECK_ARG(spec->n_joints_head >= 1 && spec->n_joints_head <= METRO_MAX_JOINTS, "softargmax01: joint count out of range");
    METRO_CHECK_ARG(spec->proc_side / spec->stride >= 2, "softargmax01: heat-map side must be >= 2");
    const SoftArgmaxArgs a = make_softargmax_args(*spec, n);
    return launch_softargmax(d_logits, a, precise, d_partials, nullptr, static_cast<hipStream_t>(stream), d_coords01_out);
}

static int check_head_args(const MetroSpec* spec, int32_t n, int32_t n_edges, const char* what) {
    METRO_CHECK_ARG(spec != nullptr && n > 0, "%s: bad argument", what);
    METRO_CHECK_ARG(spec->n_joints_head >= 1 && spec->n_joints_head <= 64 && spec->n_joints_out >= 1 &&
                        spec->n_joints_out <= 64, "%s: joint counts out of range (<= 64)", what);
    METRO_CHECK_ARG(n_edges >= 0 && n_edges <= 64, "%s: at most 64 stick-figure edges (got %d)", what, n_edges);
    return METRO_OK;
}

int metro_backproject_bone_lengths(const float* d_coords01, const float* d_inv_intrinsics, const double* d_bone_lengths,
                                   int32_t per_pose_lengths, const int32_t* d_edges, int32_t n_edges, int32_t n,
                                   const MetroSpec* spec, int32_t root_relative, int32_t permute, float* d_coords3d_out,
                                   float* d_z_offset_out, void* stream) {
    int st = check_head_args(spec, n, n_edges, "backproject_bone_lengths");
    if (st) return st;
    METRO_CHECK_ARG(d_coords01 && d_inv_intrinsics && d_bone_lengths && d_edges && d_coords3d_out && n_edges >= 1,
                    "backproject_bone_lengths: NULL tensor pointer or no edges");
    return launch_backproject(d_coords01, d_inv_intrinsics, d_bone_lengths, per_pose_lengths != 0, nullptr, d_edges, n,
                              spec->n_joints_head, n_edges, *spec, root_relative, permute, d_coords3d_out, d_z_offset_out,
                              static_cast<hipStream_t>(stream));
}

int metro_backproject_root_depth(const float* d_coords01, const float* d_inv_intrinsics, const float* d_root_z, int32_t n,
                                 const MetroSpec* spec, int32_t root_relative, int32_t permute, float* d_coords3d_out,
                                 void* stream) {
    int st = check_head_args(spec, n, 0, "backproject_root_depth");
    if (st) return st;
    METRO_CHECK_ARG(d_coords01 && d_inv_intrinsics && d_root_z && d_coords3d_out, "backproject_root_depth: NULL tensor pointer");
    return launch_backproject(d_coords01, d_inv_intrinsics, nullptr, 0, d_root_z, nullptr, n, spec->n_joints_head, 0, *spec,
                              root_relative, permute, d_coords3d_out, nullptr, static_cast<hipStream_t>(stream));
}

int metro_heatmap_to_25d(const float* d_coords01, int32_t n, const MetroSpec* spec, float* d_out, void* stream) {
    int st = check_head_args(spec, n, 0, "heatmap_to_25d");
    if (st) return st;
    METRO_CHECK_ARG(d_coords01 && d_out, "heatmap_to_25d: NULL tensor pointer");
    return launch_heatmap_to_25d(d_coords01, d_out, n, *spec, static_cast<hipStream_t>(stream));
}

int metro_to_orig_cam(const float* d_coords, const float* d_rot, const int32_t* d_mirror, float* d_out, int32_t n,
                      int32_t n_joints, void* stream) {
    METRO_CHECK_ARG(d_coords && d_rot && d_mirror && d_out && n > 0 && n_joints >= 1 && n_joints <= 64,
                    "to_orig_cam: bad argument (1 <= joints <= 64)");
    return launch_to_orig_cam(d_coords, d_rot, d_mirror, d_out, n, n_joints, static_cast<hipStream_t>(stream));
}

const char* metro_last_error(void) { return metro::get_error(); }
int32_t metro_abi_version(void) { return METRO_ABI_VERSION; }

}  // extern "C"
