// C ABI of libmetro_experimental.so (metro_experimental.h): the argument checks of metro_conv_f16_gemm4w, then the kernel.
#include "../metro_common.h"
#include "metro_experimental.h"

using namespace metro;

extern "C" {

int metro_conv_f16_gemm8p(const MetroConvDesc* d, const void* d_in, const void* d_w, const float* d_bias,
                          const void* d_pro_scale, const void* d_pro_shift, const void* d_residual, void* d_out,
                          int32_t split, void* d_out2, void* stream) {
    int st = validate_conv_desc(d);
    if (st) return st;
    METRO_CHECK_ARG(d_in && d_w && d_bias && d_out, "conv_f16_gemm8p: NULL tensor pointer");
    METRO_CHECK_ARG(!d->has_prologue || (d_pro_scale && d_pro_shift), "conv_f16_gemm8p: prologue tensors missing");
    METRO_CHECK_ARG(!d->has_residual || d_residual, "conv_f16_gemm8p: residual tensor missing");
    METRO_CHECK_ARG(split >= 0 && split < d->c_out && (split == 0 || d_out2), "conv_f16_gemm8p: bad split %d / missing second output", split);
    ConvSplit sp;
    sp.split = split; sp.c_out2 = d->c_out - split; sp.relu2 = 1; sp.out2 = d_out2;
    return launch_conv_gemm8p(*d, d_in, d_w, d_bias, d_pro_scale, d_pro_shift, d_residual, d_out,
                              static_cast<hipStream_t>(stream), split > 0 ? &sp : nullptr);
}

int metro_conv_f16_gemm4d_geo(const MetroConvDesc* d, const void* d_in, const void* d_w, const float* d_bias,
                              const void* d_pro_scale, const void* d_pro_shift, const void* d_residual, void* d_out,
                              int32_t split, void* d_out2, int32_t geometry, void* stream) {
    int st = validate_conv_desc(d);
    if (st) return st;
    METRO_CHECK_ARG(d_in && d_w && d_bias && d_out, "conv_f16_gemm4d: NULL tensor pointer");
    METRO_CHECK_ARG(!d->has_prologue || (d_pro_scale && d_pro_shift), "conv_f16_gemm4d: prologue tensors missing");
    METRO_CHECK_ARG(!d->has_residual || d_residual, "conv_f16_gemm4d: residual tensor missing");
    METRO_CHECK_ARG(split >= 0 && split < d->c_out && (split == 0 || d_out2), "conv_f16_gemm4d: bad split %d / missing second output", split);
    METRO_CHECK_ARG(geometry >= 0 && geometry <= 3, "conv_f16_gemm4d: tile geometry %d (0 = 256 x 256, 1 = 128 x 128, 2 = 128 couts x 256 pixels, 3 = 256 x 256 with the pre-activation in place)", geometry);
    ConvSplit sp;
    sp.split = split; sp.c_out2 = d->c_out - split; sp.relu2 = 1; sp.out2 = d_out2;
    return launch_conv_gemm4d(*d, d_in, d_w, d_bias, d_pro_scale, d_pro_shift, d_residual, d_out,
                              static_cast<hipStream_t>(stream), split > 0 ? &sp : nullptr, geometry);
}

int metro_conv_f16_gemm4d(const MetroConvDesc* d, const void* d_in, const void* d_w, const float* d_bias,
                          const void* d_pro_scale, const void* d_pro_shift, const void* d_residual, void* d_out,
                          int32_t split, void* d_out2, void* stream) {
    return metro_conv_f16_gemm4d_geo(d, d_in, d_w, d_bias, d_pro_scale, d_pro_shift, d_residual, d_out, split, d_out2, 0, stream);
}

}  // extern "C"
