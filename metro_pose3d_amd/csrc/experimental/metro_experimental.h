/*
 * metro_experimental.h -- C ABI of libmetro_experimental.so: kernels that metro_forward NEVER dispatches.
 *
 * Two earlier forms of the deep-K 1x1 GEMM (reference resnet_v2.py:122-128: conv1 / projection shortcut on `preact`) that the
 * product's conv_gemm4w.hip replaced or tied with inside the forward (NOTES_dead_ends.md has the measurements):
 *   conv_gemm8p.hip (round 2)  eight waves of 128 x 64, LDS-DMA ring, 8-phase two-wave-group schedule;
 *   conv_gemm4d.hip (round 4)  four waves of 128 x 128, both operands by LDS-DMA in 128-byte row pieces, four fragment sets;
 *                              also on 128 x 128 and 128 x 256 block tiles (geometry 1 / 2).
 * Same contract as metro_conv_f16_gemm4w (include/metro_hip.h); same K order and one fp32 accumulator per output: all three give
 * the same bits (tests/test_gpu_kernels.py::test_conv_gemm_experimental).  The library links against libmetro_hip.so (error
 * strings, dispatch notes, descriptor validation) and is loaded by tools/ probes and that test only.
 */
#ifndef METRO_EXPERIMENTAL_H
#define METRO_EXPERIMENTAL_H

#include "../../../include/metro_hip.h"

#ifdef __cplusplus
extern "C" {
#endif
int  metro_conv_f16_gemm8p(const MetroConvDesc* d, const void* d_in, const void* d_w, const float* d_bias,
                           const void* d_pro_scale, const void* d_pro_shift, const void* d_residual, void* d_out,
                           int32_t split, void* d_out2, void* stream);
int  metro_conv_f16_gemm4d(const MetroConvDesc* d, const void* d_in, const void* d_w, const float* d_bias,
                           const void* d_pro_scale, const void* d_pro_shift, const void* d_residual, void* d_out,
                           int32_t split, void* d_out2, void* stream);
/* geometry 0 = 256 couts x 256 pixels, 1 = 128 x 128 (two blocks per CU), 2 = 128 couts x 256 pixels */
int  metro_conv_f16_gemm4d_geo(const MetroConvDesc* d, const void* d_in, const void* d_w, const float* d_bias,
                               const void* d_pro_scale, const void* d_pro_shift, const void* d_residual, void* d_out,
                               int32_t split, void* d_out2, int32_t geometry, void* stream);
#ifdef __cplusplus
}

#include <hip/hip_runtime.h>
namespace metro {
struct ConvSplit;
int launch_conv_gemm8p(const MetroConvDesc& d, const void* in, const void* w, const float* bias, const void* pro_scale,
                       const void* pro_shift, const void* residual, void* out, hipStream_t stream, const ConvSplit* split);
bool conv_gemm4d_shape_ok(const MetroConvDesc& d, const ConvSplit* split);
bool conv_gemm4d_geo_ok(const MetroConvDesc& d, const ConvSplit* split, int geo);
int launch_conv_gemm4d(const MetroConvDesc& d, const void* in, const void* w, const float* bias, const void* pro_scale,
                       const void* pro_shift, const void* residual, void* out, hipStream_t stream, const ConvSplit* split = nullptr,
                       int geo = 0);
}  // namespace metro
#endif
#endif
