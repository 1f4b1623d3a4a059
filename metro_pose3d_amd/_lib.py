"""ctypes bindings of include/metro_hip.h.  Loading fails loudly: there is NO CPU fallback."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('METRO_HIP_LIB') or os.path.join(HERE, 'libmetro_hip.so')   # override: timing experiments

METRO_MAX_JOINTS = 64
ABI_VERSION = 8          # include/metro_hip.h METRO_ABI_VERSION
METRO_PREC_F16, METRO_PREC_F32, METRO_PREC_F64, METRO_PREC_F32M = 0, 1, 2, 3
METRO_F16, METRO_F32, METRO_F64 = 0, 1, 2
PARAM_CONV_W, PARAM_BIAS, PARAM_PRO_SCALE, PARAM_PRO_SHIFT = 0, 1, 2, 3
LAYER_PREP, LAYER_CONV, LAYER_POOL, LAYER_SOFTARGMAX = 0, 1, 2, 3
FUSED_CONV1_IN_FRONT, FUSED_PROJECTION_SHORTCUT = 1, 2      # MetroLayerInfo.fused_flags (METRO_FUSED_*)
FUSED_OUT_ON_CHIP, FUSED_REBUILT_SHORTCUT, FUSED_COMPACT_SHORTCUT = 4, 8, 16


class MetroSpec(C.Structure):
    _fields_ = [('arch', C.c_int32), ('stride', C.c_int32), ('n_joints_head', C.c_int32),
                ('depth', C.c_int32), ('centered_stride', C.c_int32), ('proc_side', C.c_int32),
                ('box_size_mm', C.c_float), ('base_width', C.c_int32), ('precision', C.c_int32),
                ('n_joints_out', C.c_int32), ('permutation', C.c_int32 * METRO_MAX_JOINTS)]


class MetroParamInfo(C.Structure):
    _fields_ = [('name', C.c_char * 96), ('conv_var', C.c_char * 160), ('bn_var', C.c_char * 160),
                ('kind', C.c_int32), ('dtype', C.c_int32),
                ('c_out', C.c_int32), ('kh', C.c_int32), ('kw', C.c_int32), ('c_in', C.c_int32),
                ('kw_pad', C.c_int32), ('c_in_pad', C.c_int32),
                ('offset', C.c_int64), ('bytes', C.c_int64)]


class MetroLayerInfo(C.Structure):
    _fields_ = [('name', C.c_char * 96), ('kind', C.c_int32),
                ('h_in', C.c_int32), ('w_in', C.c_int32), ('c_in', C.c_int32),
                ('h_out', C.c_int32), ('w_out', C.c_int32), ('c_out', C.c_int32),
                ('kh', C.c_int32), ('kw', C.c_int32), ('stride', C.c_int32),
                ('dilation', C.c_int32), ('pad_top', C.c_int32), ('pad_left', C.c_int32),
                ('has_prologue', C.c_int32), ('relu', C.c_int32), ('has_residual', C.c_int32),
                ('res_stride', C.c_int32), ('res_offset', C.c_int32), ('out_dtype', C.c_int32),
                ('out_offset', C.c_int64), ('out_bytes_per_image', C.c_int64),
                ('flops_per_image', C.c_double),
                ('out2_offset', C.c_int64), ('out2_channels', C.c_int32), ('fused_flags', C.c_int32),
                ('algo_act_bytes_per_image', C.c_int64), ('algo_param_bytes', C.c_int64),
                ('out_sub_offset', C.c_int64), ('out_sub_side', C.c_int32), ('out_sub_off', C.c_int32)]


class MetroConvDesc(C.Structure):
    _fields_ = [('n', C.c_int32), ('h_in', C.c_int32), ('w_in', C.c_int32), ('c_in', C.c_int32),
                ('in_pix_stride', C.c_int32),
                ('h_out', C.c_int32), ('w_out', C.c_int32), ('c_out', C.c_int32),
                ('kh', C.c_int32), ('kw', C.c_int32), ('stride', C.c_int32), ('dilation', C.c_int32),
                ('pad_top', C.c_int32), ('pad_left', C.c_int32),
                ('has_prologue', C.c_int32), ('relu', C.c_int32), ('has_residual', C.c_int32),
                ('res_h', C.c_int32), ('res_w', C.c_int32),
                ('res_stride', C.c_int32), ('res_offset', C.c_int32), ('out_dtype', C.c_int32),
                ('in_dtype', C.c_int32)]


# symbol -> (restype, argtypes); must list every function include/metro_hip.h declares
_P = C.c_void_p
SIGNATURES = {
    'metro_plan_create': (C.c_int, [C.POINTER(MetroSpec), C.c_int32, C.POINTER(_P)]),
    'metro_plan_destroy': (C.c_int, [_P]),
    'metro_plan_workspace_bytes': (C.c_int64, [_P]),
    'metro_plan_param_bytes': (C.c_int64, [_P]),
    'metro_plan_num_params': (C.c_int32, [_P]),
    'metro_plan_param_info': (C.c_int, [_P, C.c_int32, C.POINTER(MetroParamInfo)]),
    'metro_plan_num_layers': (C.c_int32, [_P]),
    'metro_plan_layer_info': (C.c_int, [_P, C.c_int32, C.POINTER(MetroLayerInfo)]),
    'metro_plan_flops_per_image': (C.c_double, [_P]),
    'metro_plan_bind_params': (C.c_int, [_P, _P]),
    'metro_plan_layer_kernel': (C.c_int, [_P, C.c_int32, C.c_int32, C.c_char_p, C.c_int32]),
    'metro_kernel_notes': (C.c_int, [C.c_int32]),
    'metro_last_kernel_id': (C.c_char_p, []),
    'metro_plan_set_graph_max_batch': (C.c_int, [_P, C.c_int32]),
    'metro_forward': (C.c_int, [_P, _P, C.c_int32, _P, _P, _P]),
    'metro_forward_status': (C.c_int, [_P, _P, C.c_int32, _P, C.POINTER(C.c_int32)]),
    'metro_plan_status_offset': (C.c_int64, [_P]),
    'metro_forward_upto': (C.c_int, [_P, _P, C.c_int32, _P, _P, _P, C.c_int32]),
    'metro_forward_timed': (C.c_int, [_P, _P, C.c_int32, _P, _P, _P, C.POINTER(C.c_float)]),
    'metro_conv_f16': (C.c_int, [C.POINTER(MetroConvDesc), _P, _P, _P, _P, _P, _P, _P, _P]),
    'metro_conv_f32m': (C.c_int, [C.POINTER(MetroConvDesc), _P, _P, _P, _P, _P, _P, _P, _P]),
    'metro_conv_f64acc': (C.c_int, [C.POINTER(MetroConvDesc), _P, _P, _P, _P, _P, _P, _P, _P]),
    'metro_conv_f16_pair': (C.c_int, [C.POINTER(MetroConvDesc), _P, _P, _P, _P, _P, _P, C.c_int32, _P, _P]),
    'metro_conv_b1_form': (C.c_int, [C.c_int32]),
    'metro_conv_f16_gemm4w': (C.c_int, [C.POINTER(MetroConvDesc), _P, _P, _P, _P, _P, _P, _P, C.c_int32, _P, _P]),
    'metro_conv_f16_conv1_conv2': (C.c_int, [C.POINTER(MetroConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'metro_conv_f16_next_proj': (C.c_int, [C.POINTER(MetroConvDesc)] + [_P] * 14 + [C.c_int32, _P]),
    'metro_conv_f16_next_rebuild': (C.c_int, [C.POINTER(MetroConvDesc)] + [_P] * 13 + [C.c_int32] + [_P] * 5 + [C.c_int32, _P]),
    'metro_conv_f16_next': (C.c_int, [C.POINTER(MetroConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int32, _P]),
    'metro_stem_pool_f16': (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, _P]),
    'metro_stem_pool_f32in': (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, _P]),
    'metro_prep_input_f16': (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P]),
    'metro_warp_crop_u8': (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, C.c_int32, _P, _P]),
    'metro_eval_metrics': (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_float, _P, _P, _P, _P]),
    'metro_maxpool3x3s2_zeropad': (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                             C.c_int32, _P]),
    'metro_softargmax_scratch_bytes': (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    'metro_softargmax01': (C.c_int, [_P, C.c_int32, C.POINTER(MetroSpec), C.c_int32, _P, _P, _P]),
    'metro_backproject_bone_lengths': (C.c_int, [_P, _P, _P, C.c_int32, _P, C.c_int32, C.c_int32, C.POINTER(MetroSpec),
                                                 C.c_int32, C.c_int32, _P, _P, _P]),
    'metro_backproject_root_depth': (C.c_int, [_P, _P, _P, C.c_int32, C.POINTER(MetroSpec), C.c_int32, C.c_int32, _P, _P]),
    'metro_to_orig_cam': (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, _P]),
    'metro_heatmap_to_25d': (C.c_int, [_P, C.c_int32, C.POINTER(MetroSpec), _P, _P]),
    'metro_head_f16_scratch_bytes': (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    'metro_head_f16': (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.POINTER(MetroSpec), _P, _P, _P, _P]),
    'metro_softargmax': (C.c_int, [_P, C.c_int32, C.POINTER(MetroSpec), C.c_int32, _P, _P, _P]),
    'metro_last_error': (C.c_char_p, []),
    'metro_abi_version': (C.c_int32, []),
}

# libmetro_experimental.so (csrc/experimental/metro_experimental.h): kernels metro_forward never dispatches, kept for the
# probes in tools/ and for tests/test_gpu_kernels.py::test_conv_gemm_experimental
EXPERIMENTAL_LIB_PATH = os.path.join(HERE, 'libmetro_experimental.so')
EXPERIMENTAL_SIGNATURES = {
    'metro_conv_f16_gemm8p': (C.c_int, [C.POINTER(MetroConvDesc), _P, _P, _P, _P, _P, _P, _P, C.c_int32, _P, _P]),
    'metro_conv_f16_gemm4d': (C.c_int, [C.POINTER(MetroConvDesc), _P, _P, _P, _P, _P, _P, _P, C.c_int32, _P, _P]),
    'metro_conv_f16_gemm4d_geo': (C.c_int, [C.POINTER(MetroConvDesc), _P, _P, _P, _P, _P, _P, _P, C.c_int32, _P, C.c_int32, _P]),
}

_lib = None
_xlib = None


class MetroError(RuntimeError):
    pass


class NonFiniteError(MetroError):
    """metro_forward_status: activations overflowed the arithmetic mode (fp16 storage tops out at 65 504)."""


def load() -> C.CDLL:
    """Loads libmetro_hip.so.  Raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MetroError(
            f'{LIB_PATH} is missing: build it with `python -m metro_pose3d_amd.build` '
            '(hipcc --offload-arch=gfx950). There is no CPU fallback for this path.')
    # torch first: it brings its own HIP runtime; loading this library before torch would pull a second libamdhip64
    # (the system one) into the process, whose kernels then see "no ROCm-capable device"
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)            # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.metro_abi_version() != ABI_VERSION:
        raise MetroError(f'ABI version mismatch: library {lib.metro_abi_version()}, bindings {ABI_VERSION}')
    _lib = lib
    return lib


def load_experimental() -> C.CDLL:
    """Loads libmetro_experimental.so (after the product library it links against)."""
    global _xlib
    if _xlib is not None:
        return _xlib
    load()
    if not os.path.exists(EXPERIMENTAL_LIB_PATH):
        raise MetroError(f'{EXPERIMENTAL_LIB_PATH} is missing: build it with `python -m metro_pose3d_amd.build`')
    lib = C.CDLL(EXPERIMENTAL_LIB_PATH)
    for name, (res, args) in EXPERIMENTAL_SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _xlib = lib
    return lib


def check(status: int, what: str = '') -> None:
    if status == 0:
        return
    msg = load().metro_last_error().decode(errors='replace')
    if status == -1:
        raise ValueError(f'{what}: {msg}')
    if status == -5:
        raise NonFiniteError(f'{what}: {msg}')
    raise MetroError(f'{what}: status {status}: {msg}')
