"""fp16-faithful restatement of the exported inference graph.  TEST INFRASTRUCTURE.

`oracle/forward.py` is the exact-math (fp64) target.  This file restates the SAME graph with a rounding to
fp16 at every place where the reference's default compute dtype puts an fp16 tensor (options.py:73
`--dtype=float16`; architectures.py:29 casts the input, tfu.py:426-440 casts every variable at use,
resnet_v2.py:119-138 are fp16 ops, architectures.py:34 casts the net output back to fp32), so that the
fp16 throughput mode of the HIP path can be held to a tight, layer-by-layer tolerance instead of being
compared with exact math only.

Arithmetic model (one rounding per stored tensor; everything between two roundings is exact, i.e. fp64):
  input            x16 = fp16(image)                                            architectures.py:29
  conv + BN + ReLU w' = fp16(w * gamma/sqrt(var+eps))  (folded in fp64, cast once), b' = fp32(beta - mean*scale);
                   y = fp16(relu(sum(w' * x) + b'))                             resnet_v2.py:127-132
  conv + bias      y = fp16(sum(fp16(w) * x) + fp32(b))                         resnet_v2.py:122-125,134-136,219-220
  pre-activation   p = fp16(max(fma(x, fp16(scale), fp16(shift)), 0))  one rounding (a fused multiply-add)
                                                                                resnet_v2.py:119,229
  residual add     out = fp16(shortcut + residual)                              resnet_v2.py:138
  logits           fp32(sum(fp16(w) * p) + fp32(b))   -- NOT rounded to fp16: the HIP path hands its fp32
                   accumulators to the soft-argmax (the reference rounds them to fp16 first,
                   architectures.py:34; keeping fp32 is strictly closer to the fp32 graph)
  soft-argmax      exact math on those fp32 logits                              tfu.py:466-499 (fp32 in the reference)

Two statements about what this is NOT:
  * it is not bit-for-bit TensorFlow: the frozen fp16 graph keeps FusedBatchNorm as its own node (the
    `fold_batch_norms` transform of main.py:150-157 only folds Mul-after-Conv2D patterns), so TF rounds the conv
    output to fp16 BEFORE the normalisation and again after it; the HIP path folds the BN scale into the weights
    and rounds once.  Both are fp16 realisations of the same fp32 graph and differ from it by the same order.
  * accumulation inside a convolution is exact here; the MFMA accumulates in fp32.  The difference (~1e-7
    relative) shows up as rare one-ulp flips of the fp16 result, which is what the layerwise tolerance allows.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from oracle.forward import BN_EPS, coords01_to_output, soft_argmax01, tf_same_pads
from oracle.spec import OracleSpec, head_joint_info, schedule


def q16(t: torch.Tensor) -> torch.Tensor:
    """Round-to-nearest-even to fp16, returned as fp64 (NumPy converts double -> half in one rounding)."""
    return torch.from_numpy(t.numpy().astype(np.float16).astype(np.float64))


def q32(t: torch.Tensor) -> torch.Tensor:
    return torch.from_numpy(t.numpy().astype(np.float32).astype(np.float64))


def _np64(a) -> np.ndarray:
    return np.asarray(a, dtype=np.float64)


def _bn_scale_shift(p: Dict[str, np.ndarray], prefix: str):
    """Inference-mode batch norm as y = x*scale + shift, in fp64 (architectures.py:9-11)."""
    scale = _np64(p[prefix + '/gamma']) / np.sqrt(_np64(p[prefix + '/moving_variance']) + BN_EPS)
    return scale, _np64(p[prefix + '/beta']) - _np64(p[prefix + '/moving_mean']) * scale


def _w16(w_hwio: np.ndarray, out_scale: Optional[np.ndarray] = None) -> torch.Tensor:
    """HWIO fp32 kernel -> (optionally BN-folded in fp64) -> fp16 -> OIHW fp64 tensor."""
    w = _np64(w_hwio)
    if out_scale is not None:
        w = w * out_scale                       # broadcast over the trailing O axis
    w = w.astype(np.float16).astype(np.float64)
    return torch.from_numpy(np.ascontiguousarray(w.transpose(3, 2, 0, 1)))


def _b32(b: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(_np64(b).astype(np.float32).astype(np.float64))[None, :, None, None]


def _preact16(x: torch.Tensor, p: Dict[str, np.ndarray], prefix: str) -> torch.Tensor:
    scale, shift = _bn_scale_shift(p, prefix)
    sc = torch.from_numpy(scale.astype(np.float16).astype(np.float64))[None, :, None, None]
    sh = torch.from_numpy(shift.astype(np.float16).astype(np.float64))[None, :, None, None]
    return torch.relu(q16(x * sc + sh))          # fp16 product is exact in fp64; one rounding of the sum


def _conv_same(x, w, stride: int, rate: int, centered: bool):
    """resnet_utils.conv2d_same (resnet_utils.py:82-135) on fp64 tensors holding fp16 values."""
    k = w.shape[-1]
    k_eff = k + (k - 1) * (rate - 1)
    if stride == 1 or centered:
        pb_h, pe_h = tf_same_pads(x.shape[2], k_eff, stride)
        pb_w, pe_w = tf_same_pads(x.shape[3], k_eff, stride)
    else:
        pb_h = pb_w = (k_eff - 1) // 2
        pe_h = pe_w = (k_eff - 1) - pb_h
    return F.conv2d(F.pad(x, (pb_w, pe_w, pb_h, pe_h)), w, None, stride=stride, dilation=rate)


# ---- one function per tensor the HIP plan stores, so a test can feed each of them the HIP path's OWN inputs ----
def stem_pool(params, root: str, images_nhwc, collect: Optional[dict] = None) -> torch.Tensor:
    """fp32 NHWC crops -> pool1 (NCHW): cast (architectures.py:29), conv1 7x7/2 + bias (resnet_v2.py:219-220),
    zero-padded 3x3/2 max-pool (resnet_utils.py:177-185)."""
    x = torch.from_numpy(np.asarray(images_nhwc, dtype=np.float32).astype(np.float16).astype(np.float64))
    x = x.permute(0, 3, 1, 2).contiguous()
    x = q16(_conv_same(x, _w16(params[root + '/conv1/weights']), 2, 1, False) + _b32(params[root + '/conv1/biases']))
    if collect is not None:
        collect['conv1'] = x
    return F.max_pool2d(F.pad(x, (1, 1, 1, 1), value=0.0), 3, 2)


def _shift(unit):
    return (lambda t: t[:, :, 1:, 1:]) if (unit.centered and unit.stride == 2) else (lambda t: t)   # resnet_v2.py:113-115


def unit_shortcut(x, p, prefix: str, unit) -> torch.Tensor:
    """resnet_v2.py:120-125: sub-sampled input, or the projection of the pre-activated input."""
    s = unit.stride
    if unit.c_in == unit.c_out:
        return _shift(unit)(x)[:, :, ::s, ::s]
    pre = _preact16(x, p, prefix + '/preact')
    return q16(F.conv2d(_shift(unit)(pre), _w16(p[prefix + '/shortcut/weights']), None, stride=s)
               + _b32(p[prefix + '/shortcut/biases']))


def unit_conv1(x, p, prefix: str) -> torch.Tensor:
    """resnet_v2.py:119,127-128: pre-activation, 1x1 conv, folded BN, ReLU."""
    pre = _preact16(x, p, prefix + '/preact')
    s1, b1 = _bn_scale_shift(p, prefix + '/conv1/BatchNorm')
    return q16(torch.relu(F.conv2d(pre, _w16(p[prefix + '/conv1/weights'], s1)) + _b32(b1)))


def unit_conv2(r1, p, prefix: str, unit) -> torch.Tensor:
    """resnet_v2.py:130-132: conv2d_same 3x3 (stride, rate), folded BN, ReLU."""
    s2, b2 = _bn_scale_shift(p, prefix + '/conv2/BatchNorm')
    return q16(torch.relu(_conv_same(r1, _w16(p[prefix + '/conv2/weights'], s2), unit.stride, unit.rate, unit.centered)
                          + _b32(b2)))


def unit_conv3_add(r2, shortcut, p, prefix: str) -> torch.Tensor:
    """resnet_v2.py:134-138: 1x1 conv + bias (an fp16 tensor), then the fp16 Add with the shortcut."""
    r3 = q16(F.conv2d(r2, _w16(p[prefix + '/conv3/weights'])) + _b32(p[prefix + '/conv3/biases']))
    return q16(shortcut + r3)


def head_logits(x, params, root: str) -> torch.Tensor:
    """resnet_v2.py:229-236: postnorm BN + ReLU, 1x1 conv + bias; fp32 result (see the header)."""
    pre = _preact16(x, params, root + '/postnorm')
    return q32(F.conv2d(pre, _w16(params[root + '/logits/weights'])) + _b32(params[root + '/logits/biases']))


def _bottleneck16(x, p, prefix: str, unit, collect: Optional[dict]):
    """resnet_v2.py:84-139 with fp16 tensors."""
    sc = unit_shortcut(x, p, prefix, unit)
    r1 = unit_conv1(x, p, prefix)
    r2 = unit_conv2(r1, p, prefix, unit)
    out = unit_conv3_add(r2, sc, p, prefix)
    if collect is not None:
        collect[unit.name + '/conv1'] = r1
        collect[unit.name + '/conv2'] = r2
        collect[unit.name + '/shortcut'] = sc
        collect[unit.name] = out
    return out


def backbone_logits(spec: OracleSpec, params: Dict[str, np.ndarray], images_nhwc,
                    collect: Optional[dict] = None) -> torch.Tensor:
    """images [N,256,256,3] fp32 in [0,1] -> logits NCHW [N, D*J, S, S] (fp32 values in an fp64 tensor)."""
    root = f'MainPart/{spec.arch_name}'
    shp = tuple(np.shape(images_nhwc))
    if len(shp) != 4 or shp[1:] != (spec.proc_side, spec.proc_side, 3):
        raise ValueError(f'expected [N,{spec.proc_side},{spec.proc_side},3], got {shp}')
    x = stem_pool(params, root, images_nhwc, collect)
    if collect is not None:
        collect['pool1'] = x
    for unit in schedule(spec):
        x = _bottleneck16(x, params, f'{root}/{unit.name}/bottleneck_v2', unit, collect)
    logits = head_logits(x, params, root)
    if collect is not None:
        collect['logits'] = logits
    return logits


def forward(spec: OracleSpec, params: Dict[str, np.ndarray], images_nhwc,
            collect: Optional[dict] = None) -> torch.Tensor:
    """Whole exported graph in the fp16 arithmetic model: images -> `output` [N, Jout, 3] (mm), fp64 tensor."""
    j = head_joint_info(spec.dataset).n_joints
    logits = backbone_logits(spec, params, images_nhwc, collect)
    _, c01 = soft_argmax01(logits, j, spec.depth)
    if collect is not None:
        collect['coords01'] = c01
    return coords01_to_output(spec, c01)
