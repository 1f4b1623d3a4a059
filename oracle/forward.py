"""fp64 (or fp32) CPU restatement of the exported inference graph.  TEST INFRASTRUCTURE.

Follows SURVEY.md appendix A.2 line by line; every step cites the reference.  Convolution
arithmetic uses torch.nn.functional.conv2d on CPU with *explicit* TF padding (the TF runtime
that the reference calls is absent; see oracle/__init__.py, "PARITY UNPINNED").
`oracle/naive.py` is an independent NumPy restatement used to cross-check this file.

Weights come in as a dict of NumPy arrays keyed by TF-slim variable names (HWIO conv kernels),
exactly what a frozen graph of the reference holds:
    MainPart/resnet_v2_50/conv1/{weights,biases}
    MainPart/resnet_v2_50/block1/unit_1/bottleneck_v2/preact/{gamma,beta,moving_mean,moving_variance}
    .../bottleneck_v2/{conv1,conv2}/weights, .../{conv1,conv2}/BatchNorm/{gamma,...}
    .../bottleneck_v2/{conv3,shortcut}/{weights,biases}
    MainPart/resnet_v2_50/postnorm/{gamma,...},  MainPart/resnet_v2_50/logits/{weights,biases}
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from oracle.spec import (OracleSpec, decode_constants, export_permutation, head_joint_info,
                         schedule)

BN_EPS = 1e-5  # reference src/model/architectures.py:10


def _t(a: np.ndarray, dtype) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype)


def _conv_w(w_hwio: np.ndarray, dtype) -> torch.Tensor:
    return _t(w_hwio, dtype).permute(3, 2, 0, 1).contiguous()  # HWIO -> OIHW


def tf_same_pads(in_size: int, k_eff: int, stride: int):
    """TF 'SAME': out = ceil(in/s); pad_total = max((out-1)*s + k_eff - in, 0); beg = total//2."""
    out = -(-in_size // stride)
    total = max((out - 1) * stride + k_eff - in_size, 0)
    return total // 2, total - total // 2


def batch_norm(x, p: Dict[str, np.ndarray], prefix: str, relu: bool):
    """Inference-mode slim.batch_norm (architectures.py:9-11), NCHW."""
    dt = x.dtype
    g = _t(p[prefix + '/gamma'], dt)
    b = _t(p[prefix + '/beta'], dt)
    m = _t(p[prefix + '/moving_mean'], dt)
    v = _t(p[prefix + '/moving_variance'], dt)
    scale = g / torch.sqrt(v + BN_EPS)
    y = (x - m[None, :, None, None]) * scale[None, :, None, None] + b[None, :, None, None]
    return torch.relu(y) if relu else y


def conv2d_same(x, w, stride: int, rate: int, centered: bool):
    """resnet_utils.py:82-135.  x NCHW, w OIHW."""
    k = w.shape[-1]
    k_eff = k + (k - 1) * (rate - 1)
    if stride == 1 or centered:
        # layers.conv2d(..., padding='SAME', stride, rate)   (resnet_utils.py:120-123)
        pb_h, pe_h = tf_same_pads(x.shape[2], k_eff, stride)
        pb_w, pe_w = tf_same_pads(x.shape[3], k_eff, stride)
    else:
        # explicit symmetric-ish pad then VALID               (resnet_utils.py:125-135)
        pad_total = k_eff - 1
        pb_h = pb_w = pad_total // 2
        pe_h = pe_w = pad_total - pad_total // 2
    x = F.pad(x, (pb_w, pe_w, pb_h, pe_h))
    return F.conv2d(x, w, None, stride=stride, padding=0, dilation=rate)


def max_pool2d_same_zeropad(x):
    """resnet_utils.py:177-185 with centered_stride=False (resnet_v2.py:222-224 never centres
    pool1): ZERO-pad (1,1), then VALID 3x3 stride-2 max-pool.  Zeros take part in the max."""
    x = F.pad(x, (1, 1, 1, 1), value=0.0)
    return F.max_pool2d(x, kernel_size=3, stride=2, padding=0)


def bottleneck(x, p, prefix: str, unit, collect: Optional[dict]):
    """resnet_v2.py:84-139."""
    dt = x.dtype
    s, r, centered = unit.stride, unit.rate, unit.centered
    shift = (lambda t: t[:, :, 1:, 1:]) if (centered and s == 2) else (lambda t: t)  # :113-115
    pre = batch_norm(x, p, prefix + '/preact', relu=True)                               # :119
    if unit.c_in == unit.c_out:
        sc = shift(x)[:, :, ::s, ::s]                       # :120-121, resnet_utils.py:76-79
    else:
        w = _conv_w(p[prefix + '/shortcut/weights'], dt)
        sc = F.conv2d(shift(pre), w, _t(p[prefix + '/shortcut/biases'], dt), stride=s)  # :122-125
    r1 = F.conv2d(pre, _conv_w(p[prefix + '/conv1/weights'], dt))                       # :127-128
    r1 = batch_norm(r1, p, prefix + '/conv1/BatchNorm', relu=True)
    r2 = conv2d_same(r1, _conv_w(p[prefix + '/conv2/weights'], dt), s, r, centered)     # :130-132
    r2 = batch_norm(r2, p, prefix + '/conv2/BatchNorm', relu=True)
    r3 = F.conv2d(r2, _conv_w(p[prefix + '/conv3/weights'], dt),
                  _t(p[prefix + '/conv3/biases'], dt))                                  # :134-136
    out = sc + r3                                                                       # :138
    if collect is not None:
        name = unit.name
        collect[name + '/conv1'] = r1
        collect[name + '/conv2'] = r2
        collect[name + '/shortcut'] = sc
        collect[name] = out
    return out


def backbone_logits(spec: OracleSpec, params: Dict[str, np.ndarray], images_nhwc,
                    dtype=torch.float64, collect: Optional[dict] = None):
    """images [N,256,256,3] in [0,1] -> logits NCHW [N, D*J, S, S] (resnet_v2.py:203-241)."""
    root = f'MainPart/{spec.arch_name}'
    x = torch.as_tensor(np.asarray(images_nhwc)).to(dtype).permute(0, 3, 1, 2).contiguous()
    if x.shape[1] != 3 or x.shape[2] != spec.proc_side or x.shape[3] != spec.proc_side:
        raise ValueError(f'expected [N,{spec.proc_side},{spec.proc_side},3], got NHWC '
                         f'{tuple(images_nhwc.shape)}')
    # conv1: 7x7/2, explicit pad 3/3, bias, no BN, no ReLU (resnet_v2.py:219-220)
    x = conv2d_same(x, _conv_w(params[root + '/conv1/weights'], dtype), 2, 1, False)
    x = x + _t(params[root + '/conv1/biases'], dtype)[None, :, None, None]
    if collect is not None:
        collect['conv1'] = x
    x = max_pool2d_same_zeropad(x)                                           # :222-224
    if collect is not None:
        collect['pool1'] = x
    for unit in schedule(spec):
        x = bottleneck(x, params, f'{root}/{unit.name}/bottleneck_v2', unit, collect)
    x = batch_norm(x, params, root + '/postnorm', relu=True)                 # :229
    if collect is not None:
        collect['postnorm'] = x
    logits = F.conv2d(x, _conv_w(params[root + '/logits/weights'], dtype),
                      _t(params[root + '/logits/biases'], dtype))            # :233-236
    if collect is not None:
        collect['logits'] = logits
    return logits


def soft_argmax01(logits_nchw, n_joints: int, depth: int):
    """volumetric.py:227-235 + tfu.py:466-499.  Returns (softmaxed [N,J,S,S,D], coords01 [N,J,3])."""
    n, c, side, _ = logits_nchw.shape
    assert c == depth * n_joints
    reshaped = logits_nchw.reshape(n, depth, n_joints, side, side)           # :231 (c = d*J + j)
    vol = reshaped.permute(0, 2, 3, 4, 1)                                    # :232 -> [N,J,H,W,D]
    m = vol.amax(dim=(2, 3, 4), keepdim=True)                                # tfu.py:468
    e = torch.exp(vol - m)                                                   # tfu.py:469
    p = e / e.sum(dim=(2, 3, 4), keepdim=True)                               # tfu.py:470-471
    dt = p.dtype

    def lin(k):  # tf.linspace(0.0, 1.0, k) is evaluated in fp32, then cast (tfu.py:481-482)
        step = np.float32(1.0) / np.float32(k - 1)
        return torch.from_numpy((np.arange(k, dtype=np.float32) * step).astype(np.float32)).to(dt)

    # decode_heatmap(softmaxed, [3, 2, 4]): x <- axis 3 (W), y <- axis 2 (H), z <- axis 4 (D)
    x01 = (p.sum(dim=(2, 4)) * lin(side)).sum(dim=-1)                        # tfu.py:491-496
    y01 = (p.sum(dim=(3, 4)) * lin(side)).sum(dim=-1)
    z01 = (p.sum(dim=(2, 3)) * lin(depth)).sum(dim=-1)
    return p, torch.stack([x01, y01, z01], dim=-1)                           # volumetric.py:234


def coords01_to_output(spec: OracleSpec, coords01):
    """heatmap_to_metric (volumetric.py:303-306) -> root_relative (tfu3d.py:23-25) ->
    tf.gather(permutation) (main.py:119-127)."""
    lrc, half = decode_constants(spec)
    xy_px = coords01[..., :2] * lrc + half                                   # :291-294
    xy_mm = xy_px * spec.box_size_mm / spec.proc_side                        # :304-305
    z_mm = coords01[..., 2:] * spec.box_size_mm                              # :306
    pose = torch.cat([xy_mm, z_mm], dim=-1)
    pose = pose - pose[:, -1:, :]                                            # tfu3d.py:24-25
    perm = export_permutation(spec.dataset)
    return pose[:, perm, :]                                                  # main.py:127


def forward(spec: OracleSpec, params: Dict[str, np.ndarray], images_nhwc,
            dtype=torch.float64, collect: Optional[dict] = None):
    """Whole exported graph: images -> `output` [N, Jout, 3] (mm, root-relative)."""
    j = head_joint_info(spec.dataset).n_joints
    logits = backbone_logits(spec, params, images_nhwc, dtype, collect)
    # architectures.py:34 casts the net output to fp32; softmax/decode run in fp32 in the
    # reference.  The oracle keeps `dtype` throughout (fp64 = exact-math target).
    _, c01 = soft_argmax01(logits, j, spec.depth)
    if collect is not None:
        collect['coords01'] = c01
    return coords01_to_output(spec, c01)


def logits_to_output(spec: OracleSpec, logits_nhwc, dtype=torch.float64):
    """Stand-alone soft-argmax + decode from NHWC logits [N,S,S,D*J] (the K6 kernel's job)."""
    j = head_joint_info(spec.dataset).n_joints
    lg = torch.as_tensor(np.asarray(logits_nhwc)).to(dtype).permute(0, 3, 1, 2)
    _, c01 = soft_argmax01(lg, j, spec.depth)
    return coords01_to_output(spec, c01)
