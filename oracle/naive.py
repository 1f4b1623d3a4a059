"""Second, independent restatement of the same graph: pure NumPy, NHWC, fp64, convolution as an
explicit sum over kernel taps of (shifted window) @ W[r, s].  TEST INFRASTRUCTURE.

Exists only to cross-check oracle/forward.py (which leans on torch's conv2d): the two share no
arithmetic code, use different layouts (NHWC here, NCHW there) and different padding code
(index arithmetic on an explicitly padded array here).  They must agree to ~1e-12 on small
cases (tests/test_oracle_cross.py).  Small inputs only: it is slow.

Reference lines followed are the same as oracle/forward.py; see there.
"""
from __future__ import annotations

from typing import Dict

import numpy as np

from oracle.spec import (OracleSpec, decode_constants, export_permutation, head_joint_info,
                         schedule)


def conv_nhwc(x, w_hwio, stride, rate, pad_beg, pad_end):
    """VALID conv over an explicitly zero-padded copy.  x [N,H,W,C], w [kh,kw,C,O]."""
    n, h, wd, c = x.shape
    kh, kw, _, o = w_hwio.shape
    xp = np.zeros((n, h + pad_beg + pad_end, wd + pad_beg + pad_end, c), dtype=x.dtype)
    xp[:, pad_beg:pad_beg + h, pad_beg:pad_beg + wd, :] = x
    k_eff = kh + (kh - 1) * (rate - 1)
    ho = (xp.shape[1] - k_eff) // stride + 1
    wo = (xp.shape[2] - k_eff) // stride + 1
    out = np.zeros((n, ho, wo, o), dtype=x.dtype)
    for r in range(kh):
        for s in range(kw):
            win = xp[:, r * rate: r * rate + (ho - 1) * stride + 1: stride,
                     s * rate: s * rate + (wo - 1) * stride + 1: stride, :]
            out += win @ w_hwio[r, s]
    return out


def same_pads(size, k_eff, stride):
    out = (size + stride - 1) // stride
    total = max((out - 1) * stride + k_eff - size, 0)
    return total // 2, total - total // 2


def bn(x, p, prefix, relu):
    inv = p[prefix + '/gamma'].astype(np.float64) / np.sqrt(
        p[prefix + '/moving_variance'].astype(np.float64) + 1e-5)
    y = (x - p[prefix + '/moving_mean'].astype(np.float64)) * inv + p[prefix + '/beta'].astype(
        np.float64)
    return np.maximum(y, 0.0) if relu else y


def forward_naive(spec: OracleSpec, params: Dict[str, np.ndarray], images_nhwc):
    f8 = lambda k: params[k].astype(np.float64)
    root = f'MainPart/{spec.arch_name}'
    x = np.asarray(images_nhwc, dtype=np.float64)
    # stem: pad 3/3, 7x7 stride 2, + bias
    x = conv_nhwc(x, f8(root + '/conv1/weights'), 2, 1, 3, 3) + f8(root + '/conv1/biases')
    # pool1: zero pad 1/1, 3x3 stride 2 max
    n, h, w, c = x.shape
    xp = np.zeros((n, h + 2, w + 2, c))
    xp[:, 1:-1, 1:-1] = x
    ho = (h + 2 - 3) // 2 + 1
    pooled = np.full((n, ho, ho, c), -np.inf)
    for r in range(3):
        for s in range(3):
            pooled = np.maximum(pooled, xp[:, r: r + 2 * (ho - 1) + 1: 2, s: s + 2 * (ho - 1) + 1: 2])
    x = pooled
    for u in schedule(spec):
        pf = f'{root}/{u.name}/bottleneck_v2'
        pre = bn(x, params, pf + '/preact', True)
        off = 1 if (u.centered and u.stride == 2) else 0
        if u.c_in == u.c_out:
            sc = x[:, off::u.stride, off::u.stride, :]
        else:
            sc = pre[:, off::u.stride, off::u.stride, :] @ f8(pf + '/shortcut/weights')[0, 0] \
                + f8(pf + '/shortcut/biases')
        r1 = bn(pre @ f8(pf + '/conv1/weights')[0, 0], params, pf + '/conv1/BatchNorm', True)
        k_eff = 3 + 2 * (u.rate - 1)
        if u.stride == 1 or u.centered:
            pb, pe = same_pads(r1.shape[1], k_eff, u.stride)
        else:
            pb = (k_eff - 1) // 2
            pe = (k_eff - 1) - pb
        r2 = conv_nhwc(r1, f8(pf + '/conv2/weights'), u.stride, u.rate, pb, pe)
        r2 = bn(r2, params, pf + '/conv2/BatchNorm', True)
        r3 = r2 @ f8(pf + '/conv3/weights')[0, 0] + f8(pf + '/conv3/biases')
        x = sc + r3
    x = bn(x, params, root + '/postnorm', True)
    logits = x @ f8(root + '/logits/weights')[0, 0] + f8(root + '/logits/biases')  # [N,S,S,D*J]
    return logits_to_pose_naive(spec, logits)


def logits_to_pose_naive(spec: OracleSpec, logits_nhwc):
    """Explicit-loop soft-argmax: channel c = d*J + j (volumetric.py:231)."""
    logits = np.asarray(logits_nhwc, dtype=np.float64)
    n, side, _, ch = logits.shape
    jn = head_joint_info(spec.dataset).n_joints
    d = spec.depth
    assert ch == d * jn
    lrc, half = decode_constants(spec)
    step_s = np.float64(np.float32(1.0) / np.float32(side - 1))
    step_d = np.float64(np.float32(1.0) / np.float32(d - 1))
    pose = np.zeros((n, jn, 3))
    for i in range(n):
        for j in range(jn):
            vol = np.stack([logits[i, :, :, dd * jn + j] for dd in range(d)], axis=-1)  # [H,W,D]
            e = np.exp(vol - vol.max())
            p = e / e.sum()
            x01 = sum(np.float64(np.float32(wi * step_s)) * p[:, wi, :].sum() for wi in range(side))
            y01 = sum(np.float64(np.float32(hi * step_s)) * p[hi, :, :].sum() for hi in range(side))
            z01 = sum(np.float64(np.float32(di * step_d)) * p[:, :, di].sum() for di in range(d))
            pose[i, j] = ((x01 * lrc + half) * spec.box_size_mm / spec.proc_side,
                          (y01 * lrc + half) * spec.box_size_mm / spec.proc_side,
                          z01 * spec.box_size_mm)
    pose = pose - pose[:, jn - 1: jn, :]
    return pose[:, export_permutation(spec.dataset), :]
