"""Shared test plumbing: calling the C ABI with torch device buffers, oracle-side references."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from metro_pose3d_amd import _lib
from metro_pose3d_amd._lib import check


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def conv_desc(n, h_in, c_in, h_out, c_out, k, stride=1, dil=1, pad=0, prologue=False, relu=False,
              residual=False, res_h=0, res_stride=1, res_offset=0, out_dtype=_lib.METRO_F16,
              w_in=None, w_out=None, in_pix_stride=None, kh=None, kw=None, in_dtype=None):
    d = _lib.MetroConvDesc()
    d.n = n
    d.h_in = h_in
    d.w_in = w_in if w_in is not None else h_in
    d.c_in = c_in
    d.in_pix_stride = in_pix_stride if in_pix_stride is not None else c_in
    d.h_out = h_out
    d.w_out = w_out if w_out is not None else h_out
    d.c_out = c_out
    d.kh = kh if kh is not None else k
    d.kw = kw if kw is not None else k
    d.stride = stride
    d.dilation = dil
    d.pad_top = d.pad_left = pad
    d.has_prologue = int(prologue)
    d.relu = int(relu)
    d.has_residual = int(residual)
    d.res_h = d.res_w = res_h
    d.res_stride = res_stride
    d.res_offset = res_offset
    d.out_dtype = out_dtype
    d.in_dtype = in_dtype if in_dtype is not None else (
        _lib.METRO_F16 if out_dtype == _lib.METRO_F16 else out_dtype)
    return d


def ref_conv_nhwc(x, w_ok, bias, stride, dil, pad, h_out, pro=None, relu=False, res=None,
                  res_stride=1, res_offset=0):
    """fp64 reference of one MetroConvDesc: x [N,H,W,C], w_ok [O, kh, kw, C] (the packed layout).

    `pad` is (pad_top == pad_left); the bottom/right pad is whatever makes the output h_out wide,
    i.e. taps that fall outside read zeros (TF zero padding, reference resnet_utils.py:125-135)."""
    x = torch.as_tensor(x, dtype=torch.float64)
    w = torch.as_tensor(w_ok, dtype=torch.float64)
    if pro is not None:
        sc, sh = (torch.as_tensor(t, dtype=torch.float64) for t in pro)
        x = torch.relu(x * sc + sh)
    n, h, wd, c = x.shape
    o, kh, kw, _ = w.shape
    need_h = (h_out - 1) * stride + (kh - 1) * dil + 1
    need_w = (h_out - 1) * stride + (kw - 1) * dil + 1
    # tap (r,s) of output (ho,wo) reads input (ho*stride - pad + r*dil, ...), zero outside
    big = max(pad, 0)
    xp = torch.zeros((n, big + max(h, need_h - pad) + 1, big + max(wd, need_w - pad) + 1, c),
                     dtype=torch.float64)
    xp[:, big:big + h, big:big + wd] = x
    xp = xp[:, big - pad:big - pad + need_h, big - pad:big - pad + need_w]
    y = torch.nn.functional.conv2d(xp.permute(0, 3, 1, 2), w.permute(0, 3, 1, 2), None, stride=stride,
                                   dilation=dil).permute(0, 2, 3, 1)
    y = y + torch.as_tensor(bias, dtype=torch.float64)
    if relu:
        y = torch.relu(y)
    if res is not None:
        r = torch.as_tensor(res, dtype=torch.float64)
        y = y + r[:, res_offset::res_stride, res_offset::res_stride][:, :h_out, :h_out]
    return y


def run_conv_f16(lib, dev, d, x, w, bias, pro=None, res=None):
    """x, w, pro, res: numpy (cast to fp16); bias fp32.  Returns numpy of d.out_dtype."""
    tx = torch.from_numpy(np.ascontiguousarray(x.astype(np.float16))).to(dev)
    tw = torch.from_numpy(np.ascontiguousarray(w.astype(np.float16))).to(dev)
    tb = torch.from_numpy(np.ascontiguousarray(bias.astype(np.float32))).to(dev)
    ts = tsh = tr = None
    if pro is not None:
        ts = torch.from_numpy(pro[0].astype(np.float16)).to(dev)
        tsh = torch.from_numpy(pro[1].astype(np.float16)).to(dev)
    if res is not None:
        tr = torch.from_numpy(np.ascontiguousarray(res.astype(np.float16))).to(dev)
    odt = torch.float16 if d.out_dtype == _lib.METRO_F16 else torch.float32
    out = torch.full((d.n, d.h_out, d.w_out, d.c_out), float('nan'), dtype=odt, device=dev)
    check(lib.metro_conv_f16(C.byref(d), ptr(tx), ptr(tw), ptr(tb), ptr(ts), ptr(tsh), ptr(tr), ptr(out),
                             C.c_void_p(0)), 'metro_conv_f16')
    torch.cuda.synchronize()
    return out.cpu().numpy()


def run_conv_f64acc(lib, dev, d, x, w, bias, pro=None, res=None):
    np_in = np.float32 if d.in_dtype == _lib.METRO_F32 else np.float64
    np_out = np.float32 if d.out_dtype == _lib.METRO_F32 else np.float64
    tx = torch.from_numpy(np.ascontiguousarray(x.astype(np_in))).to(dev)
    tw = torch.from_numpy(np.ascontiguousarray(w.astype(np.float64))).to(dev)
    tb = torch.from_numpy(np.ascontiguousarray(bias.astype(np.float64))).to(dev)
    ts = tsh = tr = None
    if pro is not None:
        ts = torch.from_numpy(pro[0].astype(np.float64)).to(dev)
        tsh = torch.from_numpy(pro[1].astype(np.float64)).to(dev)
    if res is not None:
        tr = torch.from_numpy(np.ascontiguousarray(res.astype(np_out))).to(dev)
    out = torch.full((d.n, d.h_out, d.w_out, d.c_out), float('nan'),
                     dtype=torch.float32 if np_out is np.float32 else torch.float64, device=dev)
    check(lib.metro_conv_f64acc(C.byref(d), ptr(tx), ptr(tw), ptr(tb), ptr(ts), ptr(tsh), ptr(tr),
                                ptr(out), C.c_void_p(0)), 'metro_conv_f64acc')
    torch.cuda.synchronize()
    return out.cpu().numpy()


def run_softargmax(lib, dev, spec, logits, precise):
    """spec: metro_pose3d_amd.ModelSpec; logits numpy [n,S,S,D*J] fp32."""
    n = logits.shape[0]
    cs = spec.to_c(int(precise))
    tl = torch.from_numpy(np.ascontiguousarray(logits.astype(np.float64 if int(precise) == 2 else np.float32))).to(dev)
    sb = lib.metro_softargmax_scratch_bytes(n, spec.heatmap_side, spec.skeleton.n_head)
    scratch = torch.empty(sb, dtype=torch.uint8, device=dev)
    out = torch.full((n, spec.skeleton.n_out, 3), float('nan'), dtype=torch.float32, device=dev)
    check(lib.metro_softargmax(ptr(tl), n, C.byref(cs), int(precise), ptr(scratch), ptr(out),
                               C.c_void_p(0)), 'metro_softargmax')
    torch.cuda.synchronize()
    return out.cpu().numpy()


def oracle_spec(spec):
    """metro_pose3d_amd.ModelSpec -> oracle.spec.OracleSpec (same field names by design)."""
    from oracle.spec import OracleSpec
    return OracleSpec(arch=spec.arch, stride=spec.stride, dataset=spec.dataset, depth=spec.depth,
                      centered_stride=spec.centered_stride, proc_side=spec.proc_side,
                      box_size_mm=spec.box_size_mm, base_width=spec.base_width)


def assert_as_accurate_as_fp16_model(spec, params, images, poses, what='', ratio_mean=2.0, ratio_max=2.5):
    """The accuracy criterion of the f16 mode: `poses` (the HIP path's, for exactly these crops) may
    not be further from exact math (fp64 oracle) than the one-rounding-per-tensor fp16 model of the graph (oracle/f16emu.py)
    is itself -- mean within ratio_mean, maximum within ratio_max (two draws of the same rounding noise)."""
    from oracle import f16emu
    from oracle import forward as OF
    ospec = oracle_spec(spec)
    exact = OF.forward(ospec, params, images, torch.float64).numpy()
    emu = f16emu.forward(ospec, params, images).numpy()
    poses = np.asarray(poses, dtype=np.float64)
    assert np.isfinite(poses).all(), what
    d, de = np.abs(poses - exact), np.abs(emu - exact)
    assert d.mean() <= ratio_mean * de.mean() and d.max() <= ratio_max * de.max(), \
        f'{what}: |hip - fp64| mean {d.mean():.3f} max {d.max():.3f} mm vs the fp16 model\'s own {de.mean():.3f} / {de.max():.3f} mm'
    return float(d.max()), float(de.max())
