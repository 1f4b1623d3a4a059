"""Host-side plumbing around libmetro_hip.so: plan creation, parameter folding/packing, device
buffers (torch) and the forward call.  No arithmetic of the hot path happens here: torch is used
for device memory and streams only.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import numpy as np
import torch

from metro_pose3d_amd import _lib
from metro_pose3d_amd._lib import check
from metro_pose3d_amd.spec import ModelSpec

BN_EPS = 1e-5   # reference src/model/architectures.py:10

_NP_DTYPE = {_lib.METRO_F16: np.float16, _lib.METRO_F32: np.float32, _lib.METRO_F64: np.float64}
_PRECISIONS = {'f16': _lib.METRO_PREC_F16, 'f32': _lib.METRO_PREC_F32, 'f64': _lib.METRO_PREC_F64, 'f32m': _lib.METRO_PREC_F32M}


def _bn_scale_shift(params: Dict[str, np.ndarray], bn: str):
    g = params[bn + '/gamma'].astype(np.float64)
    b = params[bn + '/beta'].astype(np.float64)
    m = params[bn + '/moving_mean'].astype(np.float64)
    v = params[bn + '/moving_variance'].astype(np.float64)
    scale = g / np.sqrt(v + BN_EPS)
    return scale, b - m * scale


def pack_param(info: _lib.MetroParamInfo, params: Dict[str, np.ndarray]) -> np.ndarray:
    """One tensor of the plan's parameter blob, folded in fp64 and cast once.

    conv followed by BN (+ReLU):  w' = w * gamma/sqrt(var+eps) per c_out, b' = beta - mean*scale
    pre-activation BN (prologue): scale = gamma/sqrt(var+eps), shift = beta - mean*scale
    """
    conv = info.conv_var.decode()
    bn = info.bn_var.decode()
    dt = _NP_DTYPE[info.dtype]
    if info.kind == _lib.PARAM_CONV_W:
        w = params[conv + '/weights'].astype(np.float64)          # HWIO
        if w.shape != (info.kh, info.kw, info.c_in, info.c_out):
            raise ValueError(f'{conv}/weights has shape {w.shape}, plan expects '
                             f'{(info.kh, info.kw, info.c_in, info.c_out)}')
        if bn:
            w = w * _bn_scale_shift(params, bn)[0]
        packed = np.zeros((info.c_out, info.kh, info.kw_pad, info.c_in_pad), dtype=np.float64)
        packed[:, :, :info.kw, :info.c_in] = w.transpose(3, 0, 1, 2)
        return packed.astype(dt)
    if info.kind == _lib.PARAM_BIAS:
        v = _bn_scale_shift(params, bn)[1] if bn else params[conv + '/biases'].astype(np.float64)
    elif info.kind == _lib.PARAM_PRO_SCALE:
        v = _bn_scale_shift(params, bn)[0]
    elif info.kind == _lib.PARAM_PRO_SHIFT:
        v = _bn_scale_shift(params, bn)[1]
    else:
        raise ValueError(f'unknown parameter kind {info.kind}')
    if v.shape != (info.c_out,):
        raise ValueError(f'{info.name.decode()}: length {v.shape} != {info.c_out}')
    return v.astype(dt)


class Engine:
    """One plan on one GPU.  Not thread-safe (like the C plan it wraps)."""

    def __init__(self, spec: ModelSpec, params: Optional[Dict[str, np.ndarray]],
                 precision: str = 'f16', max_batch: int = 64, device: Optional[torch.device] = None):
        if precision not in _PRECISIONS:
            raise ValueError(f"precision must be 'f16', 'f32', 'f32m' or 'f64', got {precision!r}")
        self.lib = _lib.load()
        self.spec = spec
        self.precision = precision
        self.max_batch = int(max_batch)
        self._plan = C.c_void_p()
        cspec = spec.to_c(_PRECISIONS[precision])
        check(self.lib.metro_plan_create(C.byref(cspec), self.max_batch, C.byref(self._plan)),
              'metro_plan_create')
        self.cspec = cspec
        # hipGraph replay of the forward for batches <= METRO_HIPGRAPH_MAX_BATCH (default 0 = off:
        # measured on MI355X, batch 1..8 is bound by per-kernel dependency latency on the GPU, not by
        # host launches -- 0.661 ms eager vs 0.669 ms replayed, tools/graph_probe.py)
        import os
        gmax = int(os.environ.get('METRO_HIPGRAPH_MAX_BATCH', '0'))
        check(self.lib.metro_plan_set_graph_max_batch(self._plan, gmax), 'metro_plan_set_graph_max_batch')
        self.device = None
        self._blob = None
        self._ws = None
        if params is not None:
            if device is None:
                if not torch.cuda.is_available():
                    raise _lib.MetroError('no HIP device visible: the MeTRo hot path has no CPU fallback')
                device = torch.device('cuda', torch.cuda.current_device())
            self.bind(params, device)

    # ---- plan introspection (CPU-only safe) ---------------------------------------------
    def param_infos(self) -> List[_lib.MetroParamInfo]:
        out = []
        for i in range(self.lib.metro_plan_num_params(self._plan)):
            pi = _lib.MetroParamInfo()
            check(self.lib.metro_plan_param_info(self._plan, i, C.byref(pi)), 'metro_plan_param_info')
            out.append(pi)
        return out

    def layer_infos(self) -> List[_lib.MetroLayerInfo]:
        out = []
        for i in range(self.lib.metro_plan_num_layers(self._plan)):
            li = _lib.MetroLayerInfo()
            check(self.lib.metro_plan_layer_info(self._plan, i, C.byref(li)), 'metro_plan_layer_info')
            out.append(li)
        return out

    def layer_kernels(self, n: int) -> List[str]:
        """The kernel instantiation every layer runs on at batch n (metro_plan_layer_kernel: a dry run of the dispatch,
        no device needed).  The choice depends on the batch; tests/test_kernel_coverage.py holds every id to a test."""
        out = []
        buf = C.create_string_buffer(1024)
        for i in range(self.lib.metro_plan_num_layers(self._plan)):
            check(self.lib.metro_plan_layer_kernel(self._plan, i, int(n), buf, len(buf)), 'metro_plan_layer_kernel')
            out.append(buf.value.decode())
        return out

    @property
    def flops_per_image(self) -> float:
        return float(self.lib.metro_plan_flops_per_image(self._plan))

    @property
    def workspace_bytes(self) -> int:
        return int(self.lib.metro_plan_workspace_bytes(self._plan))

    @property
    def param_bytes(self) -> int:
        return int(self.lib.metro_plan_param_bytes(self._plan))

    def pack_params(self, params: Dict[str, np.ndarray]) -> np.ndarray:
        blob = np.zeros(self.param_bytes, dtype=np.uint8)
        for pi in self.param_infos():
            t = pack_param(pi, params)
            raw = np.ascontiguousarray(t).view(np.uint8).reshape(-1)
            if raw.size != pi.bytes:
                raise ValueError(f'{pi.name.decode()}: packed {raw.size} bytes, plan expects {pi.bytes}')
            blob[pi.offset:pi.offset + pi.bytes] = raw
        return blob

    # ---- device side -----------------------------------------------------------------------
    def bind(self, params: Dict[str, np.ndarray], device: torch.device) -> None:
        self.device = torch.device(device)
        blob = self.pack_params(params)
        with torch.cuda.device(self.device):
            self._blob = torch.from_numpy(blob).to(self.device)
            self._ws = torch.empty(self.workspace_bytes, dtype=torch.uint8, device=self.device)
            if self._blob.data_ptr() % 256 or self._ws.data_ptr() % 256:
                raise _lib.MetroError('torch returned a device buffer that is not 256-byte aligned')
            check(self.lib.metro_plan_bind_params(self._plan, C.c_void_p(self._blob.data_ptr())),
                  'metro_plan_bind_params')

    def _check_images(self, images: torch.Tensor) -> torch.Tensor:
        s = self.spec.proc_side
        if self._blob is None:
            raise _lib.MetroError('Engine has no parameters bound')
        if not isinstance(images, torch.Tensor):
            raise ValueError('images must be a torch.Tensor on the GPU')
        if images.dim() != 4 or tuple(images.shape[1:]) != (s, s, 3):
            raise ValueError(f'images must be [N,{s},{s},3] NHWC, got {tuple(images.shape)}')
        if images.dtype != torch.float32:
            raise ValueError(f'images must be float32 in [0,1], got {images.dtype}')
        if images.device != self.device:
            raise ValueError(f'images are on {images.device}, the plan is on {self.device}')
        if images.shape[0] < 1 or images.shape[0] > self.max_batch:
            raise ValueError(f'batch {images.shape[0]} outside [1, {self.max_batch}]')
        return images.contiguous()

    def forward(self, images: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """images fp32 [n,256,256,3] on the plan's device -> poses fp32 [n,Jout,3] (mm).  Enqueued
        on torch's current stream; no synchronisation."""
        images = self._check_images(images)
        n = images.shape[0]
        if out is None:
            out = torch.empty((n, self.spec.skeleton.n_out, 3), dtype=torch.float32, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.metro_forward(self._plan, C.c_void_p(images.data_ptr()), n,
                                     C.c_void_p(out.data_ptr()), C.c_void_p(self._ws.data_ptr()),
                                     C.c_void_p(stream)), 'metro_forward')
        return out

    def check_finite(self, n: int) -> None:
        """Non-finite screen of the last forward(n) on this engine: raises _lib.NonFiniteError when activations overflowed
        the arithmetic mode on their way to the soft-argmax (metro_forward_status).  Synchronises the current stream."""
        stream = torch.cuda.current_stream(self.device).cuda_stream
        bad = C.c_int32(0)
        check(self.lib.metro_forward_status(self._plan, C.c_void_p(self._ws.data_ptr()), int(n), C.c_void_p(stream), C.byref(bad)),
              f'{self.spec.arch_name} stride {self.spec.stride} in precision {self.precision!r}')

    def status_words(self, n: int) -> torch.Tensor:
        """Device view (int32 [n]) of the non-finite words the LAST forward(n) on this engine wrote (1 = that crop reached the
        soft-argmax with non-finite statistics).  No synchronisation: valid until the next forward, on the same stream."""
        off = int(self.lib.metro_plan_status_offset(self._plan))
        return self._ws[off:off + 4 * int(n)].view(torch.int32)

    def forward_upto(self, images: torch.Tensor, layer: int, second: bool = False) -> torch.Tensor:
        """Runs layers [0..layer] and returns that layer's output tensor [n,h,w,c] (a copy); `second` selects
        the second output of a fused launch (MetroLayerInfo.out2_offset)."""
        images = self._check_images(images)
        n = images.shape[0]
        li = self.layer_infos()[layer]
        poses = torch.empty((n, self.spec.skeleton.n_out, 3), dtype=torch.float32, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.metro_forward_upto(self._plan, C.c_void_p(images.data_ptr()), n,
                                          C.c_void_p(poses.data_ptr()), C.c_void_p(self._ws.data_ptr()),
                                          C.c_void_p(stream), layer), 'metro_forward_upto')
        if li.kind == _lib.LAYER_SOFTARGMAX:
            return poses
        dt = {_lib.METRO_F16: torch.float16, _lib.METRO_F32: torch.float32,
              _lib.METRO_F64: torch.float64}[li.out_dtype]
        if second:
            if li.out2_offset < 0:
                raise ValueError(f'{li.name.decode()}: layer has no second output')
            nb2 = li.h_out * li.w_out * li.out2_channels * 2 * n
            return self._ws[li.out2_offset:li.out2_offset + nb2].view(torch.float16).view(
                n, li.h_out, li.w_out, li.out2_channels).clone()
        nbytes = li.out_bytes_per_image * n
        raw = self._ws[li.out_offset:li.out_offset + nbytes]
        return raw.view(dt).view(n, li.h_out, li.w_out, li.c_out).clone()

    def forward_timed(self, images: torch.Tensor, reps: int = 1) -> np.ndarray:
        """Per-layer milliseconds (HIP events on the launch stream), averaged over `reps`."""
        images = self._check_images(images)
        n = images.shape[0]
        nl = self.lib.metro_plan_num_layers(self._plan)
        ms = (C.c_float * nl)()
        poses = torch.empty((n, self.spec.skeleton.n_out, 3), dtype=torch.float32, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        for _ in range(reps):
            check(self.lib.metro_forward_timed(self._plan, C.c_void_p(images.data_ptr()), n,
                                               C.c_void_p(poses.data_ptr()),
                                               C.c_void_p(self._ws.data_ptr()), C.c_void_p(stream), ms),
                  'metro_forward_timed')
        return np.asarray(ms, dtype=np.float64) / reps

    def close(self) -> None:
        if self._plan:
            if self._ws is not None and self.device is not None:
                torch.cuda.synchronize(self.device)      # queued launches still read the plan's parameter blob / workspace
            self.lib.metro_plan_destroy(self._plan)
            self._plan = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
