"""Evaluation metrics on the GPU: the step after the hot path.

Mirrors `build_eval_metrics` of the reference (src/main.py:339-359): root-relative MPJPE
(`mean_error`), Procrustes-aligned MPJPE with scale and without reflection
(`mean_error_procrustes`), per-joint and mean PCK@150 mm and AUC.  Poses stay on the device.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from metro_pose3d_amd import _lib
from metro_pose3d_amd._lib import check


def eval_metrics(pred: torch.Tensor, true: torch.Tensor, valid: Optional[torch.Tensor] = None,
                 threshold_mm: float = 150.0) -> Dict[str, object]:
    """pred, true: fp32 [N, J, 3] on the GPU, root joint LAST (the convention inside the reference,
    tfu3d.py:23-25); valid: bool/uint8 [N, J] (default all valid)."""
    if not (isinstance(pred, torch.Tensor) and pred.is_cuda and pred.dtype == torch.float32 and pred.dim() == 3 and
            pred.shape[2] == 3):
        raise ValueError('pred must be a float32 [N, J, 3] tensor on the GPU')
    if true.shape != pred.shape or true.dtype != torch.float32 or true.device != pred.device:
        raise ValueError('true must match pred in shape, dtype and device')
    n, nj = pred.shape[0], pred.shape[1]
    if valid is None:
        valid = torch.ones((n, nj), dtype=torch.uint8, device=pred.device)
    valid = valid.to(torch.uint8).contiguous()
    if tuple(valid.shape) != (n, nj):
        raise ValueError(f'valid must be [N, J] = {(n, nj)}, got {tuple(valid.shape)}')
    pred, true = pred.contiguous(), true.contiguous()
    dist = torch.empty((n, nj), dtype=torch.float32, device=pred.device)
    dist_pa = torch.empty_like(dist)
    sums = torch.empty((nj, 5), dtype=torch.float64, device=pred.device)
    lib = _lib.load()
    stream = torch.cuda.current_stream(pred.device).cuda_stream
    check(lib.metro_eval_metrics(C.c_void_p(pred.data_ptr()), C.c_void_p(true.data_ptr()), C.c_void_p(valid.data_ptr()),
                                 n, nj, C.c_float(threshold_mm), C.c_void_p(dist.data_ptr()),
                                 C.c_void_p(dist_pa.data_ptr()), C.c_void_p(sums.data_ptr()), C.c_void_p(stream)),
          'metro_eval_metrics')
    s = sums.cpu().numpy()
    cnt = s[:, 0]
    with np.errstate(invalid='ignore', divide='ignore'):
        return {
            'mean_error': s[:, 1].sum() / cnt.sum(), 'mean_error_procrustes': s[:, 2].sum() / cnt.sum(),
            'auc': s[:, 3] / cnt, 'mean_auc': s[:, 3].sum() / cnt.sum(),
            'pck': s[:, 4] / cnt, 'mean_pck': s[:, 4].sum() / cnt.sum(),
            'dist': dist, 'dist_procrustes': dist_pa,
        }
