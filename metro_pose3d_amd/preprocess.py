"""Crop pre-processing on the GPU: the step before the hot path.

Reference: `cameralib.reproject_image_fast` (src/cameralib.py:406-429) warps the frame through the
homography between the original camera and a virtual camera that looks at the person box
(src/data/data_loading.py:43-58,93), then `improc.normalize01` (src/improc.py:56-61) scales to [0,1].
The demo in inference.py:17 simply resizes the whole frame.  Here the homographies are computed on
the host (3x3 each) and the sampling runs in one HIP kernel (`metro_warp_crop_u8`) that writes the
fp32 NHWC crops `metro_forward` consumes, so a uint8 frame never round-trips through host memory.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np
import torch

from metro_pose3d_amd import _lib
from metro_pose3d_amd._lib import check


def homography_between_cameras(k_old, r_old, k_new, r_new) -> np.ndarray:
    """Maps NEW-camera pixel (x, y, 1) to OLD-camera pixel coordinates; both cameras share the optical
    centre.  Same expression as the reference: solve(new_matrix.T, old_matrix.T).T with matrix = K @ R
    (src/cameralib.py:410-412), cast to float32 like there."""
    old_m = np.asarray(k_old, np.float64) @ np.asarray(r_old, np.float64)
    new_m = np.asarray(k_new, np.float64) @ np.asarray(r_new, np.float64)
    return np.linalg.solve(new_m.T, old_m.T).T.astype(np.float32)


def box_homography(box: Sequence[float], side: int = 256) -> np.ndarray:
    """Axis-aligned crop of the square that contains box = (x, y, w, h), centred on it and scaled to
    side x side (the no-camera special case: pure zoom + shift; output pixel centres map to
    source = corner + (dst + 0.5) * scale - 0.5)."""
    x, y, w, h = (float(v) for v in box)
    crop = max(w, h)
    cx, cy = x + w / 2, y + h / 2
    s = crop / side
    return np.array([[s, 0, cx - crop / 2 + 0.5 * s - 0.5],
                     [0, s, cy - crop / 2 + 0.5 * s - 0.5],
                     [0, 0, 1]], dtype=np.float32)


def warp_crops(image_u8: torch.Tensor, homographies, side: int = 256, out: torch.Tensor = None) -> torch.Tensor:
    """image_u8: uint8 [H, W, 3] on the GPU; homographies: [n, 3, 3] (array-like or tensor).
    Returns fp32 [n, side, side, 3] in [0, 1] on the same device (enqueued on the current stream)."""
    if not isinstance(image_u8, torch.Tensor) or not image_u8.is_cuda:
        raise ValueError('image must be a uint8 torch.Tensor on the GPU')
    if image_u8.dtype != torch.uint8 or image_u8.dim() != 3 or image_u8.shape[2] != 3:
        raise ValueError(f'image must be uint8 [H, W, 3], got {image_u8.dtype} {tuple(image_u8.shape)}')
    image_u8 = image_u8.contiguous()
    hom = torch.as_tensor(np.asarray(homographies, dtype=np.float32) if not isinstance(homographies, torch.Tensor)
                          else homographies, dtype=torch.float32).reshape(-1, 9).to(image_u8.device).contiguous()
    n = hom.shape[0]
    if out is None:
        out = torch.empty((n, side, side, 3), dtype=torch.float32, device=image_u8.device)
    lib = _lib.load()
    h, w = image_u8.shape[0], image_u8.shape[1]
    stream = torch.cuda.current_stream(image_u8.device).cuda_stream
    check(lib.metro_warp_crop_u8(C.c_void_p(image_u8.data_ptr()), h, w, 3 * w, C.c_void_p(hom.data_ptr()), n, side,
                                 C.c_void_p(out.data_ptr()), C.c_void_p(stream)), 'metro_warp_crop_u8')
    return out
