"""CPU restatement of the reference's crop pre-processing (row f2).  TEST INFRASTRUCTURE.

Follows src/cameralib.py:406-429 (`reproject_image_fast`): grid of output pixel coordinates
(x, y, 1) -> homography (float32) -> perspective divide -> `cv2.remap(image, mapx, mapy,
INTER_LINEAR, BORDER_CONSTANT, 0)`, then src/improc.py:56-61 (`normalize01`: float32 / 255, clip to
[-1, 1]).

OpenCV (cv2, pinned `opencv-python` in the reference's docs/DEPENDENCIES.md) is ABSENT from this
image, so `remap` is restated from its published definition: dst(x,y) = bilinear interpolation of src
at (mapx, mapy) with out-of-image taps replaced by the border value.  OpenCV evaluates this in fixed
point for 8-bit images (coordinates rounded to 1/32 pixel, weights to 15 bits, result rounded to
uint8); that quantisation is NOT restated: this oracle interpolates exactly (float64), so it pins the
HIP kernel to the mathematical definition, and both may differ from real cv2 output by ~1 LSB
(1/255).  PARITY UNPINNED against cv2 itself.
"""
from __future__ import annotations

import numpy as np


def remap_bilinear_constant0(image: np.ndarray, mapx: np.ndarray, mapy: np.ndarray) -> np.ndarray:
    """image [H, W, C]; maps [h, w] of source coordinates; returns float64 [h, w, C]."""
    img = image.astype(np.float64)
    h, w = img.shape[:2]
    x0 = np.floor(mapx).astype(np.int64)
    y0 = np.floor(mapy).astype(np.int64)
    a = (mapx - x0).astype(np.float64)
    b = (mapy - y0).astype(np.float64)
    out = np.zeros(mapx.shape + (img.shape[2],), dtype=np.float64)
    finite = np.isfinite(mapx) & np.isfinite(mapy)
    for dy in (0, 1):
        for dx in (0, 1):
            xx, yy = x0 + dx, y0 + dy
            wgt = (a if dx else 1 - a) * (b if dy else 1 - b)
            ok = finite & (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
            xs = np.clip(xx, 0, w - 1)
            ys = np.clip(yy, 0, h - 1)
            out += np.where(ok, wgt, 0.0)[..., None] * img[ys, xs]
    return out


def reproject_image_fast(image_u8: np.ndarray, homography: np.ndarray, side: int) -> np.ndarray:
    """-> float32 [side, side, 3] in [0, 1] (reproject_image_fast + normalize01)."""
    hmat = np.asarray(homography, dtype=np.float32)                       # cameralib.py:412 (.astype(np.float32))
    y, x = np.mgrid[:side, :side].astype(np.float32)                       # cameralib.py:402
    coords = np.stack([x, y, np.ones_like(x)], axis=0).reshape(3, -1)
    c = (hmat @ coords).astype(np.float32)                                 # float32 matmul like the reference
    uv = (c[:2] / c[2:]).astype(np.float32).reshape(2, side, side)         # cameralib.py:416-417
    warped = remap_bilinear_constant0(image_u8, uv[0], uv[1])
    return np.clip(warped / 255.0, -1.0, 1.0).astype(np.float32)           # improc.py:57-60
