"""CPU restatement of the reference's crop pre-processing (row f2).  TEST INFRASTRUCTURE.

Follows src/cameralib.py:406-429 (`reproject_image_fast`): grid of output pixel coordinates
(x, y, 1) (float32, :400-403) -> `homography @ coords` (float32 matmul, :412-415) -> perspective divide (:416) ->
`cv2.remap(image, mapx, mapy, INTER_LINEAR, BORDER_CONSTANT, 0)` on the UINT8 frame (:424-425), then src/improc.py:56-61
(`normalize01`: float32 / 255, clip to [-1, 1]).  It is a BYTE path: the warped crop is uint8 before it is scaled.

OpenCV (`conda install opencv3 -c menpo`, install_dependencies.sh:12) is ABSENT from this image, so `remap` is restated
from its published source, modules/imgproc/src/imgwarp.cpp (3.x; remap.cpp in 4.x), the branch the reference's call takes
-- 8-bit image, two CV_32FC1 maps, INTER_LINEAR:
  * RemapInvoker converts the float maps to fixed point per pixel: sx = cvRound(mapx * INTER_TAB_SIZE), INTER_BITS = 5,
    INTER_TAB_SIZE = 32 (round half to even; NaN / out-of-int-range -> INT_MIN, the x86 "integer indefinite"); integer part
    saturate_cast<short>(sx >> 5), table index alpha = (sy & 31) * 32 + (sx & 31);
  * initInterTab2D(INTER_LINEAR, fixpt): 1-D weights (1 - k/32, k/32) as float, 2-D products scaled by
    INTER_REMAP_COEF_SCALE = 2^15 and saturate_cast<short> -- exact integers 32*(32-ay)*(32-ax), ... except the entry
    alpha = 0, whose 1.0 saturates to 32767; the table is then corrected to sum to 2^15 by adding the difference to the
    largest (or subtracting from the smallest) weight found by a 2 x 2 scan that starts at index [1][1] of the 2 x 2
    entry and runs past it into the not-yet-filled (zero) next entry: alpha = 0 becomes (32767, 0, 0, 1);
  * remapBilinear<FixedPtCast<int, uchar, 15>>: dst = saturate_cast<uchar>((S00*w0 + S01*w1 + S10*w2 + S11*w3 + 2^14)
    >> 15), taps outside the image replaced by the border value (0); all four outside -> the border value.
`homography @ coords` is NumPy's float32 matmul: with the BLAS of this image each element is
fma(h2, 1, fma(h1, y, rn(h0 * x))) (verified in tests/test_preprocess.py); the HIP kernel evaluates exactly that chain.

PARITY UNPINNED against cv2 itself (no OpenCV here to execute): pinned to the restated rule, bit for bit.
"""
from __future__ import annotations

import numpy as np

INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS
INTER_REMAP_COEF_BITS = 15
INTER_REMAP_COEF_SCALE = 1 << INTER_REMAP_COEF_BITS


def _saturate_short(v: int) -> int:
    return max(-32768, min(32767, int(v)))


def bilinear_tab_i() -> np.ndarray:
    """initInterTab2D(INTER_LINEAR, fixpt=true): int16 [1024, 4] (w00, w01 = x+1, w10 = y+1, w11)."""
    scale = np.float32(1.0) / np.float32(INTER_TAB_SIZE)
    tab1 = np.empty((INTER_TAB_SIZE, 2), np.float32)
    for i in range(INTER_TAB_SIZE):
        x = np.float32(i) * scale
        tab1[i] = (np.float32(1.0) - x, x)                               # interpolateLinear
    itab = np.zeros(INTER_TAB_SIZE * INTER_TAB_SIZE * 4 + 8, np.int64)   # flat and zero-initialised like the static array
    for i in range(INTER_TAB_SIZE):
        for j in range(INTER_TAB_SIZE):
            base = (i * INTER_TAB_SIZE + j) * 4
            isum = 0
            for k1 in range(2):
                vy = tab1[i, k1]
                for k2 in range(2):
                    v = np.float32(vy * tab1[j, k2])
                    q = _saturate_short(np.rint(np.float32(v * np.float32(INTER_REMAP_COEF_SCALE))))
                    itab[base + k1 * 2 + k2] = q
                    isum += q
            if isum != INTER_REMAP_COEF_SCALE:
                diff = isum - INTER_REMAP_COEF_SCALE
                ksize, ksize2 = 2, 1
                mk = Mk = (ksize2, ksize2)
                for k1 in range(ksize2, ksize2 + 2):
                    for k2 in range(ksize2, ksize2 + 2):
                        cur = itab[base + k1 * ksize + k2]
                        if cur < itab[base + mk[0] * ksize + mk[1]]:
                            mk = (k1, k2)
                        elif cur > itab[base + Mk[0] * ksize + Mk[1]]:
                            Mk = (k1, k2)
                if diff < 0:
                    itab[base + Mk[0] * ksize + Mk[1]] -= diff
                else:
                    itab[base + mk[0] * ksize + mk[1]] -= diff
    return itab[:INTER_TAB_SIZE * INTER_TAB_SIZE * 4].reshape(-1, 4)


_TAB = None


def cv_round_x86(v: np.ndarray) -> np.ndarray:
    """cvRound on float32 as OpenCV's x86 builds evaluate it (cvtss2si): round half to even; NaN and values outside int32
    give INT_MIN."""
    v = np.asarray(v, np.float32)
    r = np.rint(v.astype(np.float64))
    bad = ~np.isfinite(v) | (r >= 2.0 ** 31) | (r < -2.0 ** 31)
    return np.where(bad, -2.0 ** 31, r).astype(np.int64)


def remap_u8_linear_constant0(image_u8: np.ndarray, mapx: np.ndarray, mapy: np.ndarray) -> np.ndarray:
    """cv2.remap(image_u8 [H, W, C], mapx, mapy (float32 [h, w]), INTER_LINEAR, BORDER_CONSTANT, 0) -> uint8 [h, w, C]."""
    global _TAB
    if _TAB is None:
        _TAB = bilinear_tab_i()
    assert image_u8.dtype == np.uint8 and image_u8.ndim == 3
    h, w = image_u8.shape[:2]
    sx = cv_round_x86(np.asarray(mapx, np.float32) * np.float32(INTER_TAB_SIZE))
    sy = cv_round_x86(np.asarray(mapy, np.float32) * np.float32(INTER_TAB_SIZE))
    alpha = (sy & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (sx & (INTER_TAB_SIZE - 1))
    x0 = np.clip(sx >> INTER_BITS, -32768, 32767)
    y0 = np.clip(sy >> INTER_BITS, -32768, 32767)
    wts = _TAB[alpha]                                                     # [h, w, 4]
    acc = np.zeros(mapx.shape + (image_u8.shape[2],), np.int64)
    img = image_u8.astype(np.int64)
    for k, (dy, dx) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
        xx, yy = x0 + dx, y0 + dy
        ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
        tap = img[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)]
        acc += np.where(ok, wts[..., k], 0)[..., None] * tap              # outside taps read the border value 0
    out = (acc + (1 << (INTER_REMAP_COEF_BITS - 1))) >> INTER_REMAP_COEF_BITS
    return np.clip(out, 0, 255).astype(np.uint8)


def crop_coordinates(homography: np.ndarray, side: int):
    """mapx, mapy float32 [side, side] exactly as cameralib.py:400-417 computes them (NumPy float32 matmul and divide)."""
    hmat = np.asarray(homography, dtype=np.float32)                       # cameralib.py:412 (.astype(np.float32))
    y, x = np.mgrid[:side, :side].astype(np.float32)                       # cameralib.py:402
    coords = np.stack([x, y, np.ones_like(x)], axis=0).reshape(3, -1)
    c = hmat @ coords                                                      # float32 matmul like the reference
    with np.errstate(divide='ignore', invalid='ignore'):
        uv = (c[:2] / c[2:]).reshape(2, side, side)                        # cameralib.py:416-417
    assert uv.dtype == np.float32
    return uv[0], uv[1]


def reproject_image_u8(image_u8: np.ndarray, homography: np.ndarray, side: int) -> np.ndarray:
    """reproject_image_fast (cameralib.py:406-429) -> uint8 [side, side, C]."""
    mapx, mapy = crop_coordinates(homography, side)
    return remap_u8_linear_constant0(image_u8, mapx, mapy)


def reproject_image_fast(image_u8: np.ndarray, homography: np.ndarray, side: int) -> np.ndarray:
    """-> float32 [side, side, 3] in [0, 1]: reproject_image_fast + normalize01 (improc.py:56-61)."""
    im = reproject_image_u8(image_u8, homography, side).astype(np.float32)
    im /= np.float32(255)
    return np.minimum(np.maximum(np.float32(-1), im), np.float32(1))
