"""Row f2: crop pre-processing (homography warp + cv2.remap's 8-bit bilinear rule + /255) -- the restated rule's known answers
on CPU, HIP kernel vs oracle BIT FOR BIT on the GPU (the reference's path is a byte path: cameralib.py:406-429 warps the uint8
frame, improc.py:56-61 scales the bytes)."""
import numpy as np
import pytest

from metro_pose3d_amd.preprocess import box_homography, homography_between_cameras
from oracle.preprocess import (bilinear_tab_i, crop_coordinates, remap_u8_linear_constant0, reproject_image_fast,
                               reproject_image_u8)


def _frame(h=120, w=160, seed=0):
    return np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)


def test_identity_homography_is_a_copy():
    img = _frame(64, 64)
    out = reproject_image_fast(img, np.eye(3), 64)
    assert np.array_equal(out, img.astype(np.float32) / np.float32(255))


def test_integer_shift_and_border_zero():
    img = _frame(64, 64)
    hom = np.array([[1, 0, 5], [0, 1, -3], [0, 0, 1]], np.float32)       # out(x,y) = src(x+5, y-3)
    out = reproject_image_fast(img, hom, 64)
    assert np.array_equal(out[3:, :59], (img[:61, 5:] / np.float32(255)).astype(np.float32))
    assert (out[:3] == 0).all() and (out[:, 59:] == 0).all()              # BORDER_CONSTANT 0


def test_half_pixel_is_the_mean_of_neighbours_rounded_half_up():
    img = _frame(8, 8)
    hom = np.array([[1, 0, 0.5], [0, 1, 0], [0, 0, 1]], np.float32)
    out = reproject_image_u8(img, hom, 8)
    a, b = img[:, :-1].astype(np.int64), img[:, 1:].astype(np.int64)
    assert np.array_equal(out[:, :-1], (a + b + 1) // 2)                  # (16384 a + 16384 b + 2^14) >> 15
    assert np.array_equal(out[:, -1], (img[:, -1].astype(np.int64) + 1) // 2)   # right neighbour is the border (0)
    f = reproject_image_fast(img, hom, 8)
    assert f.dtype == np.float32 and np.array_equal(f, out.astype(np.float32) / np.float32(255))


def test_fixed_point_rules_of_cv_remap():
    """INTER_BITS = 5: coordinates snap to 1/32 px with round-half-to-even, weights are 15-bit, the result is a byte."""
    tab = bilinear_tab_i()
    assert tab.shape == (1024, 4) and (tab.sum(axis=1) == 32768).all() and tab[0].tolist() == [32767, 0, 0, 1]
    ay, ax = np.divmod(np.arange(1024), 32)
    assert np.array_equal(tab[1:], np.stack([32 * (32 - ay) * (32 - ax), 32 * (32 - ay) * ax, 32 * ay * (32 - ax), 32 * ay * ax], 1)[1:])
    img = np.zeros((4, 6, 1), np.uint8)
    img[1, 2, 0], img[1, 3, 0] = 200, 100
    at = lambda x, y: int(remap_u8_linear_constant0(img, np.array([[x]], np.float32), np.array([[y]], np.float32))[0, 0, 0])
    assert at(2.0, 1.0) == 200 and at(3.0, 1.0) == 100
    assert at(2.25, 1.0) == 175                                          # 8/32: (24 * 200 + 8 * 100) / 32
    assert at(2 + 1 / 64, 1.0) == at(2.0, 1.0)                           # 0.5/32 rounds to even: 64.5 -> 64
    assert at(2 + 3 / 64, 1.0) == at(2 + 2 / 32, 1.0)                    # 65.5 -> 66
    assert at(2.01, 1.0) == 200 and at(2.02, 1.0) == (31 * 200 + 100 + 16) // 32        # 0.32 -> 0, 0.64 -> 1 thirty-second
    assert at(-0.5, 1.0) == 0 and at(5.5, 1.0) == 0 and at(np.nan, 1.0) == 0 and at(1e30, 1.0) == 0 and at(2.0, -np.inf) == 0
    assert at(2.0, 0.5) == 100 and at(2.0, 3.75) == 0                    # vertical blend with zero rows; below the last row: border


def test_coordinates_are_the_fma_chain_of_numpy_matmul():
    """cameralib.py:412-416 evaluates `homography @ coords` with NumPy's float32 matmul; the HIP kernel reproduces its rounding
    (fma(h2, 1, fma(h1, y, rn(h0 x)))) -- checked here against NumPy itself so a BLAS with another order would be noticed."""
    rng = np.random.default_rng(1)
    hom = (rng.standard_normal((3, 3)) * np.array([[1, 0.1, 300], [0.1, 1, 200], [1e-4, 1e-4, 1]])).astype(np.float32)
    mapx, mapy = crop_coordinates(hom, 64)
    y, x = np.mgrid[:64, :64].astype(np.float64)
    rows = []
    for r in range(3):
        acc = (np.float64(hom[r, 0]) * x).astype(np.float32)
        acc = (np.float64(hom[r, 1]) * y + acc.astype(np.float64)).astype(np.float32)
        rows.append((np.float64(hom[r, 2]) + acc.astype(np.float64)).astype(np.float32))
    assert np.array_equal(mapx, rows[0] / rows[2]) and np.array_equal(mapy, rows[1] / rows[2])


def test_box_homography_maps_corners_like_a_resize():
    hom = box_homography((10, 20, 64, 32), side=256)                       # square side 64 around centre (42, 36)
    s = 64 / 256
    assert np.allclose(hom @ [0, 0, 1], [10 + 0.5 * s - 0.5, 4 + 0.5 * s - 0.5, 1])
    assert np.allclose((hom @ [255, 255, 1])[:2], [10 + 255.5 * s - 0.5, 4 + 255.5 * s - 0.5])


def test_camera_homography_matches_reference_expression():
    rng = np.random.default_rng(3)
    k_old = np.array([[1100., 0, 500], [0, 1100, 480], [0, 0, 1]])
    k_new = np.array([[2400., 0, 128], [0, 2400, 128], [0, 0, 1]])
    ang = 0.1
    r_new = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    hom = homography_between_cameras(k_old, np.eye(3), k_new, r_new)
    ray = rng.standard_normal(3) + [0, 0, 5]
    p_old = k_old @ ray
    p_new = k_new @ r_new @ ray
    back = hom @ (p_new / p_new[2])
    assert np.allclose(back[:2] / back[2], p_old[:2] / p_old[2], atol=1e-2)


@pytest.mark.gpu
def test_hip_warp_matches_oracle_to_the_byte(cuda):
    import torch
    from metro_pose3d_amd.preprocess import warp_crops
    img = _frame(480, 640, seed=5)
    k_old = np.array([[1100., 0, 320], [0, 1100, 240], [0, 0, 1]])
    ang = 0.15
    r_new = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    homs = [np.eye(3, dtype=np.float32), box_homography((100, 50, 300, 380)), box_homography((-40, -40, 200, 200)),
            homography_between_cameras(k_old, np.eye(3), np.array([[900., 0, 128], [0, 900, 128], [0, 0, 1]]), r_new),
            np.array([[1, 0, 700], [0, 1, -300], [0, 0, 1]], np.float32),                 # entirely out of the frame
            np.array([[1, 0, 400], [0, 1, 240], [0, 0, 1]], np.float32),                  # exact grid, crossing the right/bottom edge
            np.array([[1.5, 0, -10.25], [0, 0.75, 3.5], [0, 0, 1]], np.float32),          # multiples of 1/32: half-to-even ties
            np.array([[1, 0, 0], [0, 1, 0], [-1 / 128., 0, 1]], np.float32),              # the divisor crosses zero inside the crop
            np.array([[3e4, 0, 0], [0, 3e4, 0], [0, 0, 1]], np.float32)]                  # coordinates beyond short and int range
    got = warp_crops(torch.from_numpy(img).to(cuda), np.stack(homs), side=256).cpu().numpy()
    assert got.shape == (len(homs), 256, 256, 3) and got.dtype == np.float32
    for i, h in enumerate(homs):
        ref = reproject_image_fast(img, h, 256)
        diff = got[i] != ref
        assert not diff.any(), f'homography {i}: {int(diff.sum())} of {diff.size} values differ, max {np.abs(got[i] - ref).max() * 255:.2f} LSB'
    assert np.array_equal(got[0], (img[:256, :256].astype(np.float32) / np.float32(255)))
    assert (got[4] == 0).all() and (got[5][:240, :240] > 0).any() and (got[5][:, 240:] == 0).all() and (got[5][240:] == 0).all()
    with pytest.raises(ValueError):
        warp_crops(torch.from_numpy(img), np.eye(3))


@pytest.mark.gpu
def test_warp_then_pose_end_to_end(cuda):
    """uint8 frame -> GPU crops -> poses, without touching host memory in between."""
    import torch
    from metro_pose3d_amd import ModelSpec, synth
    from metro_pose3d_amd.engine import Engine
    from metro_pose3d_amd.preprocess import warp_crops
    from oracle import forward as OF
    from tests import helpers as H
    spec = ModelSpec(50, 32, 'h36m', base_width=8)
    params = synth.make_params(50, spec.n_head_channels, 8, seed=1, logit_gain=0.84)
    img = _frame(300, 400, seed=8)
    homs = np.stack([box_homography((50, 20, 200, 260)), box_homography((150, 40, 220, 220))])
    crops = warp_crops(torch.from_numpy(img).to(cuda), homs)
    poses = Engine(spec, params, 'f64', max_batch=2, device=cuda).forward(crops).cpu().numpy()
    ref_crops = np.stack([reproject_image_fast(img, h, 256) for h in homs])
    ref = OF.forward(H.oracle_spec(spec), params, ref_crops, torch.float64).numpy()
    assert np.abs(poses - ref).max() <= 1e-3      # mm: the crops are bit-identical, the net runs in the parity mode
