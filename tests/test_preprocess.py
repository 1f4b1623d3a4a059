"""Row f2: crop pre-processing (homography warp + bilinear + /255) -- oracle known answers on CPU,
HIP kernel vs oracle on the GPU."""
import numpy as np
import pytest

from metro_pose3d_amd.preprocess import box_homography, homography_between_cameras
from oracle.preprocess import reproject_image_fast


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


def test_half_pixel_is_the_mean_of_neighbours():
    img = _frame(8, 8)
    hom = np.array([[1, 0, 0.5], [0, 1, 0], [0, 0, 1]], np.float32)
    out = reproject_image_fast(img, hom, 8)
    exp = (img[:, :-1].astype(np.float64) + img[:, 1:]) / 2 / 255
    assert np.abs(out[:, :-1] - exp).max() < 1e-6
    assert np.abs(out[:, -1] - img[:, -1] / 510.0).max() < 1e-6           # right neighbour is border (0)


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
def test_hip_warp_matches_oracle(cuda):
    import torch
    from metro_pose3d_amd.preprocess import warp_crops
    img = _frame(480, 640, seed=5)
    k_old = np.array([[1100., 0, 320], [0, 1100, 240], [0, 0, 1]])
    ang = 0.15
    r_new = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    homs = [np.eye(3, dtype=np.float32), box_homography((100, 50, 300, 380)), box_homography((-40, -40, 200, 200)),
            homography_between_cameras(k_old, np.eye(3), np.array([[900., 0, 128], [0, 900, 128], [0, 0, 1]]), r_new)]
    got = warp_crops(torch.from_numpy(img).to(cuda), np.stack(homs), side=256).cpu().numpy()
    assert got.shape == (4, 256, 256, 3) and got.dtype == np.float32
    for i, h in enumerate(homs):
        ref = reproject_image_fast(img, h, 256)
        # fp32 interpolation vs exact: a few ulp of the coordinate (~500 px * 6e-8) times the local gradient
        assert np.abs(got[i] - ref).max() <= 2e-4, (i, np.abs(got[i] - ref).max())
    assert np.array_equal(got[0][:, :, :], (img[:256, :256].astype(np.float32) / np.float32(255)))
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
    assert np.abs(poses - ref).max() <= 0.05      # mm: the 2e-4 crop tolerance propagated through the net
