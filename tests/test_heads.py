"""Alternative decode heads.  CPU part: the oracle (reference code restated with
NumPy + the reference's own scipy call) against the independent lmder restatement and analytical known answers.
GPU part: the HIP path through the C ABI against the oracle."""
import numpy as np
import pytest
import scipy.optimize

from metro_pose3d_amd import ModelSpec
from oracle import heads as OH
from oracle.lm1 import lmder1
from oracle.spec import head_joint_info


def _poses(rng, n, j):
    p = rng.normal(0, 300, (n, j, 3))
    p[..., 2] += rng.uniform(1500, 6000, (n, 1))
    return p


def _problem(rng, spec, n):
    """Synthetic but geometrically consistent inputs: coords01 such that rays * depth reproduce a pose."""
    ji = head_joint_info(spec.dataset)
    j = ji.n_joints
    p = _poses(rng, n, j)
    f = rng.uniform(900, 1400, n)
    kk = np.zeros((n, 3, 3)); kk[:, 0, 0] = f; kk[:, 1, 1] = f; kk[:, 0, 2] = 128; kk[:, 1, 2] = 128; kk[:, 2, 2] = 1
    uv = np.einsum('nij,ncj->nci', kk, p / p[..., 2:3])[..., :2]
    last = spec.proc_side - 1
    lrc = last - (last % spec.stride) - 1
    c01 = np.empty((n, j, 3), np.float32)
    c01[..., :2] = ((uv - (spec.stride // 2 if spec.centered_stride else 0)) / lrc).astype(np.float32)
    c01[..., 2] = ((p[..., 2] - p[:, -1:, 2]) / 2200.0 + 0.5 + rng.normal(0, 0.01, (n, j))).astype(np.float32)
    inv_k = np.linalg.inv(kk).astype(np.float32)
    bones = np.array([[np.linalg.norm(p[i, a] - p[i, b]) for a, b in ji.edges] for i in range(n)])
    return ji, p, c01, inv_k, bones


def test_lmder_restatement_matches_scipy():
    """oracle/lm1.py follows MINPACK's control flow: same x as scipy.optimize.least_squares(method='lm') on the
    reference's residual / (inexact) Jacobian, to fp64 rounding, on 300 random poses."""
    rng = np.random.default_rng(0)
    spec = ModelSpec(50, 16, 'h36m')
    ji, p, c01, inv_k, bones = _problem(rng, spec, 300)
    cam, dz = OH.camcoords_and_delta_z(c01, inv_k, spec.stride)
    target = bones.mean(axis=0) * 1.03
    worst = 0.0
    for i in range(300):
        x, d_z = cam[i], dz[i]
        a = np.asarray([x[u] - x[v] for u, v in ji.edges]); y = x * d_z[:, None]
        b = np.asarray([y[u] - y[v] for u, v in ji.edges])
        c, d, e = np.sum(a ** 2, axis=1), np.sum(2 * a * b, axis=1), np.sum(b ** 2, axis=1)
        rec = lambda z: np.sqrt(z ** 2 * c + z * d + e)
        ref = OH.optimize_z_offset_by_bones_single(x, d_z, target, ji.edges)
        mine, info, nfev, _ = lmder1(lambda z: rec(np.float64(z)) - target, lambda z: (np.float64(z) * c + d) / rec(np.float64(z)), 2000.0)
        assert 1 <= info <= 4 and nfev < 100
        worst = max(worst, abs(mine - ref))
    assert worst <= 1e-9, worst


def test_bone_length_head_known_answers():
    """KA: exact per-pose bone lengths and noise-free delta_z -> the solve recovers the root depth, back_project
    the pose (to fp32 rounding of the rays); true-root-depth reproduces it by construction."""
    rng = np.random.default_rng(1)
    spec = ModelSpec(50, 16, 'h36m')
    ji, p, c01, inv_k, bones = _problem(rng, spec, 8)
    c01[..., 2] = ((p[..., 2] - p[:, -1:, 2]) / 2200.0 + 0.5).astype(np.float32)      # no depth noise
    out, z = OH.backproject_bone_lengths(c01, inv_k, bones, ji.edges, spec.stride)
    assert np.abs(z - p[:, -1, 2]).max() < 2.0                       # mm; fp32 rays at ~4 m
    assert np.abs(out - p).max() < 3.0
    out2 = OH.backproject_root_depth(c01, inv_k, p[:, -1, 2], spec.stride)
    assert np.abs(out2 - p).max() < 1.0


def test_to_orig_cam_mirrors_on_negative_determinant():
    ji = head_joint_info('h36m')
    rng = np.random.default_rng(2)
    x = rng.normal(0, 500, (2, ji.n_joints, 3)).astype(np.float32)
    rot = np.stack([np.eye(3), np.diag([-1.0, 1.0, 1.0])]).astype(np.float32)
    y = OH.to_orig_cam(x, rot, ji.mirror_mapping)
    assert np.array_equal(y[0], x[0])
    flipped = x[1] * np.array([-1, 1, 1], np.float32)
    assert np.array_equal(y[1], flipped[ji.mirror_mapping])
    assert ji.names[ji.mirror_mapping[ji.names.index('lwri')]] == 'rwri' and ji.mirror_mapping[ji.names.index('neck')] == ji.names.index('neck')


def test_skeleton_tables_agree_with_oracle():
    for ds in ('h36m', 'merged', 'many19'):
        sk = ModelSpec(50, 16, ds).skeleton
        ji = head_joint_info(ds)
        assert list(sk.head_mirror) == ji.mirror_mapping
        assert [tuple(e) for e in sk.head_edges] == [tuple(e) for e in ji.edges]


# ---------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize('spec', [ModelSpec(50, 16, 'h36m'), ModelSpec(50, 8, 'many19'), ModelSpec(50, 32, 'h36m', centered_stride=False)],
                         ids=['h36m-s16', 'many19-s8', 'h36m-s32-nc'])
def test_gpu_bone_length_head(cuda, spec):
    import torch
    from metro_pose3d_amd import heads as MH
    rng = np.random.default_rng(spec.stride)
    n = 130                                                          # > one 64-thread block of poses
    ji, p, c01, inv_k, bones = _problem(rng, spec, n)
    target = bones.mean(axis=0) * 0.97
    ref, zref = OH.backproject_bone_lengths(c01, inv_k, target, ji.edges, spec.stride, spec.proc_side, spec.centered_stride)
    got, z = MH.backproject_bone_lengths(torch.from_numpy(c01).to(cuda), inv_k, target, spec)
    got, z = got.cpu().numpy(), z.cpu().numpy()
    assert np.abs(z - zref).max() <= 1e-3, np.abs(z - zref).max()             # mm (fp32 z_offset at ~4000 mm: ulp 2.4e-4)
    assert np.abs(got - ref).max() <= 1e-3 * 4, np.abs(got - ref).max()
    # per-pose targets ('bone-lengths-true'), root-relative + export permutation
    ref2, _ = OH.backproject_bone_lengths(c01, inv_k, bones, ji.edges, spec.stride, spec.proc_side, spec.centered_stride)
    ref2 = OH.root_relative(ref2)[:, list(spec.skeleton.permutation)]
    got2, _ = MH.backproject_bone_lengths(torch.from_numpy(c01).to(cuda), inv_k, bones, spec, root_relative=True, permute=True)
    assert got2.shape == (n, spec.skeleton.n_out, 3)
    assert np.abs(got2.cpu().numpy() - ref2).max() <= 4e-3
    # true-root-depth
    ref3 = OH.backproject_root_depth(c01, inv_k, p[:, -1, 2], spec.stride, spec.proc_side, spec.centered_stride)
    got3 = MH.backproject_root_depth(torch.from_numpy(c01).to(cuda), inv_k, p[:, -1, 2], spec).cpu().numpy()
    assert np.abs(got3 - ref3).max() <= 1e-3


@pytest.mark.gpu
def test_gpu_coords01_and_to_orig_cam(cuda):
    import torch
    from metro_pose3d_amd import heads as MH
    from oracle.forward import soft_argmax01
    spec = ModelSpec(50, 16, 'h36m')
    rng = np.random.default_rng(5)
    logits = (rng.standard_normal((5, 16, 16, spec.n_head_channels)) * 4).astype(np.float32)
    got = MH.coords01_from_logits(torch.from_numpy(logits).to(cuda), spec, precise=1).cpu().numpy()
    ref = soft_argmax01(torch.from_numpy(logits).permute(0, 3, 1, 2).double(), spec.skeleton.n_head, spec.depth)[1].numpy()
    assert got.shape == ref.shape == (5, 17, 3)
    assert np.abs(got - ref).max() <= 1e-6
    c01 = rng.uniform(0, 1, (7, 17, 3)).astype(np.float32)
    assert np.array_equal(MH.heatmap_to_25d(torch.from_numpy(c01).to(cuda), spec).cpu().numpy(), OH.heatmap_to_25d(c01, spec.stride))
    ji = head_joint_info('h36m')
    x = rng.normal(0, 500, (6, 17, 3)).astype(np.float32)
    q, _ = np.linalg.qr(rng.normal(size=(6, 3, 3)))
    q[::2] *= np.sign(np.linalg.det(q[::2]))[:, None, None]                    # proper rotations
    q[1::2] *= -np.sign(np.linalg.det(q[1::2]))[:, None, None]                 # reflections
    rot = q.astype(np.float32)
    got = MH.to_orig_cam(torch.from_numpy(x).to(cuda), rot, ji.mirror_mapping).cpu().numpy()
    ref = OH.to_orig_cam(x, rot, ji.mirror_mapping)
    assert np.abs(got - ref).max() <= 1e-3


@pytest.mark.gpu
def test_gpu_head_argument_errors(cuda):
    import torch
    from metro_pose3d_amd import heads as MH
    spec = ModelSpec(50, 16, 'h36m')
    c = torch.zeros((2, 17, 3), device=cuda)
    with pytest.raises(ValueError):
        MH.backproject_bone_lengths(c, np.zeros((2, 3, 3)), np.ones(5), spec)            # wrong number of bones
    with pytest.raises(ValueError):
        MH.backproject_bone_lengths(c[:, :5], np.zeros((2, 3, 3)), np.ones(16), spec)    # wrong joint count
    with pytest.raises(ValueError):
        MH.to_orig_cam(c, np.zeros((2, 3, 3)), [0, 1, 2])
