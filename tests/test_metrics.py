"""Row f4: evaluation metrics (MPJPE, Procrustes-MPJPE, PCK, AUC) -- oracle known answers on CPU,
HIP kernels vs oracle on the GPU."""
import numpy as np
import pytest

from oracle.metrics import eval_metrics as oracle_metrics


def _poses(n=50, nj=17, seed=0):
    rng = np.random.default_rng(seed)
    true = (rng.standard_normal((n, nj, 3)) * 250).astype(np.float32)
    pred = true + (rng.standard_normal((n, nj, 3)) * 60).astype(np.float32)
    valid = rng.random((n, nj)) > 0.15
    valid[:, :4] = True
    return pred, true, valid


def _rot(rng):
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    return q * np.sign(np.linalg.det(q))


def test_rigid_motion_and_scale_are_removed_by_alignment():
    rng = np.random.default_rng(1)
    _, true, _ = _poses(8)
    pred = np.stack([(1.7 * t @ _rot(rng) + rng.standard_normal(3) * 300) for t in true]).astype(np.float32)
    m = oracle_metrics(pred, true)
    assert m['mean_error'] > 50 and m['mean_error_procrustes'] < 1e-3


def test_reflection_is_not_allowed():
    _, true, _ = _poses(4, seed=2)
    pred = true * np.array([-1, 1, 1], np.float32)             # mirrored pose
    m = oracle_metrics(pred, true)
    assert m['mean_error_procrustes'] > 10                      # a reflection would give 0


def test_plain_metrics_known_values():
    true = (np.random.default_rng(0).standard_normal((2, 3, 3)) * 100).astype(np.float32)
    pred = true.copy()
    pred[0, 0] += [30, 40, 0]         # 50 mm
    pred[1, 1] += [0, 0, 300]         # 300 mm
    m = oracle_metrics(pred, true, np.ones((2, 3), bool))
    assert np.isclose(m['mean_error'], (50 + 300) / 6)
    assert np.allclose(m['pck'], [1, 0.5, 1]) and np.isclose(m['mean_pck'], 5 / 6)
    assert np.allclose(m['auc'], [(1 - 50 / 150 + 1) / 2, 0.5, 1])
    # root-relative: moving the root (last joint) moves every other joint's error
    pred2 = pred.copy()
    pred2[:, 2] += [0, 10, 0]
    d2 = oracle_metrics(pred2, true)['dist']
    assert np.isclose(d2[0, 2], 0) and np.isclose(d2[1, 0], 10, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize('masked', [False, True])
def test_hip_metrics_match_oracle(cuda, masked):
    import torch
    from metro_pose3d_amd.metrics import eval_metrics
    pred, true, valid = _poses(300, 19, seed=3)
    rng = np.random.default_rng(4)
    pred[:20] = np.stack([(1.3 * t @ _rot(rng) + 100) for t in true[:20]])       # exact similarity transforms
    pred[20:30] = true[20:30] * np.array([-1, 1, 1], np.float32)                 # reflections
    v = valid if masked else None
    ref = oracle_metrics(pred, true, v)
    got = eval_metrics(torch.from_numpy(pred).to(cuda), torch.from_numpy(true).to(cuda),
                       torch.from_numpy(valid).to(cuda) if masked else None)
    assert np.abs(got['dist'].cpu().numpy() - ref['dist']).max() < 1e-3
    assert np.abs(got['dist_procrustes'].cpu().numpy() - ref['dist_procrustes']).max() < 2e-3, \
        np.abs(got['dist_procrustes'].cpu().numpy() - ref['dist_procrustes']).max()
    for k in ('mean_error', 'mean_error_procrustes', 'mean_auc', 'mean_pck'):
        assert abs(got[k] - ref[k]) < 1e-4 * max(1.0, abs(ref[k])), k
    assert np.allclose(got['auc'], ref['auc'], atol=1e-5) and np.allclose(got['pck'], ref['pck'], atol=1e-6)
