"""Vectors computed by the REFERENCE's own code (tests/golden/make_ref_fixtures.py ran src/eval/procrustes.py and the
body of bone_length_based_backproj.optimize_z_offset_by_bones_single in the build container; only data travels).

CPU: the oracle restatements of rows f3/f4 reproduce them (this is what pins oracle/metrics.py and oracle/heads.py to
the reference rather than to themselves).  GPU (`-m gpu`): the HIP kernels behind metro_eval_metrics and
metro_backproject_bone_lengths reproduce them through the C ABI.
"""
import os

import numpy as np
import pytest

from oracle import heads as OH
from oracle import metrics as OM
from oracle.lm1 import lmder1

REF = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ref_heads_metrics_v1.npz')


@pytest.fixture(scope='module')
def ref():
    return np.load(REF)


def test_oracle_procrustes_matches_reference(ref):
    """reference src/eval/procrustes.py:6-107 with scaling=True, reflection=False (util3d.rigid_align :139-159)."""
    true, pred = ref['pa/true'].astype(np.float64), ref['pa/pred'].astype(np.float64)
    for i in range(true.shape[0]):
        t, b, c = OM.procrustes_no_reflection(true[i], pred[i])
        assert np.abs(t - ref['pa/rotation'][i]).max() < 1e-9
        assert abs(b - ref['pa/scale'][i]) < 1e-9 * max(1.0, abs(ref['pa/scale'][i]))
        assert np.abs(c - ref['pa/translation'][i]).max() < 1e-7
        assert np.abs(b * pred[i] @ t + c - ref['pa/aligned'][i]).max() < 1e-7
    # the fixture holds what it claims: exact similarity transforms align to ~0, mirrored poses do not
    err = np.linalg.norm(ref['pa/aligned'] - true, axis=-1).mean(axis=1)
    assert err[:6].max() < 1e-2 and err[6:10].min() > 10


def test_oracle_metrics_use_the_reference_alignment(ref):
    true, pred = ref['pa/true'], ref['pa/pred']
    m = OM.eval_metrics(pred, true)
    aligned = ref['pa/aligned'].astype(np.float32).astype(np.float64)           # the py_func returns float32
    rr = lambda d: d - d[:, -1:, :]
    want = np.linalg.norm(rr(aligned - true), axis=-1)
    assert np.abs(m['dist_procrustes'] - want).max() < 1e-4


MET_KEYS = ('mean_error', 'mean_error_procrustes', 'mean_pck', 'mean_auc', 'pck', 'auc')


def check_masked_metrics(got, ref, tol_dist, rtol):
    """build_eval_metrics (main.py:339-359) with a validity mask, against the numbers the reference's own
    rigid_align_many / root_relative / get_pck / get_auc produced (tests/golden/make_ref_fixtures.py)."""
    valid = ref['met/valid']
    for k in ('dist', 'dist_procrustes'):
        g = got[k].cpu().numpy() if hasattr(got[k], 'cpu') else got[k]
        assert np.abs(g - ref['met/' + k])[valid].max() < tol_dist, k
    for k in MET_KEYS:
        want = ref['met/' + k]
        assert np.allclose(np.asarray(got[k], np.float64), want, rtol=rtol, atol=0), (k, got[k], want)


def test_oracle_masked_metrics_match_reference(ref):
    # the py_func hands float32 arrays to procrustes, so the reference aligns in float32; the oracle (and the device
    # kernel) align in float64: they agree to float32 rounding of ~200 mm distances
    check_masked_metrics(OM.eval_metrics(ref['met/pred'], ref['met/true'], ref['met/valid']), ref, 1e-3, 2e-6)


def test_oracle_bone_length_solve_matches_reference(ref):
    """reference bone_length_based_backproj.py:38-62: scipy LM with the reference's (inexact) Jacobian."""
    x, dz, edges = ref['bl/x'], ref['bl/delta_z'], [tuple(e) for e in ref['bl/edges']]
    for key_t, key_z in (('bl/target_mean', 'bl/z_mean_targets'), ('bl/target_per_pose', 'bl/z_per_pose_targets')):
        t = ref[key_t]
        for i in range(x.shape[0]):
            ti = t if t.ndim == 1 else t[i]
            z = OH.optimize_z_offset_by_bones_single(x[i], dz[i], ti, edges)
            assert abs(z - ref[key_z][i]) <= 1e-9 * abs(ref[key_z][i]), (key_z, i)
            # and the independent MINPACK restatement the device solver follows
            a = np.asarray([x[i][u] - x[i][v] for u, v in edges]); y = x[i] * dz[i][:, None]
            b = np.asarray([y[u] - y[v] for u, v in edges])
            c, d, e = np.sum(a ** 2, axis=1), np.sum(2 * a * b, axis=1), np.sum(b ** 2, axis=1)
            rec = lambda zz: np.sqrt(zz ** 2 * c + zz * d + e)
            mine, info, _, _ = lmder1(lambda zz: rec(np.float64(zz)) - ti, lambda zz: (np.float64(zz) * c + d) / rec(np.float64(zz)), 2000.0)
            assert 1 <= info <= 4 and abs(mine - ref[key_z][i]) <= 1e-8 * abs(ref[key_z][i])


def test_fixture_prelude_is_the_oracle_prelude(ref):
    """The rays / delta_z stored next to coords01 are oracle/heads.camcoords_and_delta_z of them (fp32, bit-exact)."""
    cam, dz = OH.camcoords_and_delta_z(ref['bl/coords01'], ref['bl/inv_intrinsics'], int(ref['bl/stride']))
    assert np.array_equal(cam, ref['bl/x']) and np.array_equal(dz, ref['bl/delta_z'])


# ---------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_hip_metrics_match_reference_alignment(cuda, ref):
    import torch
    from metro_pose3d_amd.metrics import eval_metrics
    true, pred = ref['pa/true'], ref['pa/pred']
    got = eval_metrics(torch.from_numpy(pred).to(cuda), torch.from_numpy(true).to(cuda))
    aligned = ref['pa/aligned'].astype(np.float32).astype(np.float64)
    rr = lambda d: d - d[:, -1:, :]
    want = np.linalg.norm(rr(aligned - true), axis=-1)
    assert np.abs(got['dist_procrustes'].cpu().numpy() - want).max() < 2e-3
    assert abs(got['mean_error_procrustes'] - want.mean()) < 1e-4 * want.mean()


@pytest.mark.gpu
def test_hip_masked_metrics_match_reference(cuda, ref):
    import torch
    from metro_pose3d_amd.metrics import eval_metrics
    got = eval_metrics(torch.from_numpy(ref['met/pred']).to(cuda), torch.from_numpy(ref['met/true']).to(cuda),
                       torch.from_numpy(ref['met/valid']).to(cuda))
    check_masked_metrics(got, ref, 2e-3, 1e-5)


@pytest.mark.gpu
def test_hip_bone_length_solve_matches_reference(cuda, ref):
    import torch
    from metro_pose3d_amd import ModelSpec
    from metro_pose3d_amd import heads as MH
    spec = ModelSpec(50, int(ref['bl/stride']), 'h36m')
    assert [tuple(e) for e in spec.skeleton.head_edges] == [tuple(e) for e in ref['bl/edges']]
    c01 = torch.from_numpy(ref['bl/coords01']).to(cuda)
    for key_t, key_z in (('bl/target_mean', 'bl/z_mean_targets'), ('bl/target_per_pose', 'bl/z_per_pose_targets')):
        got, z = MH.backproject_bone_lengths(c01, ref['bl/inv_intrinsics'], ref[key_t], spec)
        zref = ref[key_z].astype(np.float32)                        # the py_func returns float32 (:16-18)
        assert np.abs(z.cpu().numpy() - zref).max() <= 2.5e-4 * 2, np.abs(z.cpu().numpy() - zref).max()   # <= one fp32 ulp at ~4 m
        want = OH.back_project(ref['bl/x'], ref['bl/delta_z'], zref)
        assert np.abs(got.cpu().numpy() - want).max() <= 2e-3


# ---- joint tables computed by the reference's own JointInfo / permute_joints code ------------------------------------
@pytest.mark.parametrize('dataset', ['h36m', 'merged'])
def test_joint_tables_match_reference(ref, dataset):
    """reference datasets.py:52-109 (JointInfo), h36m.py:25-31 / datasets.py:142-154 (names, edges), main.py:119-141
    (export permutation, permuted names and re-indexed edges): the oracle's restatement (oracle/spec.py) and the
    product's tables (metro_pose3d_amd/joints.py) against what the reference's code computed."""
    from metro_pose3d_amd.joints import skeleton
    from oracle.spec import export_permutation, head_joint_info, output_joint_info
    t = {k.split('/')[-1]: ref[k] for k in ref.files if k.startswith(f'joints/{dataset}/')}
    head, out = head_joint_info(dataset), output_joint_info(dataset)
    assert [n.encode() for n in head.names] == list(t['head_names'])
    assert [tuple(e) for e in head.edges] == [tuple(e) for e in t['head_edges'].tolist()]
    assert head.mirror_mapping == t['mirror'].tolist()
    assert export_permutation(dataset) == t['permutation'].tolist()
    assert [n.encode() for n in out.names] == list(t['out_names'])
    assert [tuple(e) for e in out.edges] == [tuple(e) for e in t['out_edges'].tolist()]
    sk = skeleton(dataset)
    assert sk.n_head == len(t['head_names']) and sk.n_out == len(t['out_names'])
    assert list(sk.names_bytes()) == list(t['out_names'])
    assert np.array_equal(sk.edges_array(), t['out_edges']) and sk.edges_array().dtype == np.int64
    assert list(sk.permutation) == t['permutation'].tolist()
    assert list(sk.head_mirror) == t['mirror'].tolist()
    assert [tuple(e) for e in sk.head_edges] == [tuple(e) for e in t['head_edges'].tolist()]
