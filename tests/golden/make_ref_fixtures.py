#!/usr/bin/env python3
"""Vectors computed BY THE REFERENCE ITSELF, in the build container (the only place /root/reference exists):

    python tests/golden/make_ref_fixtures.py        # writes tests/golden/ref_heads_metrics_v1.npz

Two pieces of the reference need nothing but NumPy / SciPy and therefore CAN run here:
  * src/eval/procrustes.py (procrustes, :6-107) -- imported as a module from its file;
  * src/model/bone_length_based_backproj.py: optimize_z_offset_by_bones_single (:38-62).  Its module imports
    TensorFlow at the top, so the function's own source lines are cut out with `ast` and executed in a namespace
    that holds `np` and `scipy` (nothing else is referenced by the function).
Only inputs and the reference's outputs are stored (data, not source); the reference never travels.  The tests
(tests/test_ref_fixtures.py) pin oracle/metrics.py, oracle/heads.py and the HIP kernels of rows f3/f4 to these vectors.
"""
from __future__ import annotations

import ast
import importlib.util
import os
import sys
import warnings

import numpy as np
import scipy
import scipy.optimize

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ref_heads_metrics_v1.npz')
H36M_EDGES = [(9, 8), (8, 7), (7, 10), (10, 11), (11, 12), (7, 13), (13, 14), (14, 15), (7, 6), (6, 16), (16, 3), (3, 4),
              (4, 5), (16, 0), (0, 1), (1, 2)]      # head order, reference src/data/h36m.py:25-31


def load_procrustes():
    spec = importlib.util.spec_from_file_location('ref_procrustes', os.path.join(REF, 'src/eval/procrustes.py'))
    mod = importlib.util.module_from_spec(spec)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')           # `is not 'best'` SyntaxWarning of the reference source
        spec.loader.exec_module(mod)
    return mod.procrustes


def load_bone_solver():
    path = os.path.join(REF, 'src/model/bone_length_based_backproj.py')
    src = open(path).read()
    tree = ast.parse(src)
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'optimize_z_offset_by_bones_single')
    code = compile(ast.Module(body=[fn], type_ignores=[]), path, 'exec')
    ns = {'np': np, 'scipy': scipy}
    exec(code, ns)
    return ns['optimize_z_offset_by_bones_single']


def rot(rng):
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    return q * np.sign(np.linalg.det(q))


def main():
    if not os.path.isdir(REF):
        sys.exit('the reference tree is only present in the build container')
    rng = np.random.default_rng(20260929)
    out = {}
    # ---- procrustes(X = true, Y = pred, scaling=True, reflection=False): the call of util3d.rigid_align (:139-159)
    procrustes = load_procrustes()
    n, nj = 24, 17
    true = rng.standard_normal((n, nj, 3)) * 250
    pred = true + rng.standard_normal((n, nj, 3)) * 60
    pred[:6] = np.stack([1.3 * t @ rot(rng) + rng.standard_normal(3) * 200 for t in true[:6]])     # exact similarity
    pred[6:10] = true[6:10] * np.array([-1.0, 1.0, 1.0])                                           # mirrored poses
    true32, pred32 = true.astype(np.float32), pred.astype(np.float32)
    zs, ts, bs, cs, ds, al = [], [], [], [], [], []
    for i in range(n):
        d, z, tform = procrustes(true32[i].astype(np.float64), pred32[i].astype(np.float64), scaling=True, reflection=False)
        zs.append(z); ts.append(tform['rotation']); bs.append(tform['scale']); cs.append(tform['translation']); ds.append(d)
        al.append(tform['scale'] * pred32[i].astype(np.float64) @ tform['rotation'] + tform['translation'])  # util3d.py:159
    out.update({'pa/true': true32, 'pa/pred': pred32, 'pa/Z': np.array(zs), 'pa/rotation': np.array(ts),
                'pa/scale': np.array(bs), 'pa/translation': np.array(cs), 'pa/d': np.array(ds), 'pa/aligned': np.array(al)})
    # ---- optimize_z_offset_by_bones_single: fp32 rays / delta_z as TF hands them to the py_func, fp64 targets
    solve = load_bone_solver()
    # The TF ops in front of the py_func (heatmap_to_image, the inv_intrinsics einsum, delta_z: volumetric.py:171-181,
    # 288-295) cannot run here; the rays and delta_z are produced from head outputs (coords01) and inverse intrinsics by
    # this repo's restatement of them (oracle/heads.py, fp32 like TF) and stored, so the REFERENCE function's input is
    # part of the fixture and a test can start either at coords01 (HIP path) or at the rays (oracle).
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from oracle import heads as OH
    m, stride = 48, 16
    pose = rng.normal(0, 300, (m, nj, 3))
    pose[..., 2] += rng.uniform(1500, 6000, (m, 1))
    f = rng.uniform(900, 1400, m)
    kk = np.zeros((m, 3, 3)); kk[:, 0, 0] = f; kk[:, 1, 1] = f; kk[:, 0, 2] = 128; kk[:, 1, 2] = 128; kk[:, 2, 2] = 1
    uv = np.einsum('nij,ncj->nci', kk, pose / pose[..., 2:3])[..., :2]
    c01 = np.empty((m, nj, 3), np.float32)
    c01[..., :2] = ((uv - stride // 2) / 239.0).astype(np.float32)
    c01[..., 2] = ((pose[..., 2] - pose[:, -1:, 2]) / 2200.0 + 0.5 + rng.normal(0, 0.01, (m, nj))).astype(np.float32)
    inv_k = np.linalg.inv(kk).astype(np.float32)
    x, delta_z = OH.camcoords_and_delta_z(c01, inv_k, stride)
    bones = np.array([[np.linalg.norm(pose[i, a] - pose[i, b]) for a, b in H36M_EDGES] for i in range(m)])
    target_mean = bones.mean(axis=0) * 1.03
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')           # float(ndarray) deprecation inside the reference function
        z_mean = np.array([solve(x[i], delta_z[i], target_mean, H36M_EDGES) for i in range(m)])
        z_true = np.array([solve(x[i], delta_z[i], bones[i], H36M_EDGES) for i in range(m)])
    out.update({'bl/coords01': c01, 'bl/inv_intrinsics': inv_k, 'bl/stride': np.int32(stride), 'bl/x': x, 'bl/delta_z': delta_z, 'bl/edges': np.array(H36M_EDGES, np.int32), 'bl/target_mean': target_mean,
                'bl/target_per_pose': bones, 'bl/z_mean_targets': z_mean, 'bl/z_per_pose_targets': z_true})
    out['versions'] = np.array([np.__version__, scipy.__version__])
    np.savez_compressed(OUT, **out)
    print(f'wrote {OUT} ({os.path.getsize(OUT) / 1024:.1f} KiB); numpy {np.__version__}, scipy {scipy.__version__}')


if __name__ == '__main__':
    main()
