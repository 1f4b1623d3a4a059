#!/usr/bin/env python3
"""Vectors computed BY THE REFERENCE ITSELF, in the build container (the only place /root/reference exists):

    python tests/golden/make_ref_fixtures.py        # writes tests/golden/ref_heads_metrics_v1.npz

Two pieces of the reference need nothing but NumPy / SciPy and therefore CAN run here:
  * src/eval/procrustes.py (procrustes, :6-107) -- imported as a module from its file;
  * src/model/bone_length_based_backproj.py: optimize_z_offset_by_bones_single (:38-62).  Its module imports
    TensorFlow at the top, so the function's own source lines are cut out with `ast` and executed in a namespace
    that holds `np` and `scipy` (nothing else is referenced by the function).
  * the joint tables: class JointInfo (src/data/datasets.py:52-109) is cut out with `ast` (its module needs packages that
    are absent here) and executed with the reference's own `util.invert_permutation` (src/util.py imports) and stdlib
    equivalents of the two third-party helpers it touches (`more_itertools.pairwise` == `itertools.pairwise`, `AttrDict` ==
    a dict); the joint-name / edge literals are read from make_h36m (src/data/h36m.py:25-31) and make_merged
    (src/data/datasets.py:142-154), the export permutations from export() (src/main.py:119-125), all through `ast`.
    Stored: head names, head edges, mirror mapping, permutation, exported names and re-indexed edges per dataset.
Only inputs and the reference's outputs are stored (data, not source); the reference never travels.  The tests
(tests/test_ref_fixtures.py) pin oracle/metrics.py, oracle/heads.py and the HIP kernels of rows f3/f4 to these vectors.
"""
from __future__ import annotations

import ast
import importlib.util
import os
import sys
import warnings

sys.dont_write_bytecode = True      # /root/reference is read-only for this repo: importing from it must not leave __pycache__ there

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


def load_functions(relpath, names, ns):
    """The named top-level functions of a reference module whose imports (TensorFlow, matplotlib) cannot run here."""
    path = os.path.join(REF, relpath)
    tree = ast.parse(open(path).read())
    fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert sorted(f.name for f in fns) == sorted(names), (relpath, names)
    exec(compile(ast.Module(body=fns, type_ignores=[]), path, 'exec'), ns)
    return ns


def reference_eval_numbers(procrustes_module_fn, rng):
    """build_eval_metrics (main.py:339-359) from the reference's own numpy pieces: tfu3d.root_relative (:23-25),
    util3d.rigid_align / rigid_align_many (:139-171, the body of the py_func, run here with an explicit validity mask),
    eval/analysis.get_pck / get_auc (:8-13).  Only the TF glue between them (tf.norm, reduce_mean_masked) is numpy here."""
    import logging
    import types
    ns = {'np': np, 'logging': logging,
          'eval': types.SimpleNamespace(procrustes=types.SimpleNamespace(procrustes=procrustes_module_fn))}
    load_functions('src/util3d.py', ['rigid_align', 'rigid_align_many'], ns)
    load_functions('src/tfu3d.py', ['root_relative'], ns)
    load_functions('src/eval/analysis.py', ['get_pck', 'get_auc'], ns)
    n, nj = 40, 17
    true = (rng.standard_normal((n, nj, 3)) * 250).astype(np.float32)
    pred = (true + rng.standard_normal((n, nj, 3)) * rng.uniform(20, 160, (n, 1, 1))).astype(np.float32)
    valid = rng.uniform(size=(n, nj)) < 0.8
    valid[:, -1] = True
    valid[:4] = True
    for i in range(n):                                   # >= 4 valid joints per pose (a well-posed alignment)
        valid[i, rng.permutation(nj - 1)[:4]] = True
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        aligned = ns['rigid_align_many'](pred, true, joint_validity_mask=valid, scale_align=True).astype(np.float32)
    dist = np.linalg.norm(ns['root_relative'](pred - true), axis=-1)
    dist_pa = np.linalg.norm(ns['root_relative'](aligned - true), axis=-1)
    rel = dist / np.float32(150)
    per_joint = lambda f: np.array([f(rel[valid[:, j], j]) for j in range(nj)])
    return {'met/true': true, 'met/pred': pred, 'met/valid': valid, 'met/aligned': aligned, 'met/dist': dist, 'met/dist_procrustes': dist_pa,
            'met/mean_error': np.mean(dist[valid]), 'met/mean_error_procrustes': np.mean(dist_pa[valid]),
            'met/mean_pck': ns['get_pck'](rel[valid]), 'met/mean_auc': ns['get_auc'](rel[valid]),
            'met/pck': per_joint(ns['get_pck']), 'met/auc': per_joint(ns['get_auc'])}


def reference_joint_tables():
    """{dataset: dict of arrays} computed by the reference's JointInfo / permute_joints (main.py:119-141)."""
    import itertools
    import types
    sys.path.insert(0, os.path.join(REF, 'src'))
    import util as ref_util                                   # src/util.py: imports without TensorFlow
    ds_src = open(os.path.join(REF, 'src/data/datasets.py')).read()
    ds_tree = ast.parse(ds_src)
    cls = next(n for n in ds_tree.body if isinstance(n, ast.ClassDef) and n.name == 'JointInfo')

    class AttrDict(dict):
        pass
    ns = {'np': np, 'itertools': itertools, 'util': ref_util, 'AttrDict': AttrDict,
          'more_itertools': types.SimpleNamespace(pairwise=itertools.pairwise)}
    exec(compile(ast.Module(body=[cls], type_ignores=[]), 'datasets.py', 'exec'), ns)
    JointInfo = ns['JointInfo']

    def assigned(tree, func, name):
        fn = next(n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == func)
        node = next(n for n in ast.walk(fn) if isinstance(n, ast.Assign) and getattr(n.targets[0], 'id', None) == name)
        return eval(compile(ast.Expression(node.value), func, 'eval'), {})

    h36m_tree = ast.parse(open(os.path.join(REF, 'src/data/h36m.py')).read())
    heads = {'h36m': (assigned(h36m_tree, 'make_h36m', 'joint_names'), assigned(h36m_tree, 'make_h36m', 'edges')),
             'merged': (assigned(ds_tree, 'make_merged', 'joint_names'), assigned(ds_tree, 'make_merged', 'edges'))}
    main_tree = ast.parse(open(os.path.join(REF, 'src/main.py')).read())
    export = next(n for n in ast.walk(main_tree) if isinstance(n, ast.FunctionDef) and n.name == 'export')
    perms = {}
    for node in ast.walk(export):                              # if FLAGS.dataset == '<name>': permutation = [...]
        if isinstance(node, ast.If) and isinstance(node.test, ast.Compare) and isinstance(node.test.comparators[0], ast.Constant):
            for st in node.body:
                if isinstance(st, ast.Assign) and getattr(st.targets[0], 'id', None) == 'permutation':
                    perms[node.test.comparators[0].value] = ast.literal_eval(st.value)
    out = {}
    for ds, (names, edges) in heads.items():
        ji = JointInfo(names, edges)
        pj = ji.permute_joints(perms[ds])
        out[ds] = {'head_names': np.array([n.encode() for n in ji.names]),
                   'head_edges': np.array([tuple(e) for e in ji.stick_figure_edges], np.int64),
                   'mirror': np.array(ji.mirror_mapping, np.int64), 'permutation': np.array(perms[ds], np.int64),
                   'out_names': np.array([n.encode() for n in pj.names]),
                   'out_edges': np.array([tuple(int(x) for x in e) for e in pj.stick_figure_edges], np.int64)}
    return out


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
    out.update(reference_eval_numbers(procrustes, rng))
    for ds, tab in reference_joint_tables().items():
        for k, v in tab.items():
            out[f'joints/{ds}/{k}'] = v
    out['versions'] = np.array([np.__version__, scipy.__version__])
    np.savez_compressed(OUT, **out)
    print(f'wrote {OUT} ({os.path.getsize(OUT) / 1024:.1f} KiB); numpy {np.__version__}, scipy {scipy.__version__}')


if __name__ == '__main__':
    main()
