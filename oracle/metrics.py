"""CPU restatement of the reference's evaluation metrics (row f4).  TEST INFRASTRUCTURE.

Follows src/main.py:339-359 (build_eval_metrics), src/tfu3d.py:23-38, src/util3d.py:139-171
(rigid_align[_many]) and src/eval/procrustes.py:6-107 (procrustes, scaling=True, reflection=False),
with NumPy's LAPACK SVD where the reference uses it.  PARITY: the reference has no test vectors for
these functions either; pinned by known answers in tests/test_metrics.py (rigid motions and scalings
of the ground truth give zero aligned error; a mirrored pose does not).
"""
from __future__ import annotations

import numpy as np


def procrustes_no_reflection(x: np.ndarray, y: np.ndarray):
    """procrustes(X=true_valid, Y=pred_valid, scaling=True, reflection=False) -> (T, b, c)."""
    mux, muy = x.mean(0), y.mean(0)
    x0, y0 = x - mux, y - muy
    normx, normy = np.sqrt((x0 ** 2).sum()), np.sqrt((y0 ** 2).sum())
    x0, y0 = x0 / normx, y0 / normy
    a = x0.T @ y0                                            # procrustes.py:61
    u, s, vt = np.linalg.svd(a, full_matrices=False)
    v = vt.T
    t = v @ u.T
    if np.linalg.det(t) < 0:                                 # reflection=False: force a proper rotation (:66-75)
        v[:, -1] *= -1
        s[-1] *= -1
        t = v @ u.T
    trace = s.sum()
    b = trace * normx / normy                                # :82
    c = mux - b * muy @ t                                    # :101
    return t, b, c


def eval_metrics(pred, true, valid=None, threshold=np.float32(150)):
    pred = np.asarray(pred, np.float32)
    true = np.asarray(true, np.float32)
    n, nj, _ = pred.shape
    valid = np.ones((n, nj), bool) if valid is None else np.asarray(valid, bool)
    rr = lambda d: d - d[:, -1:, :]                          # tfu3d.root_relative
    dist = np.linalg.norm(rr(pred.astype(np.float64) - true), axis=-1)
    aligned = np.empty_like(pred)
    for i in range(n):                                       # util3d.rigid_align_many
        try:
            t, b, c = procrustes_no_reflection(true[i][valid[i]].astype(np.float64), pred[i][valid[i]].astype(np.float64))
            aligned[i] = (b * pred[i].astype(np.float64) @ t + c).astype(np.float32)   # py_func returns float32
        except np.linalg.LinAlgError:                        # util3d.py:152-154: keep the prediction
            aligned[i] = pred[i]
    dist_pa = np.linalg.norm(rr(aligned.astype(np.float64) - true), axis=-1)
    d32 = dist.astype(np.float32)
    auc_score = np.maximum(np.float32(0), 1 - d32 / threshold)
    correct = (d32 <= threshold).astype(np.float32)
    cnt = valid.sum(0)
    m = lambda a: (a * valid).sum() / valid.sum()
    mj = lambda a: (a * valid).sum(0) / cnt
    return {'mean_error': m(dist), 'mean_error_procrustes': m(dist_pa), 'auc': mj(auc_score), 'mean_auc': m(auc_score),
            'pck': mj(correct), 'mean_pck': m(correct), 'dist': dist, 'dist_procrustes': dist_pa}
