"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): the reference's alternative decode heads, SURVEY.md
section 8 row f3, restated with NumPy in the reference's own dtypes (fp32 tensors, fp64 inside scipy).

  * `coords01` is the output of `net_output_to_heatmap_and_coords` (reference src/model/volumetric.py:227-235):
    soft-argmax coordinates in [0,1], head joint order, (x, y, z).
  * `heatmap_to_image`                      volumetric.py:288-295
  * `backproject_bone_lengths`              volumetric.py:171-191 (`bone-lengths`, `bone-lengths-true`) with
    `optimize_z_offset_by_bones_single`     bone_length_based_backproj.py:38-62 (scipy LM, called verbatim)
  * `backproject_root_depth`                volumetric.py:192-199 (`true-root-depth`)
  * `back_project`                          volumetric.py:284-285
  * `to_orig_cam`                           volumetric.py:277-281 (mirror joints when det(R) <= 0)
  * `root_relative`                         tfu3d.py:23-25

PARITY UNPINNED against TensorFlow (the reference cannot run here); the scipy call is the reference's own.
The 3x3 einsum `Bij,BCj->BCi` is evaluated as ((k0*u + k1*v) + k2*1) in fp32 (TF's summation order inside
einsum is not specified by the reference)."""
from __future__ import annotations

import numpy as np
import scipy.optimize


def heatmap_to_image(coords01_xy: np.ndarray, stride: int, proc_side: int = 256, centered: bool = True) -> np.ndarray:
    last = proc_side - 1
    lrc = last - (last % stride) - 1
    out = coords01_xy.astype(np.float32) * np.float32(lrc)
    if centered:
        out = out + np.float32(stride // 2)
    return out.astype(np.float32)


def heatmap_to_25d(coords01, stride, proc_side=256, centered=True, box_size_mm=2200.0):
    """volumetric.py:298-300."""
    c = np.asarray(coords01, np.float32)
    return np.concatenate([heatmap_to_image(c[..., :2], stride, proc_side, centered),
                           c[..., 2:] * np.float32(box_size_mm)], axis=-1).astype(np.float32)


def camcoords_and_delta_z(coords01, inv_intrinsics, stride, proc_side=256, centered=True, box_size_mm=2200.0):
    c = np.asarray(coords01, np.float32)
    k = np.asarray(inv_intrinsics, np.float32)
    uv = heatmap_to_image(c[..., :2], stride, proc_side, centered)
    u, v = uv[..., 0], uv[..., 1]
    one = np.float32(1.0)
    cam = np.stack([(k[:, None, i, 0] * u + k[:, None, i, 1] * v) + k[:, None, i, 2] * one for i in range(3)], axis=-1)
    delta_z = (c[..., 2] - c[:, -1:, 2]) * np.float32(box_size_mm)
    return cam.astype(np.float32), delta_z.astype(np.float32)


def optimize_z_offset_by_bones_single(x, delta_z, target_bone_lengths, edges, initial_guess=2000):
    """bone_length_based_backproj.py:38-62, line by line (x, delta_z fp32 as TF hands them to the py_func)."""
    a = np.asarray([x[i] - x[j] for i, j in edges])
    y = x * np.expand_dims(delta_z, -1)
    b = np.asarray([y[i] - y[j] for i, j in edges])
    c = np.sum(a ** 2, axis=1)
    d = np.sum(2 * a * b, axis=1)
    e = np.sum(b ** 2, axis=1)

    def reconstruct_bone_lengths(z):
        return np.sqrt(z ** 2 * c + z * d + e)

    def fn(z):
        return reconstruct_bone_lengths(z) - target_bone_lengths

    def jacobian(z):
        return ((z * c + d) / reconstruct_bone_lengths(z)).reshape([-1, 1])

    solution = scipy.optimize.least_squares(fn, jac=jacobian, x0=initial_guess, method='lm')
    return float(solution.x[0])


def back_project(camcoords2d_homog, delta_z, z_offset):
    return (camcoords2d_homog * np.expand_dims(delta_z + np.expand_dims(z_offset, -1), -1)).astype(np.float32)


def root_relative(coords):
    return coords - coords[:, -1:]


def backproject_bone_lengths(coords01, inv_intrinsics, target_bone_lengths, edges, stride, proc_side=256,
                             centered=True, box_size_mm=2200.0):
    """target_bone_lengths: [E] (dataset means, `bone-lengths`) or [N,E] (`bone-lengths-true`).
    Returns (coords3d_pred [N,J,3] fp32, z_offset [N] fp32)."""
    cam, dz = camcoords_and_delta_z(coords01, inv_intrinsics, stride, proc_side, centered, box_size_mm)
    t = np.asarray(target_bone_lengths, np.float64)
    z = np.array([optimize_z_offset_by_bones_single(cam[i], dz[i], t if t.ndim == 1 else t[i], edges)
                  for i in range(cam.shape[0])], dtype=np.float32)
    return back_project(cam, dz, z), z


def backproject_root_depth(coords01, inv_intrinsics, root_z, stride, proc_side=256, centered=True, box_size_mm=2200.0):
    cam, dz = camcoords_and_delta_z(coords01, inv_intrinsics, stride, proc_side, centered, box_size_mm)
    return back_project(cam, dz, np.asarray(root_z, np.float32))


def to_orig_cam(x, rot, mirror_mapping):
    x = np.asarray(x, np.float32)
    rot = np.asarray(rot, np.float32)
    y = np.stack([(rot[:, None, i, 0] * x[..., 0] + rot[:, None, i, 1] * x[..., 1]) + rot[:, None, i, 2] * x[..., 2]
                  for i in range(3)], axis=-1).astype(np.float32)
    det = np.linalg.det(rot.astype(np.float64))
    return np.where((det > 0)[:, None, None], y, y[:, list(mirror_mapping)])
