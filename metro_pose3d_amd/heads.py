"""Alternative decode heads of the reference, the step AFTER the hot path, on the GPU
through the C ABI (csrc/heads.hip).  Names and argument meaning follow the reference:

  coords01_from_logits      net_output_to_heatmap_and_coords          src/model/volumetric.py:227-235
  backproject_bone_lengths  scale_recovery 'bone-lengths' / '-true'   volumetric.py:171-191,
                            optimize_z_offset_by_bones(_tensor)       src/model/bone_length_based_backproj.py:15-62
  backproject_root_depth    scale_recovery 'true-root-depth'          volumetric.py:192-199
  heatmap_to_25d            crop pixels + z * box_size                volumetric.py:298-300
  to_orig_cam               rotation + mirror on det(R) <= 0          volumetric.py:277-281

MeTRo's own output (`scale_recovery == 'metro'`, volumetric.py:200-201) is `Engine.forward` / `estimate_pose`.
There is no CPU fallback: without the HIP library these raise MetroError."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from metro_pose3d_amd import _lib
from metro_pose3d_amd._lib import check
from metro_pose3d_amd.spec import ModelSpec


def _p(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream(dev) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _f32(x, dev, shape_tail) -> torch.Tensor:
    t = torch.as_tensor(x, dtype=torch.float32, device=dev).contiguous()
    if tuple(t.shape[1:]) != tuple(shape_tail):
        raise ValueError(f'expected [N,{",".join(map(str, shape_tail))}], got {tuple(t.shape)}')
    return t


def coords01_from_logits(logits: torch.Tensor, spec: ModelSpec, precise: int = 1) -> torch.Tensor:
    """fp32 (precise 0/1) or fp64 (precise 2) NHWC logits [N,S,S,D*J] -> soft-argmax coords in [0,1] [N,J,3]."""
    lib = _lib.load()
    if logits.dim() != 4 or logits.shape[1] != spec.heatmap_side or logits.shape[3] != spec.n_head_channels:
        raise ValueError(f'logits must be [N,{spec.heatmap_side},{spec.heatmap_side},{spec.n_head_channels}]')
    logits = logits.to(torch.float64 if precise == 2 else torch.float32).contiguous()
    n = logits.shape[0]
    cs = spec.to_c(int(precise))
    scratch = torch.empty(lib.metro_softargmax_scratch_bytes(n, spec.heatmap_side, spec.skeleton.n_head),
                          dtype=torch.uint8, device=logits.device)
    out = torch.empty((n, spec.skeleton.n_head, 3), dtype=torch.float32, device=logits.device)
    check(lib.metro_softargmax01(_p(logits), n, C.byref(cs), int(precise), _p(scratch), _p(out), _stream(logits.device)),
          'metro_softargmax01')
    return out


def backproject_bone_lengths(coords01: torch.Tensor, inv_intrinsics, bone_lengths, spec: ModelSpec,
                             edges: Optional[Sequence[Tuple[int, int]]] = None, root_relative: bool = False,
                             permute: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """coords01 [N,J,3] (head order) + inv_intrinsics [N,3,3] + target bone lengths [E] (dataset means) or [N,E]
    (per pose) -> (coords3d_pred [N,J or Jout,3] mm in camera space, z_offset [N])."""
    lib = _lib.load()
    dev = coords01.device
    c = _f32(coords01, dev, (spec.skeleton.n_head, 3))
    n = c.shape[0]
    k = _f32(torch.as_tensor(inv_intrinsics).reshape(n, 9), dev, (9,))
    e = np.asarray(spec.skeleton.head_edges if edges is None else edges, dtype=np.int32).reshape(-1, 2)
    if e.size == 0 or e.min() < 0 or e.max() >= spec.skeleton.n_head:
        raise ValueError('edges must index head joints')
    te = torch.from_numpy(np.ascontiguousarray(e)).to(dev)
    t = torch.as_tensor(np.asarray(bone_lengths, dtype=np.float64), device=dev).contiguous()
    if t.shape not in ((len(e),), (n, len(e))):
        raise ValueError(f'bone_lengths must be [{len(e)}] or [{n},{len(e)}], got {tuple(t.shape)}')
    cs = spec.to_c(1)
    out = torch.empty((n, spec.skeleton.n_out if permute else spec.skeleton.n_head, 3), dtype=torch.float32, device=dev)
    z = torch.empty((n,), dtype=torch.float32, device=dev)
    check(lib.metro_backproject_bone_lengths(_p(c), _p(k), _p(t), int(t.dim() == 2), _p(te), len(e), n, C.byref(cs),
                                             int(root_relative), int(permute), _p(out), _p(z), _stream(dev)),
          'metro_backproject_bone_lengths')
    return out, z


def backproject_root_depth(coords01: torch.Tensor, inv_intrinsics, root_z, spec: ModelSpec,
                           root_relative: bool = False, permute: bool = False) -> torch.Tensor:
    lib = _lib.load()
    dev = coords01.device
    c = _f32(coords01, dev, (spec.skeleton.n_head, 3))
    n = c.shape[0]
    k = _f32(torch.as_tensor(inv_intrinsics).reshape(n, 9), dev, (9,))
    rz = torch.as_tensor(root_z, dtype=torch.float32, device=dev).contiguous().reshape(n)
    cs = spec.to_c(1)
    out = torch.empty((n, spec.skeleton.n_out if permute else spec.skeleton.n_head, 3), dtype=torch.float32, device=dev)
    check(lib.metro_backproject_root_depth(_p(c), _p(k), _p(rz), n, C.byref(cs), int(root_relative), int(permute), _p(out),
                                           _stream(dev)), 'metro_backproject_root_depth')
    return out


def heatmap_to_25d(coords01: torch.Tensor, spec: ModelSpec) -> torch.Tensor:
    """heatmap_to_25d (volumetric.py:298-300): [N,J,3] in [0,1] -> (x px, y px, z mm), head order."""
    lib = _lib.load()
    dev = coords01.device
    c = _f32(coords01, dev, (spec.skeleton.n_head, 3))
    cs = spec.to_c(1)
    out = torch.empty_like(c)
    check(lib.metro_heatmap_to_25d(_p(c), c.shape[0], C.byref(cs), _p(out), _stream(dev)), 'metro_heatmap_to_25d')
    return out


def to_orig_cam(coords: torch.Tensor, rot_to_orig_cam, mirror_mapping: Sequence[int]) -> torch.Tensor:
    lib = _lib.load()
    dev = coords.device
    x = torch.as_tensor(coords, dtype=torch.float32, device=dev).contiguous()
    n, nj = x.shape[0], x.shape[1]
    r = _f32(torch.as_tensor(rot_to_orig_cam).reshape(n, 9), dev, (9,))
    m = np.asarray(mirror_mapping, dtype=np.int32)
    if m.shape != (nj,) or m.min() < 0 or m.max() >= nj:
        raise ValueError(f'mirror_mapping must be a permutation-like int array of length {nj}')
    tm = torch.from_numpy(m).to(dev)
    out = torch.empty_like(x)
    check(lib.metro_to_orig_cam(_p(x), _p(r), _p(tm), _p(out), n, nj, _stream(dev)), 'metro_to_orig_cam')
    return out
