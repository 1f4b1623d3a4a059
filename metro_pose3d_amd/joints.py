"""Skeleton tables baked into an exported model: `joint_names` and `joint_edges` outputs.

Reference: head joint order and stick-figure edges from src/data/h36m.py:25-31 and
src/data/datasets.py:142-154; the export-time permutation from src/main.py:119-125; the
re-indexing of names/edges from JointInfo.permute_joints (src/data/datasets.py:104-108).
"""
from __future__ import annotations

import dataclasses
from typing import List, Tuple

import numpy as np

_H36M_HEAD = ['rhip', 'rkne', 'rank', 'lhip', 'lkne', 'lank', 'tors', 'neck', 'head', 'htop',
              'lsho', 'lelb', 'lwri', 'rsho', 'relb', 'rwri', 'pelv']
_H36M_CHAINS = [['htop', 'head', 'neck', 'lsho', 'lelb', 'lwri'], ['neck', 'rsho', 'relb', 'rwri'],
                ['neck', 'tors', 'pelv', 'lhip', 'lkne', 'lank'], ['pelv', 'rhip', 'rkne', 'rank']]

_MERGED_HEAD = (
    'neck nose lsho lelb lwri lhip lkne lank rsho relb rwri rhip rkne rank leye lear reye rear pelv '
    'htop_tdhp neck_tdhp rsho_tdhp lsho_tdhp rhip_tdhp lhip_tdhp spin_tdhp head_tdhp pelv_tdhp '
    'rhip_h36m lhip_h36m tors_h36m neck_h36m head_h36m htop_h36m lsho_h36m rsho_h36m pelv_h36m '
    'lhip_tdpw rhip_tdpw bell_tdpw che1_tdpw che2_tdpw ltoe_tdpw rtoe_tdpw neck_tdpw lcla_tdpw '
    'rcla_tdpw head_tdpw lsho_tdpw rsho_tdpw lhan_tdpw rhan_tdpw pelv_tdpw').split()
_MERGED_EDGES = [(1, 0), (0, 18), (0, 2), (2, 3), (3, 4), (0, 8), (8, 9), (9, 10), (18, 5), (5, 6),
                 (6, 7), (18, 11), (11, 12), (12, 13), (15, 14), (14, 1), (17, 16), (16, 1)]

_PERM_H36M = [16] + list(range(16))
_PERM_MERGED = [0, 1, 18] + list(range(2, 18))


@dataclasses.dataclass(frozen=True)
class Skeleton:
    head_names: Tuple[str, ...]      # order of the head's joints (root = last, tfu3d.py:23-25)
    permutation: Tuple[int, ...]     # output row i = head joint permutation[i]
    names: Tuple[str, ...]           # `joint_names` output
    edges: Tuple[Tuple[int, int], ...]  # `joint_edges` output (indices into `names`)
    head_edges: Tuple[Tuple[int, int], ...] = ()   # stick-figure edges in head order (bone-length heads)

    @property
    def head_mirror(self) -> Tuple[int, ...]:
        """index of the opposite-side joint, head order (reference datasets.py:77-80,94-100)."""
        idx = {n: i for i, n in enumerate(self.head_names)}
        other = lambda n: ('r' + n[1:]) if n.startswith('l') else ('l' + n[1:]) if n.startswith('r') else n
        return tuple(idx[other(n)] for n in self.head_names)

    @property
    def n_head(self) -> int:
        return len(self.head_names)

    @property
    def n_out(self) -> int:
        return len(self.names)

    def edges_array(self) -> np.ndarray:
        return np.asarray(self.edges, dtype=np.int64).reshape(-1, 2)   # int64 like main.py:141

    def names_bytes(self) -> List[bytes]:
        return [n.encode() for n in self.names]                        # TF string tensor -> bytes


def _make(head: List[str], head_edges: List[Tuple[int, int]], perm: List[int]) -> Skeleton:
    position_of = {h: i for i, h in enumerate(perm)}   # head index -> output row
    names = tuple(head[h] for h in perm)
    edges = tuple((position_of[a], position_of[b]) for a, b in head_edges)
    return Skeleton(tuple(head), tuple(perm), names, edges, tuple((int(a), int(b)) for a, b in head_edges))


def skeleton(dataset: str) -> Skeleton:
    if dataset == 'h36m':
        idx = {n: i for i, n in enumerate(_H36M_HEAD)}
        e = [(idx[a], idx[b]) for chain in _H36M_CHAINS for a, b in zip(chain, chain[1:])]
        return _make(_H36M_HEAD, e, _PERM_H36M)
    if dataset == 'merged':
        return _make(_MERGED_HEAD, _MERGED_EDGES, _PERM_MERGED)
    if dataset == 'many19':
        # 19-joint head (README.md:27-28 "19 joints"; BASELINE.json configs 3-4)
        return _make(_MERGED_HEAD[:19], _MERGED_EDGES, _PERM_MERGED)
    raise ValueError(f'unknown dataset {dataset!r} (h36m | merged | many19)')
