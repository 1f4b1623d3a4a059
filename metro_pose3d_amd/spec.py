"""ModelSpec: the flag values a frozen graph of the reference bakes in, made explicit.

Reference defaults: src/options.py:41 (proc_side), :73 (dtype), :96 (stride_test), :109-111
(architecture), :113 (depth), :118 (centered_stride), :119 (box_size_mm); dataset -> joints via
src/data/datasets.py and src/main.py:119-125 (see joints.py).
"""
from __future__ import annotations

import dataclasses
import json

from metro_pose3d_amd import _lib
from metro_pose3d_amd.joints import Skeleton, skeleton


@dataclasses.dataclass(frozen=True)
class ModelSpec:
    arch: int = 50
    stride: int = 16
    dataset: str = 'h36m'
    depth: int = 8
    centered_stride: bool = True
    proc_side: int = 256
    box_size_mm: float = 2200.0
    base_width: int = 64

    def __post_init__(self):
        if self.arch not in (50, 101):
            raise ValueError(f'arch must be 50 or 101, got {self.arch}')
        if self.stride not in (4, 8, 16, 32):
            raise ValueError(f'stride must be one of 4, 8, 16, 32, got {self.stride}')
        skeleton(self.dataset)   # validates the dataset name

    @property
    def skeleton(self) -> Skeleton:
        return skeleton(self.dataset)

    @property
    def arch_name(self) -> str:
        return f'resnet_v2_{self.arch}'

    @property
    def n_head_channels(self) -> int:
        return self.depth * self.skeleton.n_head

    @property
    def heatmap_side(self) -> int:
        return self.proc_side // self.stride

    def to_c(self, precision: int) -> _lib.MetroSpec:
        sk = self.skeleton
        cs = _lib.MetroSpec()
        cs.arch = self.arch
        cs.stride = self.stride
        cs.n_joints_head = sk.n_head
        cs.depth = self.depth
        cs.centered_stride = int(self.centered_stride)
        cs.proc_side = self.proc_side
        cs.box_size_mm = float(self.box_size_mm)
        cs.base_width = self.base_width
        cs.precision = precision
        cs.n_joints_out = sk.n_out
        for i, p in enumerate(sk.permutation):
            cs.permutation[i] = p
        return cs

    def to_json(self) -> str:
        return json.dumps(dataclasses.asdict(self), sort_keys=True)

    @staticmethod
    def from_json(s: str) -> 'ModelSpec':
        return ModelSpec(**json.loads(s))
