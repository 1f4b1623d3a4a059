"""Batch sharding of the hot path over the GPUs of one node (one process per GPU).

No op of the path mixes images (BatchNorm is inference-mode, reference architectures.py:32;
softmax is per (image, joint), volumetric.py:233), so a batch shards by image with replicated
weights and NO activation exchange.  The only collective is one all-gather of the pose outputs
([N/G, Jout, 3] fp32, <= 15 KB per rank: latency-bound), RCCL over xGMI when the process group
is `nccl`, gloo in the CPU tests.
Bits: a shard has the bits of the single-GPU run as long as both run the same kernel instantiations.  In the `f64` parity
mode that is always the case (one kernel configuration whatever the batch: tests/test_gpu_forward.py,
test_f64_mode_is_shard_invariant).  In the `f16` throughput mode tile shapes follow the crops per call (thresholds: 128 crops
at stride 16, 32 at stride 8, 8 at stride 4 for the head, 128 / 32 / 16 for the 3x3 layers -- see inference.estimate_pose), so a shard below a threshold and a full batch above
it differ by fp16 rounding flips; below the thresholds the bits are the same.
A failure on ONE rank -- fp16 overflow on its shard, or an exception in its model load / plan build / upload / forward -- is
made collective: the rank still joins the gather and its status row tells every rank to raise (all_gather_poses_with_status),
so nobody is left blocking in the collective.  Not covered: a rank that cannot allocate a device tensor any more (sticky HIP
error under `nccl`) -- the process group's timeout ends the others (inference.estimate_pose lists the cases).
(The reference has no multi-GPU code at all; this is defined by BASELINE.json's north star.)
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous chunk [begin, end) of rank `rank`; the first n % world ranks get one extra."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f'bad rank/world {rank}/{world}')
    q, r = divmod(n, world)
    begin = rank * q + min(rank, r)
    return begin, begin + q + (1 if rank < r else 0)


def all_gather_poses(local: torch.Tensor, n_total: int, group: Optional[dist.ProcessGroup] = None,
                     out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Gathers the per-rank [n_r, J, 3] outputs (contiguous shards, rank order) into [n_total, J, 3]."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    j, c = local.shape[1], local.shape[2]
    if out is None:
        out = torch.empty((n_total, j, c), dtype=local.dtype, device=local.device)
    if n_total % world == 0:
        if local.shape[0] != n_total // world:
            raise ValueError(f'rank {rank}: local batch {local.shape[0]} != {n_total // world}')
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)   # one ncclAllGather
        return out
    # ragged tail: pad every shard to the largest, gather, then compact
    q = -(-n_total // world)
    padded = torch.zeros((q, j, c), dtype=local.dtype, device=local.device)
    padded[:local.shape[0]] = local
    buf = torch.empty((world * q, j, c), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, padded, group=group)
    for r in range(world):
        b, e = shard_range(n_total, r, world)
        out[b:e] = buf[r * q: r * q + (e - b)]
    return out


def all_gather_poses_with_status(local: torch.Tensor, n_total: int, status: int,
                                 group: Optional[dist.ProcessGroup] = None) -> Tuple[torch.Tensor, list]:
    """all_gather_poses with one extra row per rank that carries an integer status (0 = fine): STILL one collective.  Every rank
    must call it, also the ones whose forward failed (their `local` may hold garbage): returns (poses [n_total, J, 3], the
    status of every rank) so that all ranks can raise together instead of leaving the healthy ones blocked in the gather."""
    world = dist.get_world_size(group)
    j, c = local.shape[1], local.shape[2]
    # the status rides as a float in element [q, 0, 0]: exact only in fp32 and only for |status| < 2**24 (saturated below)
    if local.dtype != torch.float32 or j < 1 or c < 1:
        raise ValueError(f'all_gather_poses_with_status: poses must be float32 [n, J >= 1, 3], got {local.dtype} {tuple(local.shape)}')
    status = max(-1, min(int(status), (1 << 24) - 1))
    q = -(-n_total // world)
    padded = torch.zeros((q + 1, j, c), dtype=local.dtype, device=local.device)
    padded[:local.shape[0]] = local
    padded[q, 0, 0] = float(status)
    buf = torch.empty((world * (q + 1), j, c), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, padded, group=group)
    buf = buf.view(world, q + 1, j, c)
    statuses = [int(v) for v in buf[:, q, 0, 0].round().to(torch.int64).cpu().tolist()]
    out = torch.empty((n_total, j, c), dtype=local.dtype, device=local.device)
    for r in range(world):
        b, e = shard_range(n_total, r, world)
        out[b:e] = buf[r, :e - b]
    return out, statuses


def sharded_forward(forward_fn: Callable[[torch.Tensor], torch.Tensor], images: torch.Tensor,
                    group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """Every rank holds the same global batch `images`; rank r runs `forward_fn` on its contiguous
    shard and all ranks return the full [N, J, 3] result."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    b, e = shard_range(images.shape[0], rank, world)
    if e > b:
        local = forward_fn(images[b:e])
    else:                                   # more ranks than images: learn [J, 3] from one image
        local = forward_fn(images[:1])[:0]
    return all_gather_poses(local, images.shape[0], group)


class OverlappedPoseGather:
    """All-gather of per-rank pose outputs that overlaps with the NEXT batch's forward.

    The collective is latency-bound (<= 15 KB per rank), so it is issued asynchronously (RCCL runs
    it on its own stream) on double-buffered outputs: `submit(i, local)` starts the gather of step i
    and only makes the caller's stream wait for the gather of step i-2, whose buffers it is about
    to reuse.  `result(i)` / `finish()` wait for what is still in flight."""

    def __init__(self, n_local: int, n_joints: int, world: int, device, group=None, depth: int = 2):
        self.group = group
        self.depth = depth
        self.local = [torch.empty((n_local, n_joints, 3), dtype=torch.float32, device=device) for _ in range(depth)]
        self.gathered = [torch.empty((n_local * world, n_joints, 3), dtype=torch.float32, device=device)
                         for _ in range(depth)]
        self.work = [None] * depth

    def local_buffer(self, step: int) -> torch.Tensor:
        """Output buffer for the forward of `step`; waits (stream-side) for the gather that last used it."""
        k = step % self.depth
        if self.work[k] is not None:
            self.work[k].wait()
            self.work[k] = None
        return self.local[k]

    def submit(self, step: int) -> None:
        k = step % self.depth
        self.work[k] = dist.all_gather_into_tensor(self.gathered[k], self.local[k], group=self.group,
                                                   async_op=True)

    def result(self, step: int) -> torch.Tensor:
        k = step % self.depth
        if self.work[k] is not None:
            self.work[k].wait()
            self.work[k] = None
        return self.gathered[k]

    def finish(self) -> None:
        for k in range(self.depth):
            if self.work[k] is not None:
                self.work[k].wait()
                self.work[k] = None
