"""Drop-in for the reference's `inference.py` (reference inference.py:11-43).

Same call: `estimate_pose(images_tensor, model_path) -> (poses, joint_edges, joint_names)`,
same CLI flag `--model-path`.  Differences that follow from not being TensorFlow:
  * eager: `poses` is a float32 torch.Tensor [N, Jout, 3] (mm, root-relative) on the GPU, already
    computed (the reference returns graph tensors to `sess.run` later, inference.py:25-27);
  * `joint_edges` is an int64 ndarray [E, 2] and `joint_names` an object ndarray of `bytes`,
    which is what `sess.run` yields for those constants (main.py:140-141);
  * `model_path` is a frozen `.pb` GraphDef like the reference's (read without TensorFlow, tfgraph.py) or the
    `.npz` container of modelfile.py;
  * N is free, as with the reference's `[None, 256, 256, 3]` placeholder (main.py:109-111): the call plans for the
    batch it is given -- one `metro_forward` for up to 256 crops, 256-crop chunks beyond.  Kernel dispatch depends on the
    crops per call (tile counts against 256 CUs: from 128 crops on the 3x3 layers take 512-pixel tiles, the head takes
    128- / 256-pixel tiles once it has 256 of them), so results are not bit-stable ACROSS call sizes >= 128 at stride 16
    (estimate_pose docstring);
  * one process per GPU under torch.distributed: the same call shards the batch by image and all-gathers the poses.
Input contract (inference.py:17-18, main.py:109-110): float32 NHWC [N,256,256,3], RGB in [0,1].
The arithmetic mode defaults to fp16, the reference's default compute dtype (options.py:73);
pass precision='f64' (or METRO_PRECISION=f64) for the parity mode (fp64 arithmetic inside).
"""
from __future__ import annotations

import argparse
import contextlib
import os
from collections import OrderedDict
from typing import Optional, Tuple

import numpy as np
import torch

from metro_pose3d_amd import _lib
from metro_pose3d_amd.engine import Engine
from metro_pose3d_amd.modelfile import load_model

# One plan (workspace sized for its max_batch) per (model file, precision, device, batch bucket), least recently used first.
# Workspace at RN50-s16: 10 MB per crop (2.6 GB at 256 of the 288 GB); the parameter blob (48 MB) is per engine.
BATCH_BUCKETS = (8, 64, 256)
MAX_CACHED_ENGINES = 6
_ENGINES: 'OrderedDict[Tuple[str, float, str, int, int], Engine]' = OrderedDict()


def batch_bucket(n: int) -> int:
    """max_batch of the engine a call with n crops runs on (calls with more than 256 crops are chunked by 256)."""
    for b in BATCH_BUCKETS:
        if n <= b:
            return b
    return BATCH_BUCKETS[-1]


def _engine_for(model_path: str, precision: str, device: torch.device, n: int = 64) -> Engine:
    path = os.path.abspath(model_path)
    key = (path, os.path.getmtime(path), precision, device.index or 0, batch_bucket(n))
    eng = _ENGINES.get(key)
    if eng is None:
        spec, params = load_model(path)
        eng = Engine(spec, params, precision=precision, max_batch=key[-1], device=device)
        _ENGINES[key] = eng
        while len(_ENGINES) > MAX_CACHED_ENGINES:
            _, old = _ENGINES.popitem(last=False)
            old.close()
    else:
        _ENGINES.move_to_end(key)
    return eng


class _ShapeError(ValueError):
    """Wrong image shape: raised identically on every rank of a sharded call, before any collective."""


def clear_cache() -> None:
    """Closes every cached engine (plan, parameter blob, workspace: 2.6 GB at bucket 256) after draining its device."""
    while _ENGINES:
        _, eng = _ENGINES.popitem(last=False)
        eng.close()


def _resolve_device(images_tensor: torch.Tensor) -> torch.device:
    if not torch.cuda.is_available():
        raise _lib.MetroError('no HIP device visible: the MeTRo hot path has no CPU fallback')
    return images_tensor.device if images_tensor.is_cuda else torch.device('cuda', torch.cuda.current_device())


def _gather_shards(local: torch.Tensor, n_total: int, group, status: int = 0):
    """The one collective of the path: all-gather of the per-rank [n_r, Jout, 3] poses with a status row per rank
    (dist.all_gather_poses_with_status).  RCCL moves device tensors (backend `nccl`); any other backend (gloo in the tests) gets
    host tensors and the result goes back.  Returns (poses, status of every rank)."""
    import torch.distributed as dist
    from metro_pose3d_amd.dist import all_gather_poses_with_status
    if dist.get_backend(group) == 'nccl' or not local.is_cuda:
        return all_gather_poses_with_status(local, n_total, status, group)
    out, st = all_gather_poses_with_status(local.cpu(), n_total, status, group)
    return out.to(local.device), st


def estimate_pose(images_tensor, model_path, precision: Optional[str] = None, check_finite: Optional[bool] = None,
                  shard: Optional[bool] = None, group=None):
    """images [N,256,256,3] float32 in [0,1] -> (poses [N,Jout,3] mm, joint_edges, joint_names).

    Multi-GPU (BASELINE.json north star; the reference's call has no such notion): when `torch.distributed` is initialised with
    more than one rank (one process per GPU), EVERY rank makes this same call with the same N images; rank r computes the
    contiguous shard dist.shard_range(N, r, world) on its own GPU and all ranks return the full [N,Jout,3] after ONE all-gather
    of the poses (RCCL over xGMI under backend `nccl`).  No activation crosses ranks, so the result has the bits of the
    single-GPU call as long as both run the same kernel instantiations: at stride 16 always below 128 crops per call (see below).
    `shard=False` keeps the call local; `group` selects a process group.

    Results are NOT bit-stable across call sizes: kernel tile shapes follow the crops per call (the engine buckets of 8 / 64 /
    256; the 3x3 layers take 512-pixel tiles once a layer has 256 of them -- from 128 crops per call at stride 16, 32 at stride 8,
    16 at stride 4; the head takes 128- and then 256-pixel tiles once it has 256 of them: n * S * S / 128 >= 256, i.e. from 128
    crops at stride 16, 32 at stride 8, 8 at stride 4) and every tile shape is another fp32 summation order.  Differences are
    rounding flips of the fp16 chain (tests/test_gpu_forward.py); calls below those sizes agree bit for bit with one another
    whatever their size.

    `check_finite` (default on; METRO_CHECK_FINITE=0 turns it off): the finalize launch's non-finite screen is folded on the
    device after every forward of the call and read back ONCE (one stream synchronisation per call, as the reference's blocking
    sess.run); NonFiniteError is raised when activations overflowed -- a checkpoint whose residual stream exceeds fp16's 65 504
    needs precision='f32m' (or 'f64').  Sharded calls fail COLLECTIVELY: the flag travels in the pose all-gather (one status
    row per rank), so an overflow or an exception on one rank raises on every rank instead of leaving the others blocked in
    the collective.  The screen reads the words of the immediately preceding forward on the engine's workspace and stream: one
    caller per cached engine at a time (an Engine is not thread-safe, like the C plan it wraps)."""
    if precision is None:
        precision = os.environ.get('METRO_PRECISION', 'f16')
    if check_finite is None:
        check_finite = os.environ.get('METRO_CHECK_FINITE', '1') != '0'
    if isinstance(images_tensor, np.ndarray):
        images_tensor = torch.from_numpy(images_tensor)
    if not isinstance(images_tensor, torch.Tensor):
        raise ValueError(f'images must be a torch.Tensor or numpy array, got {type(images_tensor)}')
    if images_tensor.dtype != torch.float32:
        raise ValueError(f'images must be float32 in [0,1] (reference inference.py:18), got {images_tensor.dtype}')
    device = _resolve_device(images_tensor)
    n = int(images_tensor.shape[0]) if images_tensor.dim() == 4 else 0
    rank, world = 0, 1
    if shard is not False:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            rank, world = dist.get_rank(group), dist.get_world_size(group)
        elif shard:
            raise _lib.MetroError('shard=True needs an initialised torch.distributed process group with more than one rank')
    from metro_pose3d_amd.dist import shard_range
    begin, end = shard_range(n, rank, world) if n else (0, 0)
    # Per-rank failures must not leave the other ranks blocked in the gather: a failing rank records its error, STILL joins the
    # collective (its status row says so) and every rank raises afterwards.  Status: 0 fine, k > 0 = k crops of this rank's
    # shard reached the soft-argmax non-finite (fp16 overflow), -1 = this rank raised -- in the model load, the plan build, the
    # workspace allocation, the shard upload or the forward loop (all inside the try since round 6).
    # What is NOT covered: a failure that leaves this rank unable to take part in a collective at all -- the model file cannot be
    # read far enough to learn the output joint count (every rank passes the same path: they then all raise here, before any
    # collective), a wrong image shape (ValueError on every rank alike), or a sticky HIP error after which no device tensor can
    # be allocated under backend `nccl` (the process group's own timeout ends the other ranks).
    status, err, eng, poses, n_out = 0, None, None, None, None
    with (torch.cuda.device(device) if device.type == 'cuda' else contextlib.nullcontext()):
        try:
            eng = _engine_for(model_path, precision, device, max(end - begin, 1))
            s = eng.spec.proc_side
            n_out = eng.spec.skeleton.n_out
            if images_tensor.dim() != 4 or tuple(images_tensor.shape[1:]) != (s, s, 3):
                raise _ShapeError(f'images must be NHWC [N,{s},{s},3] (reference main.py:109-110), got '
                                  f'{tuple(images_tensor.shape)}')
            images = images_tensor[begin:end].to(device, non_blocking=True).contiguous()      # only this rank's shard goes to its GPU
            poses = torch.empty((end - begin, n_out, 3), dtype=torch.float32, device=device)
            bad = None
            for i in range(0, end - begin, eng.max_batch):
                k = min(eng.max_batch, end - begin - i)
                eng.forward(images[i:i + k], out=poses[i:i + k])
                if check_finite:       # folded on the device after every chunk: ONE synchronisation per call, below
                    cnt = eng.status_words(k).ne(0).sum()
                    bad = cnt if bad is None else bad + cnt
            if bad is not None:
                status = int(bad.item())                                 # the call's one stream synchronisation
        except _ShapeError:
            raise                       # the same on every rank (same images): no collective was entered by anybody
        except Exception as e:          # noqa: BLE001 -- re-raised below, after the collective
            status, err = -1, e
        if world > 1:
            try:
                if err is not None:
                    # join with a FRESH tensor that only carries the status row: `poses` may be unallocated or hold garbage
                    if n_out is None:
                        n_out = load_model(model_path)[0].skeleton.n_out       # raises if the file is unreadable (every rank alike)
                    on_host = torch.distributed.get_backend(group) != 'nccl'
                    poses = torch.zeros((end - begin, n_out, 3), dtype=torch.float32, device='cpu' if on_host else device)
                poses, statuses = _gather_shards(poses, n, group, status)
            except Exception:
                if err is not None:
                    raise err
                raise
        else:
            statuses = [status]
    if err is not None:
        raise err
    if any(st != 0 for st in statuses):
        failed = ', '.join(f'rank {r}: ' + ('raised (model load, plan build, upload or forward)' if st < 0 else f'{st} crops') for r, st in enumerate(statuses) if st != 0)
        if any(st < 0 for st in statuses):
            raise _lib.MetroError(f'estimate_pose failed on another rank ({failed}); no poses returned on any rank')
        raise _lib.NonFiniteError(
            f'{eng.spec.arch_name} stride {eng.spec.stride} in precision {precision!r}: crops reached the soft-argmax with non-finite '
            f'statistics ({failed})' + (' (fp16 storage overflows at 65504: run this model with precision f32m or f64)' if precision == 'f16' else ''))
    sk = eng.spec.skeleton
    names = np.empty(sk.n_out, dtype=object)
    names[:] = sk.names_bytes()
    return poses, sk.edges_array(), names


def visualize_pose(image, coords, edges):
    """Stick-figure plot like reference inference.py:46-76 (matplotlib optional)."""
    import matplotlib
    matplotlib.use(os.environ.get('MPLBACKEND', 'Agg'))
    import matplotlib.pyplot as plt
    from mpl_toolkits.mplot3d import Axes3D  # noqa: F401

    pts = np.asarray(coords, dtype=np.float64)
    # camera frame has y down / z forward; matplotlib wants z up (reference inference.py:52-56)
    plot_pts = np.stack([pts[:, 0], pts[:, 2], -pts[:, 1]], axis=1)
    fig = plt.figure(figsize=(10, 5))
    ax_im = fig.add_subplot(1, 2, 1)
    ax_im.set_title('Input')
    ax_im.imshow(np.clip(np.asarray(image), 0, 1))
    ax = fig.add_subplot(1, 2, 2, projection='3d')
    ax.set_title('Prediction')
    lim = 800
    ax.set_xlim3d(-lim, lim); ax.set_ylim3d(-lim, lim); ax.set_zlim3d(-lim, lim)
    for a, b in np.asarray(edges):
        ax.plot(*zip(plot_pts[a], plot_pts[b]), marker='o', markersize=2)
    ax.scatter(plot_pts[:, 0], plot_pts[:, 1], plot_pts[:, 2], s=2)
    fig.tight_layout()
    return fig


def main(argv=None):
    parser = argparse.ArgumentParser(description='MeTRo-Pose3D on MI355X', allow_abbrev=False)
    parser.add_argument('--model-path', type=str, required=True)
    parser.add_argument('--image', type=str, default=None,
                        help='.npy file with a [256,256,3] float image in [0,1]; default: seeded noise')
    parser.add_argument('--precision', type=str, default=None, choices=['f16', 'f32', 'f32m', 'f64'])
    parser.add_argument('--plot', type=str, default=None, help='write the stick-figure plot to this file')
    opts = parser.parse_args(argv)
    if opts.image:
        img = np.load(opts.image).astype(np.float32)
    else:
        from metro_pose3d_amd.synth import make_images
        img = make_images(1)[0]
    images = torch.from_numpy(img[None])
    poses, edges, names = estimate_pose(images, opts.model_path, precision=opts.precision)
    poses = poses.cpu().numpy()
    for name, p in zip(names, poses[0]):
        print(f'{name.decode():>10s}  {p[0]:9.2f} {p[1]:9.2f} {p[2]:9.2f}')
    if opts.plot:
        visualize_pose(img, poses[0], edges).savefig(opts.plot)


if __name__ == '__main__':
    main()
