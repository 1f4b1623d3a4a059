"""Batch sharding + the one all-gather of pose outputs, on CPU with gloo, world_size 2 and 3.

The compute stub is a deterministic per-image function, which is all the sharding logic may
assume about the hot path (no op crosses the batch dimension).  The gathered
result must be bit-identical to the unsharded one, for even and ragged batches."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from metro_pose3d_amd.dist import OverlappedPoseGather, all_gather_poses, shard_range, sharded_forward


def _fake_forward(images):                      # [n, 4, 4, 3] -> [n, 5, 3], per-image only
    n = images.shape[0]
    s = images.reshape(n, -1)
    return torch.stack([s[:, i::5][:, :3] * (i + 1) for i in range(5)], dim=1).contiguous()


def _worker(rank, world, port, n, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(7)
        images = torch.rand((n, 4, 4, 3), generator=g)
        full = _fake_forward(images)
        got = sharded_forward(_fake_forward, images)
        b, e = shard_range(n, rank, world)
        again = all_gather_poses(_fake_forward(images[b:e]) if e > b else _fake_forward(images[:1])[:0], n)
        ok3 = True
        if n % world == 0:
            # overlapped, double-buffered gather over several "steps": every step's result is exact
            nl = n // world
            g = OverlappedPoseGather(nl, 5, world, torch.device('cpu'))
            for step in range(5):
                buf = g.local_buffer(step)
                buf.copy_(_fake_forward(images[b:e]) + step)
                g.submit(step)
                if step >= 1:
                    ok3 = ok3 and bool(torch.equal(g.result(step - 1), full + (step - 1)))
            g.finish()
            ok3 = ok3 and bool(torch.equal(g.result(4), full + 4))
        q.put((rank, bool(torch.equal(got, full)), bool(torch.equal(again, full)) and ok3))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@pytest.mark.parametrize('world,n', [(2, 8), (2, 7), (3, 8), (2, 1)])
def test_sharded_forward_equals_unsharded(world, n):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=10) for _ in range(world))
    assert res == [(r, True, True) for r in range(world)]


def test_shard_range_partitions_exactly():
    for n in (1, 7, 64, 512):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(e - b for b, e in spans) - min(e - b for b, e in spans) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


# ---- the drop-in call itself: estimate_pose(images, model_path) under an initialised process group -------------------------
class _FakeEngine:
    """Stands in for Engine in the CPU test of estimate_pose's sharding logic (the real engine needs a GPU; the real-engine
    version of this test is tests/test_gpu_forward.py::test_estimate_pose_shards_across_ranks)."""

    def __init__(self, max_batch, bad_crops=0, raise_in_forward=False):
        from metro_pose3d_amd import ModelSpec
        self.spec = ModelSpec(50, 16, 'h36m')
        self.max_batch = max_batch
        self.calls = []
        self.bad_crops = bad_crops                   # crops of every forward flagged non-finite by the stand-in screen
        self.raise_in_forward = raise_in_forward

    def forward(self, images, out=None):
        self.calls.append(int(images.shape[0]))
        if self.raise_in_forward:
            raise RuntimeError('stand-in HIP failure on this rank')
        n = images.shape[0]
        flat = images.reshape(n, -1)
        res = torch.stack([flat[:, 1000 * j:1000 * j + 3] * (j + 1) for j in range(17)], dim=1)
        out.copy_(res)
        return out

    def status_words(self, n):
        self.calls.append(('check', n))
        w = torch.zeros(n, dtype=torch.int32)
        w[:min(self.bad_crops, n)] = 1
        return w


def _estimate_pose_worker(rank, world, port, n, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from metro_pose3d_amd import inference as INF
        from metro_pose3d_amd.inference import batch_bucket
        engines = {}

        def engine_for(model_path, precision, device, n_call=64):
            return engines.setdefault(batch_bucket(n_call), _FakeEngine(batch_bucket(n_call)))
        INF._engine_for = engine_for
        INF._resolve_device = lambda t: torch.device('cpu')
        g = torch.Generator().manual_seed(11)
        images = torch.rand((n, 256, 256, 3), generator=g)
        poses, edges, names = INF.estimate_pose(images, 'unused.npz')
        local, _, _ = INF.estimate_pose(images, 'unused.npz', shard=False)
        b, e = shard_range(n, rank, world)
        calls = [c for eng in engines.values() for c in eng.calls]
        q.put((rank, bool(torch.equal(poses, local)), tuple(poses.shape), len(names), (e - b) in calls or e == b, ('check', max(e - b, 0)) in calls or e == b))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,n', [(2, 10), (2, 7), (3, 4), (2, 1)])
def test_estimate_pose_shards_by_image_and_gathers(world, n):
    """Every rank calls estimate_pose with the same N crops: rank r forwards only shard_range(N, r, world), every rank returns
    all N poses, identical to the unsharded call (the reference's call has one signature: inference.py:31-43)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_estimate_pose_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=10) for _ in range(world))
    assert res == [(r, True, (n, 17, 3), 17, True, True) for r in range(world)]


def _failing_rank_worker(rank, world, port, n, mode, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from metro_pose3d_amd import _lib
        from metro_pose3d_amd import inference as INF
        # only rank 1 misbehaves: its shard "overflows" (2 crops flagged) or its forward raises
        eng = _FakeEngine(64, bad_crops=2 if (mode == 'overflow' and rank == 1) else 0,
                          raise_in_forward=(mode == 'raise' and rank == 1))
        INF._engine_for = lambda model_path, precision, device, n_call=64: eng
        model_path = 'unused.npz'
        if mode == 'engine':
            # round 6: rank 1 fails BEFORE the forward loop -- plan build / workspace allocation inside _engine_for.  It
            # still joins the gather: the output joint count comes from the model file, which it can read
            import tempfile
            from metro_pose3d_amd import ModelSpec, save_model, synth
            spec = ModelSpec(50, 32, 'h36m', base_width=8)
            model_path = os.path.join(tempfile.mkdtemp(prefix=f'metro_gloo_{rank}_'), 'toy.npz')
            save_model(model_path, spec, synth.make_params(spec.arch, spec.n_head_channels, spec.base_width, seed=0))
            if rank == 1:
                def failing(model_path, precision, device, n_call=64):
                    raise RuntimeError('stand-in plan failure on this rank')
                INF._engine_for = failing
        INF._resolve_device = lambda t: torch.device('cpu')
        images = torch.rand((n, 256, 256, 3), generator=torch.Generator().manual_seed(3))
        try:
            INF.estimate_pose(images, model_path)
            q.put((rank, 'returned', ''))
        except _lib.NonFiniteError as e:
            q.put((rank, 'nonfinite', str(e)))
        except _lib.MetroError as e:
            q.put((rank, 'metro', str(e)))
        except RuntimeError as e:
            q.put((rank, 'runtime', str(e)))
        # the group is still usable: nobody is stuck in a half-joined collective
        t = torch.tensor([float(rank)])
        dist.all_reduce(t)
        q.put((rank, 'after', float(t.item())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('mode', ['overflow', 'raise', 'engine'])
def test_a_failure_on_one_rank_raises_on_every_rank_instead_of_hanging(mode):
    """rank 1's shard overflows fp16 / its forward raises.  Every rank must still join the one gather and
    then raise -- the healthy rank may not block in all_gather_into_tensor until the watchdog fires."""
    world, n = 2, 6
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_rank_worker, args=(r, world, port, n, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0, 'a rank hung or crashed'
    res = sorted(q.get(timeout=10) for _ in range(2 * world))
    outcome = {r: (kind, msg) for r, kind, msg in res if kind != 'after'}
    after = {r: v for r, kind, v in res if kind == 'after'}
    assert after == {0: 1.0, 1: 1.0}
    if mode == 'overflow':
        assert outcome[0][0] == outcome[1][0] == 'nonfinite'
        assert 'rank 1: 2 crops' in outcome[0][1] and 'f32m' in outcome[0][1]
    elif mode == 'engine':
        assert outcome[1] == ('runtime', 'stand-in plan failure on this rank')
        assert outcome[0][0] == 'metro' and 'rank 1: raised' in outcome[0][1]
    else:
        assert outcome[1] == ('runtime', 'stand-in HIP failure on this rank')       # the failing rank re-raises its own error
        assert outcome[0][0] == 'metro' and 'rank 1: raised' in outcome[0][1]
