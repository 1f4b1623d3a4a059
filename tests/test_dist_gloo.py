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
