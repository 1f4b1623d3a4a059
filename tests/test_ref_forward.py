"""oracle/forward.py (and the HIP path) against the reference's own network-building code EXECUTED ON NUMBERS.

tests/golden/ref_forward_v2.npz was written in the build container by tests/golden/make_ref_forward.py: the reference's
`architectures.resnet` -> `resnet_v2_50/101` -> `stack_blocks_dense` -> `bottleneck` -> `conv2d_same` / `max_pool2d_same` /
`subsample` / `spatial_slice` (their source lines, cut out with `ast`, unmodified) run on a tape whose tensors carry fp64 values --
every op the reference issues is evaluated in NumPy on the weights its variable scope names.  Stored: the logits of two 64-pixel
crops per configuration (stride 4: every second row / column of both + the full map of crop 0) and (sum, sum |x|) of every op output.  Weights and crops are regenerated here from the stored seeds
(metro_pose3d_amd/synth.py); nothing under /root/reference is read.

What the comparison pins: the oracle's dataflow (which tensor feeds which op with which weights, pads, strides, rates, shortcut
taps, centered_stride shifts, postnorm, logits) as NUMBERS of the reference's graph code, to 1e-10 relative in fp64.  What it does
not: TensorFlow's op kernels themselves (the generator's five NumPy kernels stand in for them; header of make_ref_forward.py).
"""
import os

import numpy as np
import pytest
import torch

from metro_pose3d_amd import synth
from oracle import forward as OF
from oracle.spec import OracleSpec, schedule

FIX = os.path.join(os.path.dirname(__file__), 'golden', 'ref_forward_v2.npz')


@pytest.fixture(scope='module')
def fix():
    return np.load(FIX)


def _cases():
    with np.load(FIX) as f:
        return [c.decode() for c in f['cases']]


def _setup(fix, key):
    arch, stride, centered, joints, pseed, side, n, iseed = (int(v) for v in fix[f'{key}/meta'])
    spec = OracleSpec(arch=arch, stride=stride, dataset='h36m' if joints == 17 else 'many19', centered_stride=bool(centered),
                      proc_side=side)
    params = synth.make_params(arch, 8 * joints, 64, seed=pseed)
    images = synth.make_images(n, side, seed=iseed)
    return spec, params, images, stride


def test_fixture_covers_the_configurations():
    keys = _cases()
    assert len(keys) == 9
    assert {k.split('_')[1] for k in keys if k.startswith('rn50')} == {'s4', 's8', 's16', 's32'}
    assert any(k.startswith('rn101') for k in keys)


@pytest.mark.parametrize('key', _cases())
def test_oracle_forward_reproduces_the_reference_graph_on_numbers(fix, key):
    spec, params, images, stride = _setup(fix, key)
    collect = {}
    logits = OF.backbone_logits(spec, params, images, dtype=torch.float64, collect=collect).permute(0, 2, 3, 1).numpy()
    want = fix[f'{key}/logits']
    got = logits[:, ::2, ::2, :] if stride == 4 else logits
    assert got.shape == want.shape
    scale = np.abs(want).max()
    assert np.abs(got - want).max() <= 1e-10 * scale, (key, np.abs(got - want).max() / scale)
    if stride == 4:           # v2: the full stride-4 map of crop 0 (the reference's default test stride, options.py:96)
        full = fix[f'{key}/logits_crop0_full']
        assert full.shape == logits[:1].shape and np.abs(logits[:1] - full).max() <= 1e-10 * scale

    # every intermediate tensor the oracle exposes, against the reference run's op of the same scope: (sum, sum |x|, size)
    names = [n.decode() for n in fix[f'{key}/op_names']]
    stats = fix[f'{key}/op_stats']
    root = f'MainPart/{spec.arch_name}'
    by_name = {n: s for n, s in zip(names, stats)}
    pairs = [(f'conv2d:{root}/conv1', 'conv1'), (f'max_pool2d:{root}/pool1', 'pool1'), (f'batch_norm:{root}/postnorm', 'postnorm'),
             (f'conv2d:{root}/logits', 'logits')]
    for u in schedule(spec):
        sc = f'{root}/{u.name}/bottleneck_v2'
        pairs += [(f'add:{sc}', u.name), (f'conv2d:{sc}/conv1', u.name + '/conv1'), (f'conv2d:{sc}/conv2', u.name + '/conv2')]
    checked = 0
    for ref_name, mine in pairs:
        assert ref_name in by_name, ref_name
        t = collect[mine].double().numpy()
        s, a, size = by_name[ref_name]
        assert t.size == int(size), (ref_name, t.shape, size)
        assert abs(t.sum() - s) <= 1e-9 * a + 1e-12 and abs(np.abs(t).sum() - a) <= 1e-10 * a + 1e-12, (key, ref_name)
        checked += 1
    assert checked == 4 + 3 * len(schedule(spec))


def test_every_op_of_the_reference_run_is_accounted_for(fix):
    """The ops of the reference run = stem (cast, pad, conv1, pad, pool1) + per unit (preact, [slice], shortcut conv | [max_pool],
    conv1, [pad], conv2, conv3, add) + postnorm + logits + unused softmax + cast: nothing the oracle does not model."""
    for key in _cases():
        spec, _, _, _ = _setup(fix, key)
        names = [n.decode() for n in fix[f'{key}/op_names']]
        kinds = [n.split(':')[0] for n in names]
        n_units = len(schedule(spec))
        assert kinds.count('add') == n_units and kinds.count('conv2d') == 2 + 3 * n_units + sum(u.c_in != u.c_out for u in schedule(spec))
        assert kinds.count('batch_norm') == n_units + 1 and kinds.count('cast') == 2
        assert kinds.count('max_pool2d') == 1 + sum(u.c_in == u.c_out and u.stride == 2 for u in schedule(spec))
        assert set(kinds) <= {'cast', 'pad', 'conv2d', 'max_pool2d', 'batch_norm', 'slice', 'add', 'softmax_unused'}


@pytest.mark.gpu
@pytest.mark.parametrize('key', _cases())
def test_hip_f64_path_reproduces_the_reference_graph_on_numbers(fix, key, cuda):
    """The HIP path in its fp64 parity mode on the same crops and weights: poses within 1e-3 mm (the north-star tolerance) of the
    reference run's logits decoded by the oracle's soft-argmax (itself held to the reference's decode lines, test_ref_schedule.py).
    Every stride, 4 included (the reference's default --stride-test, options.py:96; BASELINE configs[4]): there the fixture holds
    the full heat map of crop 0, whose pose is compared."""
    from metro_pose3d_amd import ModelSpec
    from metro_pose3d_amd.engine import Engine
    spec, params, images, stride = _setup(fix, key)
    want = OF.logits_to_output(spec, fix[f'{key}/logits_crop0_full' if stride == 4 else f'{key}/logits']).numpy()
    ms = ModelSpec(spec.arch, spec.stride, spec.dataset, centered_stride=spec.centered_stride, proc_side=spec.proc_side)
    eng = Engine(ms, params, 'f64', max_batch=images.shape[0], device=cuda)
    got = eng.forward(torch.from_numpy(images).to(cuda)).cpu().numpy()
    assert np.isfinite(got).all()
    got = got[:want.shape[0]]
    assert np.abs(got - want).max() <= 1e-3, (key, np.abs(got - want).max())
