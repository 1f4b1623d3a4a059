"""Committed golden vectors (tests/golden/golden_poses_v1.npz, made by make_golden.py with the
fp64 oracle in the build container).  CPU: the oracle and the seeded generators still reproduce
them.  GPU (`-m gpu`): the HIP path reproduces them in every precision mode."""
import json
import os
import zlib

import numpy as np
import pytest
import torch

from metro_pose3d_amd import ModelSpec, synth
from oracle import forward as OF
from tests import helpers as H

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'golden_poses_v1.npz')
GOLD_F16 = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'golden_f16emu_v1.npz')
F16_POSE_RATIO = 2.0     # f16 mode vs exact math: mean |d| at most this multiple of the fp16 oracle's own
F16_POSE_RATIO_MAX = 2.5 # ... and the maximum (a noisy statistic of 17-57 values per crop) within this factor


@pytest.fixture(scope='module')
def gold():
    z = np.load(GOLD)
    return z, json.loads(bytes(z['__meta__']).decode())


def _case(meta, name):
    m = meta[name]
    spec = ModelSpec(**m['spec'])
    params = synth.make_params(spec.arch, spec.n_head_channels, spec.base_width, seed=m['param_seed'],
                               logit_gain=m['logit_gain'])
    images = synth.make_images(m['batch'], spec.proc_side, seed=m['image_seed'])
    return spec, params, images


def _crc(params):
    crc = 0
    for k in sorted(params):
        crc = zlib.crc32(k.encode(), crc)
        crc = zlib.crc32(np.ascontiguousarray(params[k]).tobytes(), crc)
    return crc


FULL = ['rn50-s32-h36m', 'rn50-s16-h36m', 'rn50-s16-many19', 'rn101-s8-many19', 'rn50-s4-h36m', 'rn50-s16-merged53']
TOY = ['toy-rn50-s32-w8', 'toy-rn50-s16-w8', 'toy-rn50-s8-w8', 'toy-rn50-s4-w8', 'toy-rn101-s8-w8',
       'toy-rn101-s4-w8', 'toy-rn50-s16-w16-noncentered']


@pytest.mark.parametrize('name', FULL + TOY)
def test_seeded_generators_are_stable(gold, name):
    """weights/images are regenerated from seeds on every machine: their CRCs must not drift."""
    z, meta = gold
    spec, params, images = _case(meta, name)
    assert _crc(params) == meta[name]['params_crc32']
    assert zlib.crc32(images.tobytes()) == meta[name]['images_crc32']


@pytest.mark.parametrize('name', TOY + ['rn50-s32-h36m'])
def test_oracle_reproduces_golden(gold, name):
    z, meta = gold
    spec, params, images = _case(meta, name)
    col = {}
    got = OF.forward(H.oracle_spec(spec), params, images, torch.float64, col).numpy()
    assert np.abs(got - z[name + '/poses']).max() < 1e-7        # mm; fp64 thread-order noise only
    assert np.abs(col['coords01'].numpy() - z[name + '/coords01']).max() < 1e-10
    for key in ('conv1', 'pool1', 'block1/unit_1', 'block2/unit_4', 'block4/unit_3', 'logits'):
        t = col[key]
        flat = t.reshape(-1)
        idx = np.linspace(0, flat.numel() - 1, 16).astype(np.int64)
        probe = np.concatenate([flat[idx].numpy(), [float(t.mean()), float(t.abs().mean())]])
        ref = z[f'{name}/probe/{key}']
        assert np.abs(probe - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max()), key


@pytest.mark.parametrize('name', TOY + ['rn50-s32-h36m'])
def test_f16emu_oracle_reproduces_golden(gold, name):
    """oracle/f16emu.py (the fp16 arithmetic model the f16 mode is held to) still gives the committed poses and
    probe values: exact arithmetic between roundings, so the bar is fp64 summation-order noise at a rounding
    boundary, i.e. equality up to isolated one-ulp flips -- poses within 1e-6 mm in practice."""
    from oracle import f16emu
    z, meta = gold
    z16 = np.load(GOLD_F16)
    spec, params, images = _case(meta, name)
    col = {}
    got = f16emu.forward(H.oracle_spec(spec), params, images, col).numpy()
    assert np.abs(got - z16[name + '/poses_f16emu']).max() < 1e-3
    for key in ('pool1', 'block1/unit_1', 'logits'):
        flat = col[key].reshape(-1)
        idx = np.linspace(0, flat.numel() - 1, 16).astype(np.int64)
        ref = z16[f'{name}/probe16/{key}']
        assert np.abs(flat[idx].numpy() - ref).max() <= 2e-3 * max(1.0, np.abs(ref).max()), key
    # and it is an fp16 model of the SAME graph: a few mm from exact math, not more
    assert np.abs(got - z[name + '/poses']).max() < 12.0


def test_softargmax_golden_on_cpu(gold):
    z, meta = gold
    for name in ('sa-rn50-s16-h36m', 'sa-rn101-s8-merged'):
        spec = ModelSpec(**meta[name]['spec'])
        lg = (np.random.default_rng(77).standard_normal(
            (2, spec.heatmap_side, spec.heatmap_side, spec.n_head_channels)) * 4).astype(np.float32)
        assert zlib.crc32(lg.tobytes()) == meta[name]['logits_crc32']
        assert np.abs(OF.logits_to_output(H.oracle_spec(spec), lg).numpy() - z[name + '/poses']).max() < 1e-9


# ---- GPU ---------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize('name', FULL + TOY)
def test_hip_path_reproduces_golden(gold, cuda, name):
    from metro_pose3d_amd.engine import Engine
    z, meta = gold
    spec, params, images = _case(meta, name)
    ref = z[name + '/poses']
    x = torch.from_numpy(images).to(cuda)
    got64 = Engine(spec, params, 'f64', max_batch=len(images), device=cuda).forward(x).cpu().numpy()
    assert np.abs(got64 - ref).max() <= 1e-3, np.abs(got64 - ref).max()          # the north-star bar
    # the benchmarked f16 mode: as close to exact math as the fp16 model of the graph (oracle/f16emu.py) is -- fp16
    # storage costs a few mm on these nets; tests/test_f16_layerwise.py holds every launch to a rounding flip
    emu = np.load(GOLD_F16)[name + '/poses_f16emu']
    got16 = Engine(spec, params, 'f16', max_batch=len(images), device=cuda).forward(x).cpu().numpy()
    assert np.isfinite(got16).all()
    e16, eemu = np.abs(got16 - ref), np.abs(emu - ref)
    assert e16.max() <= F16_POSE_RATIO_MAX * eemu.max() and e16.mean() <= F16_POSE_RATIO * eemu.mean(), \
        (e16.max(), eemu.max(), e16.mean(), eemu.mean())


@pytest.mark.gpu
def test_hip_softargmax_reproduces_golden(gold, cuda, lib):
    z, meta = gold
    for name in ('sa-rn50-s16-h36m', 'sa-rn101-s8-merged'):
        spec = ModelSpec(**meta[name]['spec'])
        lg = (np.random.default_rng(77).standard_normal(
            (2, spec.heatmap_side, spec.heatmap_side, spec.n_head_channels)) * 4).astype(np.float32)
        for precise, tol in ((0, 1e-3), (1, 1e-3), (2, 1e-3)):
            got = H.run_softargmax(lib, cuda, spec, lg, precise)
            assert np.abs(got - z[name + '/poses']).max() <= tol
