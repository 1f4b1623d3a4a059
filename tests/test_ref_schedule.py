"""The oracle's restatement of the graph's control flow (oracle/spec.py, oracle/forward.py) AND the product's planner
(csrc/plan.cpp through metro_plan_layer_info) are each held to what the REFERENCE'S OWN Python produced when its
graph-building functions were executed in the build container (tests/golden/make_ref_schedule.py ->
tests/golden/ref_schedule_v1.npz: resnet_v2_50/101, resnet_v2_block, bottleneck, stack_blocks_dense, conv2d_same,
max_pool2d_same, architectures.resnet on a recording tape; build_inference_model, net_output_to_heatmap_and_coords,
tfu.softmax, tfu.decode_heatmap, heatmap_to_image/metric, root_relative on NumPy).

This pins CONTROL FLOW (which ops, wired how, with which stride / rate / padding / pads / bias / activation; which axis is
x, y, z; channel order; decode constants; root joint; export permutation), not TensorFlow's floating-point arithmetic."""
import json
import os

import numpy as np
import pytest
import torch

from metro_pose3d_amd import ModelSpec, _lib
from metro_pose3d_amd.engine import Engine
from oracle import forward as OF
from oracle.spec import OracleSpec, decode_constants, schedule

from tests import helpers as H

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ref_schedule_v1.npz')
CONFIGS = [(a, s, c) for a in (50, 101) for s in (4, 8, 16, 32) for c in (True, False)]
ids = lambda k: f'rn{k[0]}-s{k[1]}-{"centered" if k[2] else "plain"}'


@pytest.fixture(scope='module')
def ref():
    return np.load(FIX)


def ref_units(ref, arch, stride, centered):
    key = f'rn{arch}_s{stride}_{"c" if centered else "n"}'
    cols = [c.decode() for c in ref['unit_columns']]
    return key, [dict(zip(cols, (int(v) for v in row)), name=n.decode()) for n, row in zip(ref[f'{key}/unit_names'], ref[f'{key}/units'])]


def ref_conv2_pad_beg(r):
    """Pad in front of conv2 as the reference graph applies it: explicit array_ops.pad (VALID) or TensorFlow's SAME rule."""
    if not r['conv2_padding_same']:
        return r['conv2_pad_beg'], r['conv2_pad_end']
    k_eff = 3 + 2 * (r['conv2_rate'] - 1)
    return OF.tf_same_pads(r['side_in'], k_eff, r['conv2_stride'])


def test_fixture_covers_what_it_claims(ref):
    assert sorted(c.decode() for c in ref['configs']) == sorted(f'rn{a}_s{s}_{"c" if c else "n"}' for a, s, c in CONFIGS)
    assert len(ref['decode_cases']) == 10
    tape = json.loads(bytes(ref['rn50_s16_c/tape_json']).decode())
    kinds = {o['op'] for o in tape}
    assert kinds == {'cast', 'pad', 'conv2d', 'max_pool2d', 'batch_norm', 'slice', 'add', 'softmax_unused'}
    # scopes are the variable names a frozen graph carries (tfgraph.py maps them): spot-check the reference's spelling
    scopes = {o['scope'] for o in tape}
    for s in ('MainPart/resnet_v2_50/conv1', 'MainPart/resnet_v2_50/block1/unit_1/bottleneck_v2/shortcut',
              'MainPart/resnet_v2_50/block4/unit_3/bottleneck_v2/conv2', 'MainPart/resnet_v2_50/postnorm', 'MainPart/resnet_v2_50/logits'):
        assert s in scopes


@pytest.mark.parametrize('arch', [50, 101])
def test_variable_names_and_shapes_equal_reference_tape(ref, arch):
    """The slim variable set a checkpoint / frozen graph of the reference carries -- every conv's HWIO weights, a bias iff the
    conv has no normalizer, four BatchNorm vectors per normalised conv and per stand-alone batch_norm -- read off the tape,
    against what the product consumes (synth.make_params has the key set and shapes engine.pack_param / tfgraph.py map)."""
    from metro_pose3d_amd import synth
    tape = json.loads(bytes(ref[f'rn{arch}_s16_c/tape_json']).decode())
    want = {}
    for o in tape:
        if o['op'] == 'conv2d':
            want[o['scope'] + '/weights'] = (*o['kernel'], o['c_in'], o['c_out'])
            if o['has_bias']:
                want[o['scope'] + '/biases'] = (o['c_out'],)
            if o['normalizer'] == 'batch_norm':
                for v in ('gamma', 'beta', 'moving_mean', 'moving_variance'):
                    want[f"{o['scope']}/BatchNorm/{v}"] = (o['c_out'],)
        elif o['op'] == 'batch_norm':
            for v in ('gamma', 'beta', 'moving_mean', 'moving_variance'):
                want[f"{o['scope']}/{v}"] = (o['in_shape'][3],)
    got = {k: tuple(v.shape) for k, v in synth.make_params(arch, 136, 64, seed=0).items()}
    assert got == want


@pytest.mark.parametrize('cfg', CONFIGS, ids=ids)
def test_oracle_schedule_equals_reference_tape(ref, cfg):
    """KA6 / KA8 / KA9 / KA12 as reference outputs: oracle/spec.schedule + oracle/forward's pad and shortcut rules."""
    arch, stride, centered = cfg
    key, runits = ref_units(ref, arch, stride, centered)
    units = schedule(OracleSpec(arch=arch, stride=stride, centered_stride=centered))
    assert [u.name for u in units] == [r['name'] for r in runits]
    for u, r in zip(units, runits):
        assert (u.block, u.unit, u.c_in, u.c_out, u.c_bott, u.stride, u.rate, u.side_in, u.side_out) == \
            (r['block'], r['unit'], r['c_in'], r['c_out'], r['c_bott'], r['conv2_stride'], r['conv2_rate'], r['side_in'], r['side_out']), u.name
        # conv2d_same as oracle/forward.conv2d_same decides it (SAME when stride 1 or centered, else explicit pad + VALID)
        k_eff = 3 + 2 * (u.rate - 1)
        if u.stride == 1 or u.centered:
            assert r['conv2_padding_same'] == 1 and (r['conv2_pad_beg'], r['conv2_pad_end']) == (0, 0), u.name
            pads = OF.tf_same_pads(u.side_in, k_eff, u.stride)
        else:
            assert r['conv2_padding_same'] == 0, u.name
            pads = ((k_eff - 1) // 2, k_eff - 1 - (k_eff - 1) // 2)
        assert pads == ref_conv2_pad_beg(r), u.name
        # shortcut (oracle/forward.bottleneck): projection on shift(preact) iff c_in != c_out, else subsample(shift(x), stride)
        shift = 1 if (u.centered and u.stride == 2) else 0
        assert r['shortcut_is_projection'] == int(u.c_in != u.c_out) and r['shortcut_from_preact'] == int(u.c_in != u.c_out), u.name
        assert (r['shortcut_stride'], r['shortcut_shift']) == (u.stride, shift), u.name
        assert (r['conv1_bias'], r['conv2_bias'], r['conv3_bias'], r['conv1_bn_relu'], r['conv2_bn_relu'], r['conv3_linear']) == (0, 0, 1, 1, 1, 1)
    # stem / pool / postnorm / logits (oracle/forward.backbone_logits)
    assert ref[f'{key}/conv1'].tolist() == [7, 2, 0, 3, 3, 1, 1, 64, 128]            # 7x7/2, VALID after pad (3,3), bias, no BN/ReLU
    assert ref[f'{key}/pool1'].tolist() == [3, 2, 0, 1, 1, 64]                        # 3x3/2, VALID after ZERO pad (1,1): never centered
    assert ref[f'{key}/postnorm'].tolist() == [1.0, 1.0, 1e-5, 0.0]                   # relu, gamma, eps 1e-5 (OF.BN_EPS), inference mode
    assert OF.BN_EPS == 1e-5
    assert ref[f'{key}/logits'].tolist() == [1, 1, 1, 1, 2048, 136, 256 // stride]
    assert ref[f'{key}/out_shape'].tolist() == [-1, 256 // stride, 256 // stride, 136]
    assert [c.decode() for c in ref[f'{key}/casts']] == ['float16', 'float32']       # architectures.py:29,34
    # decode constants (KA4): heatmap_to_image(0), heatmap_to_image(1)
    lrc, half = decode_constants(OracleSpec(arch=arch, stride=stride, centered_stride=centered))
    assert ref[f'{key}/decode'].tolist() == [half, half + lrc]


@pytest.mark.parametrize('cfg', CONFIGS, ids=ids)
@pytest.mark.parametrize('prec', ['f16', 'f64'])
def test_planner_equals_reference_tape(ref, cfg, prec):
    """csrc/plan.cpp (metro_plan_layer_info) against the reference's tape directly -- not through oracle/spec.py."""
    arch, stride, centered = cfg
    key, runits = ref_units(ref, arch, stride, centered)
    layers = {li.name.decode(): li for li in Engine(ModelSpec(arch, stride, 'h36m', centered_stride=centered), None, prec, max_batch=1).layer_infos()}
    if 'conv1+pool1' in layers:
        st = layers['conv1+pool1']
        assert (st.h_out, st.c_out, st.kh, st.stride) == (64, 64, 7, 2)         # stem conv + zero-pad max-pool as one launch
    else:
        c1, p1 = layers['conv1'], layers['pool1']
        assert (c1.kh, c1.stride, c1.pad_top, c1.h_out, c1.c_out, c1.relu, c1.has_prologue) == (7, 2, 3, 128, 64, 0, 0)
        assert (p1.kh, p1.stride, p1.pad_top, p1.h_out) == (3, 2, 1, 64)
    lg = layers['logits']
    assert (lg.kh, lg.c_in, lg.c_out, lg.h_out, lg.has_prologue, lg.relu) == (1, 2048, 136, 256 // stride, 1, 0)
    # launches that carry more than one of the reference's ops: name -> the layer info that holds each op's geometry
    alias, next_conv1 = {}, {}
    for name, li in layers.items():
        if '/conv3+' in name:                           # conv3 of unit u + conv1 of the next unit in one launch
            unit, nxt = name.split('/conv3+')
            alias[f'{unit}/conv3'] = li
            next_conv1[f'{unit.split("/")[0]}/{nxt[:-len("/conv1")]}'] = li
    covered = 0
    for r in runits:
        n = r['name']
        pad_beg, _ = ref_conv2_pad_beg(r)
        c2 = layers.get(f'{n}/conv2') or layers[f'{n}/conv1+conv2']
        assert (c2.kh, c2.kw, c2.stride, c2.dilation, c2.pad_top, c2.pad_left, c2.h_in, c2.h_out, c2.c_out, c2.relu) == \
            (3, 3, r['conv2_stride'], r['conv2_rate'], pad_beg, pad_beg, r['side_in'], r['side_out'], r['c_bott'], 1), n
        c3 = layers.get(f'{n}/conv3') or alias[f'{n}/conv3']
        assert (c3.kh, c3.c_in, c3.c_out, c3.relu, c3.has_prologue, c3.h_out) == (1, r['c_bott'], r['c_out'], 0, 0, r['side_out']), n
        # conv1: own launch, fused behind the previous unit's conv3, in front of conv2, or paired with the projection shortcut
        if f'{n}/conv1' in layers:
            c1 = layers[f'{n}/conv1']
            assert (c1.kh, c1.stride, c1.c_in, c1.c_out, c1.has_prologue, c1.relu, c1.h_in) == (1, 1, r['c_in'], r['c_bott'], 1, 1, r['side_in']), n
        elif n in next_conv1:
            assert next_conv1[n].out2_channels == r['c_bott'] and next_conv1[n].c_out == r['c_in'], n
        elif f'{n}/conv1+conv2' in layers:
            assert layers[f'{n}/conv1+conv2'].fused_flags == _lib.FUSED_CONV1_IN_FRONT and layers[f'{n}/conv1+conv2'].c_in == r['c_in'], n
        else:
            pair = layers[f'{n}/shortcut+conv1']
            assert (pair.c_in, pair.c_out, pair.has_prologue, pair.h_in) == (r['c_in'], r['c_out'], 1, r['side_in']), n
        # the shortcut
        if r['shortcut_is_projection']:
            assert r['shortcut_from_preact'] == 1
            if f'{n}/shortcut' in layers:
                sc = layers[f'{n}/shortcut']
                assert (sc.kh, sc.stride, sc.pad_top, sc.has_prologue, sc.c_out, sc.relu) == (1, r['shortcut_stride'], -r['shortcut_shift'], 1, r['c_out'], 0), n
                assert (c3.has_residual, c3.res_stride, c3.res_offset) == (1, 1, 0), n
            elif f'{n}/shortcut+conv1' in layers:
                assert (r['shortcut_stride'], r['shortcut_shift']) == (1, 0) and (c3.has_residual, c3.res_stride, c3.res_offset) == (1, 1, 0), n
            else:
                assert c3.fused_flags & _lib.FUSED_PROJECTION_SHORTCUT and (r['shortcut_stride'], r['shortcut_shift']) == (1, 0), n
        else:
            assert r['shortcut_from_preact'] == 0 and f'{n}/shortcut' not in layers
            # (round 5, block1: the shortcut may be rebuilt in the launch or arrive as a compact sub-sampled copy -- the info
            # states the reference's shortcut either way, and the copy's geometry is that gather)
            assert (c3.has_residual, c3.res_stride, c3.res_offset) == (1, r['shortcut_stride'], r['shortcut_shift']), n
            if c3.fused_flags & _lib.FUSED_COMPACT_SHORTCUT:
                prev = alias[f"{n.split('/')[0]}/unit_{int(n.split('_')[-1]) - 1}/conv3"]
                assert (prev.out_sub_side, prev.out_sub_off) == (r['side_out'], r['shortcut_shift']) and r['shortcut_stride'] == 2, n
        covered += 1
    assert covered == len(runits) == {50: 16, 101: 33}[arch]


DECODE = [(32, 'h36m'), (16, 'h36m'), (8, 'many19'), (4, 'h36m'), (16, 'merged')]


@pytest.mark.parametrize('stride,ds', DECODE)
@pytest.mark.parametrize('centered', [True, False])
def test_oracle_decode_equals_reference_lines(ref, stride, ds, centered):
    """oracle/forward.soft_argmax01 + coords01_to_output against build_inference_model's own lines (KA1, KA2, KA4, KA5)."""
    key = f'decode/s{stride}_{ds}_{"c" if centered else "n"}'
    logits = ref[f'{key}/logits']
    spec = OracleSpec(arch=50, stride=stride, dataset=ds, centered_stride=centered)
    from oracle.spec import export_permutation, head_joint_info
    assert list(export_permutation(ds)) == ref[f'{key}/permutation'].tolist()
    j = head_joint_info(ds).n_joints
    assert logits.shape[-1] == 8 * j
    for dtype, tol01, tolmm in ((torch.float64, 1e-12, 1e-9), (torch.float32, 2e-6, 4e-3)):
        lg = torch.from_numpy(logits).to(dtype).permute(0, 3, 1, 2)
        p, c01 = OF.soft_argmax01(lg, j, 8)
        assert np.abs(c01.numpy() - ref[f'{key}/coords01']).max() <= tol01
        assert np.abs(p.sum(dim=(2, 3)).numpy() - ref[f'{key}/heatmap_pred_z']).max() <= 1e-5          # volumetric.py:165 (z marginal)
        out = OF.coords01_to_output(spec, c01).numpy()
        assert out.shape == ref[f'{key}/output'].shape
        assert np.abs(out - ref[f'{key}/output']).max() <= tolmm
    # image 1 holds a peak per joint at (w, h, d) = ((3j+1) % S, (5j+2) % S, j % 8): the reference's own answer is KA1
    s = 256 // stride
    c = ref[f'{key}/coords01'][1]
    exp = np.array([[((3 * jj + 1) % s) / (s - 1), ((5 * jj + 2) % s) / (s - 1), (jj % 8) / 7] for jj in range(j)])
    assert np.abs(c - exp).max() < 1e-6
    assert np.abs(ref[f'{key}/coords01'][2] - 0.5).max() < 1e-6                                        # KA2


@pytest.mark.gpu
@pytest.mark.parametrize('stride,ds', DECODE)
@pytest.mark.parametrize('centered', [True, False])
@pytest.mark.parametrize('precise', [0, 1, 2])
def test_hip_softargmax_equals_reference_lines(ref, lib, cuda, stride, ds, centered, precise):
    """metro_softargmax (partial + finalize: softmax over the volume, expectation, mm decode, root, export gather) against
    the poses the reference's own decode lines produce from the same logits."""
    key = f'decode/s{stride}_{ds}_{"c" if centered else "n"}'
    spec = ModelSpec(50, stride, ds, centered_stride=centered)
    got = H.run_softargmax(lib, cuda, spec, ref[f'{key}/logits'], precise)
    want = ref[f'{key}/output']
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 2e-3, np.abs(got - want).max()
