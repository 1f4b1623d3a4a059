"""CPU-side checks of the boundary: the C ABI library loads and exports every symbol the header
declares, ctypes struct layouts equal the compiler's, and the C++ planner agrees with the
oracle's independent restatement of the reference's scheduler.  No compute, no GPU."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from metro_pose3d_amd import ModelSpec, _lib, synth
from metro_pose3d_amd.engine import Engine, pack_param
from oracle.spec import OracleSpec, schedule
from oracle.forward import tf_same_pads

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'metro_hip.h')


def _declared_functions():
    text = re.sub(r'/\*.*?\*/', '', open(HEADER).read(), flags=re.S)
    return sorted(set(re.findall(r'\b(metro_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol(lib):
    declared = _declared_functions()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in include/metro_hip.h but not exported'
    assert sorted(_lib.SIGNATURES) == declared, 'bindings and header disagree'
    assert lib.metro_abi_version() == _lib.ABI_VERSION == 8


def test_experimental_and_probe_libraries_export_their_symbols(lib):
    """libmetro_experimental.so (kernels metro_forward never dispatches) exports what its header declares and NOTHING of it is
    in the product's header; tools/libmetro_probe.so (bench.py's measured ceilings) exports its three entry points."""
    xhdr = os.path.join(ROOT, 'metro_pose3d_amd', 'csrc', 'experimental', 'metro_experimental.h')
    text = re.sub(r'/\*.*?\*/', '', open(xhdr).read(), flags=re.S)
    declared = sorted(set(re.findall(r'\b(metro_[a-z0-9_]+)\s*\(', text)))
    assert declared == sorted(_lib.EXPERIMENTAL_SIGNATURES)
    assert not set(declared) & set(_declared_functions())
    xlib = _lib.load_experimental()
    for name in declared:
        assert hasattr(xlib, name)
        assert not hasattr(lib, name), f'{name} is still exported by the product library'
    probe = C.CDLL(os.path.join(ROOT, 'tools', 'libmetro_probe.so'))
    for name in ('metro_probe_mfma_f16', 'metro_probe_hbm', 'metro_probe_last_error'):
        assert hasattr(probe, name)


def test_ctypes_struct_layout_matches_compiler(tmp_path):
    src = tmp_path / 'layout.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "metro_hip.h"\nint main(void){'
                   'printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(MetroSpec), sizeof(MetroParamInfo), '
                   'sizeof(MetroLayerInfo), sizeof(MetroConvDesc), offsetof(MetroParamInfo, offset), '
                   'offsetof(MetroLayerInfo, flops_per_image), offsetof(MetroSpec, permutation));return 0;}')
    exe = tmp_path / 'layout'
    subprocess.check_call(['gcc', '-std=c99', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    exp = [C.sizeof(_lib.MetroSpec), C.sizeof(_lib.MetroParamInfo), C.sizeof(_lib.MetroLayerInfo),
           C.sizeof(_lib.MetroConvDesc), _lib.MetroParamInfo.offset.offset,
           _lib.MetroLayerInfo.flops_per_image.offset, _lib.MetroSpec.permutation.offset]
    assert got == exp


# SURVEY.md 8(d) / BASELINE.md section 4: algorithmic GFLOP per crop
FLOPS = {(50, 32, 'h36m'): 9.127, (50, 16, 'h36m'): 15.299, (50, 16, 'many19'): 15.316,
         (101, 8, 'many19'): 88.733, (50, 4, 'h36m'): 194.656, (50, 8, 'h36m'): 49.877,
         (101, 4, 'many19'): 350.08, (50, 16, 'merged'): 15.601, (101, 8, 'merged'): 89.873}


@pytest.mark.parametrize('key', sorted(FLOPS), ids=lambda k: f'rn{k[0]}-s{k[1]}-{k[2]}')
def test_plan_flops_match_published_accounting(key):
    arch, stride, ds = key
    for prec in ('f16', 'f64'):
        eng = Engine(ModelSpec(arch, stride, ds), None, prec, max_batch=2)
        assert abs(eng.flops_per_image / 1e9 - FLOPS[key]) < 5e-3 * max(1, FLOPS[key] / 100)


@pytest.mark.parametrize('arch', [50, 101])
@pytest.mark.parametrize('stride', [32, 16, 8, 4])
@pytest.mark.parametrize('centered', [True, False])
@pytest.mark.parametrize('prec', ['f16', 'f64'])
def test_planner_agrees_with_oracle_schedule(arch, stride, centered, prec):
    spec = ModelSpec(arch, stride, 'h36m', centered_stride=centered)
    units = schedule(OracleSpec(arch=arch, stride=stride, centered_stride=centered))
    layers = {li.name.decode(): li for li in Engine(spec, None, prec, max_batch=1).layer_infos()}
    if 'conv1+pool1' in layers:       # fp16, base width 64: stem conv and max-pool are one launch
        assert prec == 'f16' and layers['conv1+pool1'].h_out == 64 and 'pool1' not in layers
    else:
        assert layers['conv1'].h_out == 128 and layers['pool1'].h_out == 64
    # fp16 plans of full-width block1 run conv1 of unit u+1 inside the conv3 launch of unit u
    fused_into = {}
    for name, li in list(layers.items()):
        if '/conv3+' in name:
            unit, nxt = name.split('/conv3+')
            layers[f'{unit}/conv3'] = li
            fused_into[f'{unit.split("/")[0]}/{nxt[:-len("/conv1")]}'] = li
    for u in units:
        if f'{u.name}/conv1+conv2' in layers:
            # fp16 plans of full-width block1/unit_1 (64-channel input): conv1 runs on the 3x3 layer's LDS-resident slab and the
            # projection shortcut inside the conv3 launch -- no conv1 / shortcut / pair layer, no shortcut tensor
            c12, c3 = layers[f'{u.name}/conv1+conv2'], layers[f'{u.name}/conv3']
            assert prec == 'f16' and u.c_in == 64 and u.c_out == 256 and u.stride == 1 and u.rate == 1
            assert not any(f'{u.name}/{k}' in layers for k in ('conv1', 'conv2', 'shortcut', 'shortcut+conv1'))
            # ... and (round 5) the unit's 256-channel sum stays on chip: it only feeds unit 2's conv1 in the same launch
            assert c12.fused_flags == _lib.FUSED_CONV1_IN_FRONT
            assert c3.fused_flags == _lib.FUSED_PROJECTION_SHORTCUT | _lib.FUSED_OUT_ON_CHIP
            assert (c12.kh, c12.stride, c12.dilation, c12.c_in, c12.c_out, c12.h_in, c12.h_out, c12.pad_top, c12.relu) == \
                (3, 1, 1, u.c_in, u.c_bott, u.side_in, u.side_out, 1, 1)
            assert abs(c12.flops_per_image - 2.0 * u.side_in ** 2 * u.c_bott * (9 * u.c_bott + u.c_in)) < 1
            assert (c3.c_in, c3.c_out, c3.has_residual, c3.has_prologue, c3.relu) == (u.c_bott, u.c_out, 0, 0, 0)
            assert abs(c3.flops_per_image - 2.0 * u.side_in ** 2 * (u.c_out * (u.c_bott + u.c_in) + u.c_bott * u.c_out)) < 1
            continue
        c2, c3 = (layers[f'{u.name}/conv{i}'] for i in (2, 3))
        if u.name in fused_into:
            host = fused_into[u.name]
            # block1 (64 -> 256, next conv1 256 -> 64) and block2 (128 -> 512, next conv1 512 -> 128)
            assert prec == 'f16' and f'{u.name}/conv1' not in layers and u.c_in == u.c_out == host.c_out and u.c_out in (256, 512)
            assert (host.out2_channels, host.h_out, host.kh, host.stride) == (u.c_bott, u.side_in, 1, 1)
            assert host.out2_offset >= 0 and host.out2_offset != host.out_offset
            assert (c2.kh, c2.stride, c2.dilation, c2.h_in, c2.h_out, c2.c_out) == (3, u.stride, u.rate, u.side_in, u.side_out, u.c_bott)
            if c3.fused_flags & _lib.FUSED_REBUILT_SHORTCUT:
                # block1/unit_2 (round 5): the identity shortcut is rebuilt in the launch; the info still states the reference's
                # shortcut; where unit 3 is strided only the pixels ITS shortcut reads are written (compact copy), else the sum
                assert u.name == 'block1/unit_2' and (c3.has_residual, c3.res_stride, c3.res_offset) == (1, 1, 0)
                u3 = units[units.index(u) + 1]
                if u3.stride == 2:
                    assert c3.fused_flags & _lib.FUSED_OUT_ON_CHIP and c3.out_sub_offset >= 0
                    assert (c3.out_sub_side, c3.out_sub_off) == (u3.side_out, 1 if u3.centered else 0)
                    assert layers['block1/unit_3/conv3'].fused_flags == _lib.FUSED_COMPACT_SHORTCUT
                else:
                    assert not (c3.fused_flags & _lib.FUSED_OUT_ON_CHIP) and c3.out_sub_offset == -1
                    assert layers['block1/unit_3/conv3'].fused_flags == 0
            continue
        if f'{u.name}/shortcut+conv1' in layers:
            # fp16 plans fuse the projection shortcut and conv1 of a unit (same pre-activated input)
            pair = layers[f'{u.name}/shortcut+conv1']
            assert u.c_in != u.c_out and u.stride == 1 and u.c_out % 256 == 0
            assert (pair.c_in, pair.c_out, pair.h_in, pair.has_prologue, pair.kh) == (u.c_in, u.c_out, u.side_in, 1, 1)
            assert abs(pair.flops_per_image - 2.0 * u.side_in ** 2 * u.c_in * (u.c_out + u.c_bott)) < 1
            assert (c3.c_out, c3.has_residual, c3.res_stride, c3.res_offset) == (u.c_out, 1, 1, 0)
            assert (c2.kh, c2.stride, c2.dilation, c2.c_out) == (3, u.stride, u.rate, u.c_bott)
            continue
        c1 = layers[f'{u.name}/conv1']
        assert (c1.c_in, c1.c_out, c1.h_in, c1.has_prologue, c1.relu) == (u.c_in, u.c_bott, u.side_in, 1, 1)
        assert (c2.kh, c2.stride, c2.dilation, c2.h_in, c2.h_out, c2.c_out) == (3, u.stride, u.rate, u.side_in, u.side_out, u.c_bott)
        k_eff = 3 + 2 * (u.rate - 1)
        pad = tf_same_pads(u.side_in, k_eff, u.stride)[0] if (u.stride == 1 or u.centered) else (k_eff - 1) // 2
        assert c2.pad_top == c2.pad_left == pad
        shift = 1 if (u.centered and u.stride == 2) else 0
        assert (c3.c_out, c3.has_residual, c3.relu, c3.has_prologue) == (u.c_out, 1, 0, 0)
        if u.c_in == u.c_out:
            assert f'{u.name}/shortcut' not in layers
            assert (c3.res_stride, c3.res_offset) == (u.stride, shift)
        else:
            sc = layers[f'{u.name}/shortcut']
            assert (sc.stride, sc.pad_top, sc.has_prologue, sc.c_out) == (u.stride, -shift, 1, u.c_out)
            assert (c3.res_stride, c3.res_offset) == (1, 0)
    lg = layers['logits']
    assert (lg.h_out, lg.c_out, lg.has_prologue) == (256 // stride, 136, 1)
    assert lg.out_dtype == (_lib.METRO_F32 if prec == 'f16' else _lib.METRO_F64)


def test_plan_rejects_bad_specs(lib):
    good = ModelSpec(50, 16, 'h36m').to_c(_lib.METRO_PREC_F16)
    plan = C.c_void_p()
    for field, bad, needle in (('arch', 34, b'arch'), ('stride', 6, b'stride'), ('precision', 7, b'precision'),
                               ('base_width', 12, b'base_width'), ('n_joints_head', 0, b'n_joints_head')):
        cs = _lib.MetroSpec.from_buffer_copy(good)
        setattr(cs, field, bad)
        assert lib.metro_plan_create(C.byref(cs), 4, C.byref(plan)) == -1
        assert needle in lib.metro_last_error()
    assert lib.metro_plan_create(C.byref(good), 0, C.byref(plan)) == -1
    # the one-launch head entry validates the head width itself (its bias comes in 16-byte pieces): no launch, no GPU needed
    odd = ModelSpec(50, 16, 'h36m', depth=2).to_c(_lib.METRO_PREC_F16)              # 2 x 17 = 34 channels
    p = C.c_void_p(256)
    assert lib.metro_head_f16(p, p, C.cast(p, C.POINTER(C.c_float)), p, p, 1, 2048, C.byref(odd), p, None,
                              C.cast(p, C.POINTER(C.c_float)), None) == -1
    assert b'multiple of 4' in lib.metro_last_error()
    with pytest.raises(ValueError):
        ModelSpec(50, 12, 'h36m')
    with pytest.raises(ValueError):
        ModelSpec(50, 16, 'coco')


def test_forward_without_bound_params_fails_loudly(lib):
    eng = Engine(ModelSpec(50, 32, 'h36m', base_width=8), None, 'f16', max_batch=1)
    st = lib.metro_forward(eng._plan, C.c_void_p(256), 1, C.c_void_p(256), C.c_void_p(256), None)
    assert st == _lib.__dict__.get('METRO_ERR_STATE', -4) and b'not bound' in lib.metro_last_error()


def test_missing_library_is_an_error_not_a_fallback(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(_lib.MetroError, match='no CPU fallback'):
        _lib.load()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'metro_pose3d_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.cpp', '.hip', '.h')):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', text, flags=re.M), f
    code = 'import sys; import metro_pose3d_amd, metro_pose3d_amd.engine, metro_pose3d_amd.inference, metro_pose3d_amd.dist; ' \
           'assert not any(m == "oracle" or m.startswith("oracle.") for m in sys.modules)'
    subprocess.check_call([sys.executable, '-c', code], cwd=ROOT)


def test_param_packing_folds_batchnorm():
    spec = ModelSpec(50, 16, 'h36m', base_width=8)
    params = synth.make_params(50, spec.n_head_channels, 8, seed=1)
    eng = Engine(spec, None, 'f64', max_batch=1)
    infos = {pi.name.decode(): pi for pi in eng.param_infos()}
    root = 'MainPart/resnet_v2_50/block1/unit_1/bottleneck_v2'
    g, b, m, v = (params[f'{root}/conv1/BatchNorm/{k}'].astype(np.float64)
                  for k in ('gamma', 'beta', 'moving_mean', 'moving_variance'))
    scale = g / np.sqrt(v + 1e-5)
    w = pack_param(infos['block1/unit_1/conv1/W'], params)
    assert w.shape == (8, 1, 1, 8) and w.dtype == np.float64
    ref = params[f'{root}/conv1/weights'].astype(np.float64)[0, 0].T * scale[:, None]
    assert np.allclose(w[:, 0, 0, :], ref, rtol=0, atol=1e-15)
    assert np.allclose(pack_param(infos['block1/unit_1/conv1/bias'], params), b - m * scale, atol=1e-15)
    # conv3 has its own biases and no BN
    assert np.array_equal(pack_param(infos['block1/unit_1/conv3/bias'], params),
                          params[f'{root}/conv3/biases'].astype(np.float64))
    # f16 plan: stem kernel padded 7x7x3 -> 7x8x4 with zeros
    e16 = Engine(spec, None, 'f16', max_batch=1)
    i16 = {pi.name.decode(): pi for pi in e16.param_infos()}
    ws = pack_param(i16['conv1/W'], params)
    assert ws.shape == (8, 7, 8, 4) and ws.dtype == np.float16
    assert (ws[:, :, 7, :] == 0).all() and (ws[:, :, :, 3] == 0).all()
    assert np.array_equal(ws[:, :, :7, :3], params['MainPart/resnet_v2_50/conv1/weights'].transpose(3, 0, 1, 2).astype(np.float16))
    blob = e16.pack_params(params)
    assert blob.dtype == np.uint8 and blob.size == e16.param_bytes


def test_model_file_roundtrip(tmp_path):
    from metro_pose3d_amd import load_model, save_model
    spec = ModelSpec(101, 8, 'merged', base_width=8)
    params = synth.make_params(101, spec.n_head_channels, 8)
    path = str(tmp_path / 'm.npz')
    save_model(path, spec, params)
    spec2, params2 = load_model(path)
    assert spec2 == spec and sorted(params2) == sorted(params)
    assert all(np.array_equal(params[k], params2[k]) for k in params)
    np.savez(str(tmp_path / 'junk.npz'), a=np.zeros(3))
    with pytest.raises(ValueError):
        load_model(str(tmp_path / 'junk.npz'))


def test_estimate_pose_plans_by_batch_bucket():
    """The drop-in call takes any N (reference placeholder [None,256,256,3], main.py:109-111): engines are planned per bucket so
    that up to 256 crops go through ONE metro_forward (kernel dispatch depends on the batch), more in chunks of 256."""
    from metro_pose3d_amd import inference as INF
    assert [INF.batch_bucket(n) for n in (1, 8, 9, 64, 65, 256, 257, 5000)] == [8, 8, 64, 64, 256, 256, 256, 256]
    assert INF.BATCH_BUCKETS[-1] == 256 and INF.MAX_CACHED_ENGINES >= len(INF.BATCH_BUCKETS)
