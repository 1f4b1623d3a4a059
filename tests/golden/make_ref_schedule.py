#!/usr/bin/env python3
"""The reference's own GRAPH-BUILDING CONTROL FLOW, executed in the build container (the only place /root/reference exists):

    python tests/golden/make_ref_schedule.py        # writes tests/golden/ref_schedule_v1.npz

What runs here are the reference's source lines, cut out of their modules with `ast` (the modules import TensorFlow 1.13 at
the top and cannot be imported) and executed unmodified:

  src/model/architectures.py   resnet_arg_scope (:8-21), resnet (:24-35; its `tfu.in_variable_scope` decorator is dropped)
  src/model/resnet_v2.py       spatial_slice (:70-81), bottleneck (:84-139), resnet_v2 (:142-241), resnet_v2_block (:247-269),
                               resnet_v2_50 (:272-291), resnet_v2_101 (:294-312)
  src/model/resnet_utils.py    Block (:52-62), subsample (:64-79), conv2d_same (:82-135), max_pool2d_same (:138-185),
                               stack_blocks_dense (:263-350)
  src/model/volumetric.py      build_inference_model (:152-216), net_output_to_heatmap_and_coords (:227-235),
                               heatmap_to_image (:288-295), heatmap_to_metric (:303-306)
  src/tfu.py                   softmax (:466-471), decode_heatmap (:474-499), the data-format helpers (:109-125,152-153,
                               174-179,265-360)
  src/tfu3d.py                 root_relative (:23-25)
  src/main.py                  the export permutations (:119-125, literals read through ast)

TensorFlow is ABSENT.  What stands in for it, and therefore what this file does and does not pin:

  PART A (network structure) -- a TAPE.  `slim.conv2d`, `slim.batch_norm`, `slim.max_pool2d`, `array_ops.pad`, tensor slicing
  and `+` are recorders: they note their arguments (after the reference's own arg_scope defaults were merged in by a 30-line
  arg_scope / add_arg_scope), the scope path `variable_scope` gives them, and the ids of the tensors they consume and produce;
  the only thing they COMPUTE is the static output shape (TensorFlow's documented shape rule: SAME -> ceil(in / stride), VALID
  -> (in - k_eff) // stride + 1).  No arithmetic is pinned by part A.  What is: which ops the reference creates, in which
  order, wired to which tensors, with which stride / rate / padding mode / explicit pad amounts / normalizer / activation /
  bias, for every (architecture, stride, centered_stride) -- i.e. KA6, KA8, KA9, KA12 of SURVEY section 8(c) as REFERENCE
  outputs, and the unit table `oracle/spec.schedule` and `csrc/plan.cpp` are each held to (tests/test_ref_schedule.py).

  PART B (decode) -- NumPy under TensorFlow's names.  `tf.reshape / transpose / reduce_max / exp / reduce_sum / linspace / cast /
  squeeze / stack / concat / identity / gather` are bound to the NumPy function of the same meaning and the reference's lines
  run on real numbers (fp32 arrays): which axis is x, y, z, the channel order d*J+j, the linspace weights, the decode
  constants, the root joint and the export permutation are the reference's; the floating-point summation order inside
  `np.sum` / `np.exp` is NumPy's, not TensorFlow's (differences at the 1e-6 relative level).  KA1, KA2, KA4, KA5 as reference
  outputs; oracle/forward.py's soft-argmax (1e-9 mm in fp64) and the HIP soft-argmax (2e-3 mm) are held to the stored poses.

Only inputs and the reference's outputs are stored (arrays and one JSON tape per configuration: data, not source).
"""
from __future__ import annotations

import ast
import collections
import contextlib
import functools
import json
import math
import os
import sys
import types

sys.dont_write_bytecode = True

import numpy as np

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ref_schedule_v1.npz')


# ------------------------------------------------------------------------------------------------------------------------
# ast plumbing
# ------------------------------------------------------------------------------------------------------------------------
def cut(relpath, names, ns, drop_decorators=()):
    """Executes the named top-level functions / classes / assignments of a reference module inside namespace `ns`."""
    path = os.path.join(REF, relpath)
    tree = ast.parse(open(path).read())
    picked = []
    for node in tree.body:
        name = getattr(node, 'name', None)
        if name is None and isinstance(node, ast.Assign) and isinstance(node.targets[0], ast.Name):
            name = node.targets[0].id
        if name in names:
            if name in drop_decorators:
                node.decorator_list = []
            picked.append(node)
    got = sorted({getattr(n, 'name', None) or n.targets[0].id for n in picked})
    assert got == sorted(names), (relpath, sorted(names), got)
    exec(compile(ast.Module(body=picked, type_ignores=[]), path, 'exec'), ns)
    return ns


# ------------------------------------------------------------------------------------------------------------------------
# PART A: the tape
# ------------------------------------------------------------------------------------------------------------------------
class Tape:
    def __init__(self):
        self.ops = []
        self.n_tensors = 0
        self.scopes = []          # variable_scope stack

    def scope_name(self, leaf=None):
        return '/'.join([s for s in self.scopes if s] + ([leaf] if leaf else []))

    def tensor(self, shape):
        t = TapeTensor(self, self.n_tensors, list(shape))
        self.n_tensors += 1
        return t

    def emit(self, op, inputs, out_shape, **attrs):
        out = self.tensor(out_shape)
        self.ops.append(dict(op=op, inputs=[t.id for t in inputs], output=out.id, in_shape=list(inputs[0].shape),
                             out_shape=list(out_shape), **attrs))
        return out


class _Shape:
    def __init__(self, dims):
        self.dims = list(dims)
        self.ndims = len(self.dims)

    def as_list(self):
        return list(self.dims)


class TapeTensor:
    """NHWC static shape [None, H, W, C] + an id; supports what the reference does to tensors outside of TF ops:
    `inp[indices]` (spatial_slice, resnet_v2.py:70-81) and `shortcut + residual` (resnet_v2.py:138)."""

    def __init__(self, tape, tid, shape):
        self.tape, self.id, self.shape = tape, tid, shape

    def get_shape(self):
        return _Shape(self.shape)

    def __getitem__(self, indices):
        assert isinstance(indices, (list, tuple)) and len(indices) == 4
        begin, out_shape = [], []
        for dim, sl in zip(self.shape, indices):
            assert isinstance(sl, slice) and sl.stop is None and sl.step is None
            b = sl.start or 0
            begin.append(b)
            out_shape.append(None if dim is None else dim - b)
        return self.tape.emit('slice', [self], out_shape, begin=begin, scope=self.tape.scope_name())

    def __add__(self, other):
        assert self.shape == other.shape, (self.shape, other.shape)       # TF would fail to build the Add otherwise
        return self.tape.emit('add', [self, other], self.shape, scope=self.tape.scope_name())


class ArgScopes:
    """tf.contrib.framework arg_scope / add_arg_scope, the part the reference uses: a stack of {function: default kwargs};
    a decorated function called inside `with arg_scope([f], **kw)` receives kw unless the call site passes the argument."""

    def __init__(self):
        self.stack = [{}]

    @contextlib.contextmanager
    def arg_scope(self, list_ops_or_scope, **kwargs):
        if isinstance(list_ops_or_scope, dict):                 # `with slim.arg_scope(resnet_arg_scope()):` re-enters a scope
            assert not kwargs
            new = {k: dict(v) for k, v in list_ops_or_scope.items()}
        else:
            new = {k: dict(v) for k, v in self.stack[-1].items()}
            for f in list_ops_or_scope:
                key = getattr(f, '_arg_scope_key', None)
                assert key is not None, f'{f} is not decorated with add_arg_scope'
                new.setdefault(key, {}).update(kwargs)
        self.stack.append(new)
        try:
            yield new
        finally:
            self.stack.pop()

    def add_arg_scope(self, fn):
        key = f'{fn.__name__}#{id(fn)}'

        @functools.wraps(fn)
        def wrapper(*args, **kwargs):
            merged = dict(self.stack[-1].get(key, {}))
            merged.update(kwargs)
            return fn(*args, **merged)
        wrapper._arg_scope_key = key
        return wrapper


def _same_out(n, stride):
    return -(-n // stride)


def _valid_out(n, k_eff, stride):
    return (n - k_eff) // stride + 1


def make_tape_namespaces(tape: Tape, data_format='NHWC'):
    """Namespaces in which the reference's model-building functions run, with TensorFlow's entry points as recorders."""
    A = ArgScopes()
    hw = (1, 2) if data_format == 'NHWC' else (2, 3)
    ch = 3 if data_format == 'NHWC' else 1

    @contextlib.contextmanager
    def variable_scope(name_or_scope, default_name=None, values=None, reuse=None):
        name = name_or_scope if name_or_scope is not None else default_name
        tape.scopes.append(name)
        try:
            yield types.SimpleNamespace(name=tape.scope_name(), original_name_scope=tape.scope_name() + '/')
        finally:
            tape.scopes.pop()

    def _k2(k):
        return [k, k] if isinstance(k, int) else list(k)

    @A.add_arg_scope
    def conv2d(inputs, num_outputs, kernel_size, stride=1, padding='SAME', data_format=None, rate=1, activation_fn='DEFAULT_RELU',
               normalizer_fn=None, normalizer_params=None, weights_initializer=None, weights_regularizer=None,
               biases_initializer='DEFAULT_ZEROS', reuse=None, outputs_collections=None, scope=None):
        kh, kw = _k2(kernel_size)
        shp = list(inputs.shape)
        for ax, k in zip(hw, (kh, kw)):
            k_eff = k + (k - 1) * (rate - 1)
            shp[ax] = _same_out(shp[ax], stride) if padding == 'SAME' else _valid_out(shp[ax], k_eff, stride)
        shp[ch] = num_outputs
        # tf.contrib.layers.conv2d: a bias is created iff there is no normalizer_fn (and biases_initializer is not None)
        return tape.emit('conv2d', [inputs], shp, scope=tape.scope_name(scope), kernel=[kh, kw], stride=stride, rate=rate,
                         padding=padding, c_in=inputs.shape[ch], c_out=num_outputs,
                         normalizer='batch_norm' if normalizer_fn is batch_norm else None if normalizer_fn is None else str(normalizer_fn),
                         normalizer_params={k: v for k, v in (normalizer_params or {}).items() if k in ('epsilon', 'scale', 'is_training', 'fused', 'decay')},
                         activation=None if activation_fn is None else getattr(activation_fn, '__name__', str(activation_fn)),
                         has_bias=normalizer_fn is None and biases_initializer is not None)

    @A.add_arg_scope
    def batch_norm(inputs, decay=0.999, center=True, scale=False, epsilon=0.001, activation_fn=None, is_training=True, fused=None,
                   data_format='NHWC', outputs_collections=None, scope=None, reuse=None):
        return tape.emit('batch_norm', [inputs], inputs.shape, scope=tape.scope_name(scope), epsilon=epsilon, scale=scale,
                         center=center, is_training=is_training,
                         activation=None if activation_fn is None else getattr(activation_fn, '__name__', str(activation_fn)))

    @A.add_arg_scope
    def max_pool2d(inputs, kernel_size, stride=2, padding='VALID', data_format='NHWC', outputs_collections=None, scope=None):
        kh, kw = _k2(kernel_size)
        shp = list(inputs.shape)
        for ax, k in zip(hw, (kh, kw)):
            shp[ax] = _same_out(shp[ax], stride) if padding == 'SAME' else _valid_out(shp[ax], k, stride)
        return tape.emit('max_pool2d', [inputs], shp, scope=tape.scope_name(scope), kernel=[kh, kw], stride=stride, padding=padding)

    @A.add_arg_scope
    def conv3d(*a, **k):
        raise AssertionError('not on the path')

    def pad(inputs, paddings):
        shp = [None if d is None else d + p[0] + p[1] for d, p in zip(inputs.shape, paddings)]
        return tape.emit('pad', [inputs], shp, paddings=[list(p) for p in paddings], scope=tape.scope_name())

    def relu(x):
        raise AssertionError('activation functions are only passed around')
    relu.__name__ = 'relu'

    def cast(x, dtype):
        return tape.emit('cast', [x], x.shape, dtype=str(dtype), scope=tape.scope_name())

    def softmax(x, scope=None):          # end_points['predictions'] (resnet_v2.py:240): built, never fetched by the export
        return tape.emit('softmax_unused', [x], x.shape, scope=tape.scope_name(scope))

    layers = types.SimpleNamespace(conv2d=conv2d, batch_norm=batch_norm, max_pool2d=max_pool2d, conv3d=conv3d, softmax=softmax,
                                   l2_regularizer=lambda s: ('l2', s), variance_scaling_initializer=lambda: 'variance_scaling',
                                   arg_scope=A.arg_scope)
    tfu = types.SimpleNamespace(
        data_format=lambda: data_format, image_axes=lambda: hw, channel_axis=lambda: ch, is_training=lambda: False,
        static_shape=lambda t: t.get_shape().as_list(), static_n_channels=lambda t: t.get_shape().as_list()[ch],
        get_dtype=lambda: 'float16')
    utils = types.SimpleNamespace(collect_named_outputs=lambda coll, name, out: out, convert_collection_to_dict=lambda c: {})

    ru = {'collections': collections, 'layers_lib': layers, 'layers': layers, 'add_arg_scope': A.add_arg_scope,
          'arg_scope': A.arg_scope, 'utils': utils, 'array_ops': types.SimpleNamespace(pad=pad),
          'variable_scope': types.SimpleNamespace(variable_scope=variable_scope), 'tfu': tfu}
    cut('src/model/resnet_utils.py', ['Block', 'subsample', 'conv2d_same', 'max_pool2d_same', 'stack_blocks_dense'], ru)
    resnet_utils = types.SimpleNamespace(**{k: ru[k] for k in ('Block', 'subsample', 'conv2d_same', 'max_pool2d_same', 'stack_blocks_dense')})

    rv = {'np': np, 'slim': layers, 'layers_lib': layers, 'layers': layers, 'add_arg_scope': A.add_arg_scope, 'arg_scope': A.arg_scope,
          'utils': utils, 'math_ops': types.SimpleNamespace(), 'nn_ops': types.SimpleNamespace(relu=relu),
          'variable_scope': types.SimpleNamespace(variable_scope=variable_scope), 'tfu': tfu, 'resnet_utils': resnet_utils}
    cut('src/model/resnet_v2.py', ['spatial_slice', 'bottleneck', 'resnet_v2', 'resnet_v2_block', 'resnet_v2_50', 'resnet_v2_101'], rv)

    ar = {'tf': types.SimpleNamespace(cast=cast, nn=types.SimpleNamespace(relu=relu), float32='float32'), 'slim': layers, 'tfu': tfu,
          'model': types.SimpleNamespace(resnet_v2=types.SimpleNamespace(resnet_v2_50=rv['resnet_v2_50'], resnet_v2_101=rv['resnet_v2_101']))}
    cut('src/model/architectures.py', ['resnet_arg_scope', 'resnet'], ar, drop_decorators=('resnet',))
    return ar, rv, ru


def run_tape(arch, stride, centered, n_out, data_format='NHWC'):
    tape = Tape()
    ar, _, _ = make_tape_namespaces(tape, data_format)
    side = 256
    inp = tape.tensor([None, side, side, 3] if data_format == 'NHWC' else [None, 3, side, side])
    with contextlib.ExitStack() as st:
        # `@tfu.in_variable_scope('Resnet', ...)` + scope='MainPart' (volumetric.py:158) name the outer scopes; they carry no logic
        tape.scopes.append('MainPart')
        out = ar['resnet'](inp, n_out, stride=stride, centered_stride=centered, resnet_name=f'resnet_v2_{arch}')
        tape.scopes.pop()
    return tape, inp, out


def unit_table(tape: Tape, data_format='NHWC'):
    """Reads the per-unit facts off the tape (no model knowledge beyond the scope names the reference itself gives)."""
    hax = 1 if data_format == 'NHWC' else 2
    by_out = {o['output']: o for o in tape.ops}
    rows, names = [], []
    adds = [o for o in tape.ops if o['op'] == 'add']
    for add in adds:
        unit_scope = add['scope']                                  # .../blockB/unit_U/bottleneck_v2
        ops = [o for o in tape.ops if o.get('scope', '').startswith(unit_scope)]
        get = lambda leaf, kind: next(o for o in ops if o['scope'] == f'{unit_scope}/{leaf}' and o['op'] == kind)
        preact, c1, c2, c3 = get('preact', 'batch_norm'), get('conv1', 'conv2d'), get('conv2', 'conv2d'), get('conv3', 'conv2d')
        unit_in = preact['inputs'][0]
        assert c1['inputs'] == [preact['output']] and c3['output'] == add['inputs'][1]
        # conv2's input chain: conv1 -> [pad] -> conv2
        src = by_out[c2['inputs'][0]]
        pad_beg = pad_end = 0
        if src['op'] == 'pad':
            pad_beg, pad_end = src['paddings'][hax]
            assert src['paddings'][hax + 1] == [pad_beg, pad_end] and src['inputs'] == [c1['output']]
        else:
            assert src is c1
        # the shortcut chain ends in add.inputs[0]: identity -> [slice] -> [max_pool 1x1] ; projection -> [slice] -> conv
        sc = by_out.get(add['inputs'][0])
        sc_kind, sc_stride, shift, sc_from = 0, 1, 0, -1        # kind 0 identity, 1 projection; from 0 = unit input, 1 = preact
        node = sc
        if node is not None and node['op'] == 'conv2d' and node['scope'] == f'{unit_scope}/shortcut':
            sc_kind, sc_stride = 1, node['stride']
            assert node['kernel'] == [1, 1] and node['has_bias'] and node['activation'] is None and node['normalizer'] is None
            node = by_out.get(node['inputs'][0])
        elif node is not None and node['op'] == 'max_pool2d' and node['scope'] == f'{unit_scope}/shortcut':
            assert node['kernel'] == [1, 1]
            sc_stride = node['stride']
            node = by_out.get(node['inputs'][0])
        if node is not None and node['op'] == 'slice' and node['scope'] == unit_scope:
            shift = node['begin'][hax]
            assert node['begin'][hax + 1] == shift
            node = by_out.get(node['inputs'][0])
        end_id = node['output'] if node is not None else add['inputs'][0]
        if sc is None or (sc['op'] not in ('conv2d', 'max_pool2d', 'slice')) or not sc['scope'].startswith(unit_scope):
            end_id = add['inputs'][0]
        sc_from = 1 if end_id == preact['output'] else 0
        assert end_id in (preact['output'], unit_in), (unit_scope, end_id)
        b, u = unit_scope.split('/')[-3:-1]
        names.append(f'{b}/{u}')
        rows.append([int(b[5:]), int(u[5:]), c1['c_in'], c3['c_out'], c1['c_out'], c2['stride'], c2['rate'],
                     1 if c2['padding'] == 'SAME' else 0, pad_beg, pad_end, sc_kind, sc_stride, shift, sc_from,
                     c1['in_shape'][hax], add['out_shape'][hax],
                     int(c1['has_bias']), int(c2['has_bias']), int(c3['has_bias']),
                     int(c1['normalizer'] == 'batch_norm' and c1['activation'] == 'relu'),
                     int(c2['normalizer'] == 'batch_norm' and c2['activation'] == 'relu'),
                     int(c3['normalizer'] is None and c3['activation'] is None)])
    return names, np.array(rows, np.int32)


UNIT_COLUMNS = ['block', 'unit', 'c_in', 'c_out', 'c_bott', 'conv2_stride', 'conv2_rate', 'conv2_padding_same', 'conv2_pad_beg',
                'conv2_pad_end', 'shortcut_is_projection', 'shortcut_stride', 'shortcut_shift', 'shortcut_from_preact', 'side_in',
                'side_out', 'conv1_bias', 'conv2_bias', 'conv3_bias', 'conv1_bn_relu', 'conv2_bn_relu', 'conv3_linear']


def stem_table(tape: Tape, data_format='NHWC'):
    hax = 1 if data_format == 'NHWC' else 2
    by_out = {o['output']: o for o in tape.ops}
    conv1 = next(o for o in tape.ops if o['op'] == 'conv2d' and o['scope'].endswith('/conv1') and 'block' not in o['scope'])
    pool1 = next(o for o in tape.ops if o['op'] == 'max_pool2d' and o['scope'].endswith('/pool1'))
    postnorm = next(o for o in tape.ops if o['op'] == 'batch_norm' and o['scope'].endswith('/postnorm'))
    logits = next(o for o in tape.ops if o['op'] == 'conv2d' and o['scope'].endswith('/logits'))
    cpad, ppad = by_out[conv1['inputs'][0]], by_out[pool1['inputs'][0]]
    assert cpad['op'] == 'pad' and ppad['op'] == 'pad' and ppad['inputs'] == [conv1['output']]
    assert logits['inputs'] == [postnorm['output']]
    first_cast = tape.ops[0]
    last_cast = tape.ops[-1]
    assert first_cast['op'] == 'cast' and last_cast['op'] == 'cast' and last_cast['inputs'] == [logits['output']]
    return dict(
        conv1=[conv1['kernel'][0], conv1['stride'], int(conv1['padding'] == 'SAME'), *cpad['paddings'][hax], int(conv1['has_bias']),
               int(conv1['normalizer'] is None and conv1['activation'] is None), conv1['c_out'], conv1['out_shape'][hax]],
        pool1=[pool1['kernel'][0], pool1['stride'], int(pool1['padding'] == 'SAME'), *ppad['paddings'][hax], pool1['out_shape'][hax]],
        postnorm=[int(postnorm['activation'] == 'relu'), int(postnorm['scale']), postnorm['epsilon'], int(postnorm['is_training'])],
        logits=[logits['kernel'][0], logits['stride'], int(logits['has_bias']), int(logits['normalizer'] is None and logits['activation'] is None),
                logits['c_in'], logits['c_out'], logits['out_shape'][hax]],
        casts=[first_cast['dtype'], last_cast['dtype']])


# ------------------------------------------------------------------------------------------------------------------------
# PART B: the decode on NumPy
# ------------------------------------------------------------------------------------------------------------------------
class T(np.ndarray):
    """ndarray with the two TensorFlow methods the reference calls on tensors (get_shape().as_list() / .ndims)."""

    def get_shape(self):
        return _Shape(self.shape)


def as_t(x):
    return np.asarray(x).view(T)


def _ax(axis):
    return tuple(axis) if isinstance(axis, (list, tuple)) else axis


def numpy_tf():
    def linspace(start, stop, num):
        # tf.linspace on float32: start + i * ((stop - start) / (num - 1)), evaluated in float32 (TF's LinSpace kernel)
        step = np.float32((np.float32(stop) - np.float32(start)) / np.float32(num - 1))
        return as_t(np.float32(start) + np.arange(num, dtype=np.float32) * step)

    return types.SimpleNamespace(
        Tensor=T, float32=np.float32,
        identity=lambda x, name=None: x,
        reshape=lambda x, shape: as_t(np.reshape(x, shape)),
        transpose=lambda x, perm: as_t(np.transpose(x, perm)),
        reduce_max=lambda x, axis=None, keepdims=False: as_t(np.max(x, axis=_ax(axis), keepdims=keepdims)),
        reduce_sum=lambda x, axis=None, keepdims=False: as_t(np.sum(x, axis=_ax(axis), keepdims=keepdims)),
        exp=lambda x: as_t(np.exp(x)),
        linspace=linspace,
        cast=lambda x, dtype: as_t(np.asarray(x).astype(dtype)),
        squeeze=lambda x, axis=None: as_t(np.squeeze(x, axis=_ax(axis))),
        stack=lambda xs, axis=0: as_t(np.stack(xs, axis=axis)),
        concat=lambda xs, axis: as_t(np.concatenate(xs, axis=axis)),
        gather=lambda x, idx, axis=0, name=None: as_t(np.take(x, idx, axis=axis)),
        name_scope=lambda *a, **k: contextlib.nullcontext())


class AttrDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def export_permutations():
    tree = ast.parse(open(os.path.join(REF, 'src/main.py')).read())
    export = next(n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == 'export')
    perms = {}
    for node in ast.walk(export):
        if isinstance(node, ast.If) and isinstance(node.test, ast.Compare) and isinstance(node.test.comparators[0], ast.Constant):
            for s in node.body:
                if isinstance(s, ast.Assign) and getattr(s.targets[0], 'id', None) == 'permutation':
                    perms[node.test.comparators[0].value] = ast.literal_eval(s.value)
    return perms


def run_decode(logits_nhwc, stride, n_joints, centered, data_format, permutation):
    """build_inference_model (volumetric.py:152-216) on given network outputs; returns what it computed + the resnet() call."""
    tf = numpy_tf()
    calls = []
    tfu_ns = {'tf': tf, 'np': np, '_DATA_FORMAT': data_format}
    cut('src/tfu.py', ['softmax', 'decode_heatmap', 'static_shape', 'static_n_channels', 'static_image_shape', 'data_format', 'channel_axis',
                       'image_axes', 'nhwc_to_nchw', 'nchw_to_nhwc', 'convert_data_format', 'nhwc_to_std', 'std_to_nchw', 'TRAIN', 'TEST'], tfu_ns)
    tfu = types.SimpleNamespace(**{k: v for k, v in tfu_ns.items() if not k.startswith('__')})
    tfu3d_ns = {'tf': tf}
    cut('src/tfu3d.py', ['root_relative'], tfu3d_ns)
    FLAGS = types.SimpleNamespace(stride_train=stride, stride_test=stride, depth=8, centered_stride=centered, architecture='resnet_v2_50',
                                  bone_length_dataset=None, scale_recovery='metro', proc_side=256, box_size_mm=2200)

    def resnet(im, n_out, scope=None, reuse=None, stride=None, centered_stride=None, resnet_name=None):
        calls.append(dict(n_out=n_out, scope=scope, stride=stride, centered_stride=centered_stride, resnet_name=resnet_name))
        return tfu.nhwc_to_std(as_t(logits_nhwc))            # the network's output in the graph's data format

    vol = {'tf': tf, 'tfu': tfu, 'tfu3d': types.SimpleNamespace(root_relative=tfu3d_ns['root_relative']), 'FLAGS': FLAGS,
           'TRAIN': tfu_ns['TRAIN'], 'model': types.SimpleNamespace(architectures=types.SimpleNamespace(resnet=resnet)),
           'data': types.SimpleNamespace(datasets=types.SimpleNamespace(current_dataset=lambda: None))}
    cut('src/model/volumetric.py', ['build_inference_model', 'net_output_to_heatmap_and_coords', 'heatmap_to_image', 'heatmap_to_metric'], vol)
    t = AttrDict()
    t.x = as_t(np.zeros((logits_nhwc.shape[0], 256, 256, 3), np.float32))          # main.py:109-111 (placeholder, already "std")
    vol['build_inference_model'](types.SimpleNamespace(n_joints=n_joints), tfu_ns['TEST'], t)
    coords01 = vol['net_output_to_heatmap_and_coords'](tfu.nhwc_to_std(as_t(logits_nhwc)), types.SimpleNamespace(n_joints=n_joints))[1]
    out = tf.gather(t.coords3d_pred_rootrel, permutation, axis=1, name='output')    # main.py:127
    return dict(softmaxed=np.asarray(t.softmaxed), heatmap_pred_z=np.asarray(t.heatmap_pred_z), coords01=np.asarray(coords01),
                coords3d_pred=np.asarray(t.coords3d_pred), rootrel=np.asarray(t.coords3d_pred_rootrel), output=np.asarray(out)), calls


def decode_constants(stride, centered):
    """heatmap_to_image (volumetric.py:288-295) applied to 0 and 1: returns (offset, offset + last_receptive_center)."""
    vol = {'FLAGS': types.SimpleNamespace(stride_train=stride, stride_test=stride, proc_side=256, centered_stride=centered, box_size_mm=2200),
           'TRAIN': 0, 'tf': numpy_tf()}
    cut('src/model/volumetric.py', ['heatmap_to_image', 'heatmap_to_metric'], vol)
    lo, hi = (vol['heatmap_to_image'](np.float64(v), 2) for v in (0.0, 1.0))
    mm = np.asarray(vol['heatmap_to_metric'](as_t(np.array([[[0.0, 1.0, 0.5]]], np.float32)), 2))
    return float(lo), float(hi), mm


# ------------------------------------------------------------------------------------------------------------------------
def main():
    if not os.path.isdir(REF):
        sys.exit('the reference tree is only present in the build container')
    out = {}
    perms = export_permutations()
    # ---- PART A
    configs = []
    for arch in (50, 101):
        for stride in (4, 8, 16, 32):
            for centered in (True, False):
                key = f'rn{arch}_s{stride}_{"c" if centered else "n"}'
                tape, inp, net = run_tape(arch, stride, centered, n_out=8 * 17)
                tape2, _, net2 = run_tape(arch, stride, centered, n_out=8 * 17, data_format='NCHW')     # the reference's default (options.py:92)
                names, rows = unit_table(tape)
                names2, rows2 = unit_table(tape2, 'NCHW')
                assert names == names2 and (rows == rows2).all() and net.shape[1] == net2.shape[2] == 256 // stride
                stem = stem_table(tape)
                assert stem == stem_table(tape2, 'NCHW')
                out[f'{key}/unit_names'] = np.array([n.encode() for n in names])
                out[f'{key}/units'] = rows
                for part in ('conv1', 'pool1', 'logits'):
                    out[f'{key}/{part}'] = np.array(stem[part], np.int32)
                out[f'{key}/postnorm'] = np.array(stem['postnorm'], np.float64)
                out[f'{key}/casts'] = np.array([c.encode() for c in stem['casts']])
                out[f'{key}/out_shape'] = np.array([-1 if d is None else d for d in net.shape], np.int32)
                out[f'{key}/tape_json'] = np.array(json.dumps(tape.ops, separators=(',', ':')).encode())
                lo, hi, mm = decode_constants(stride, centered)
                out[f'{key}/decode'] = np.array([lo, hi], np.float64)          # offset, offset + last_receptive_center
                out[f'{key}/decode_mm_of_0_1_half'] = mm
                configs.append(key)
    out['configs'] = np.array([c.encode() for c in configs])
    out['unit_columns'] = np.array([c.encode() for c in UNIT_COLUMNS])
    out['conv1_columns'] = np.array([b'kernel', b'stride', b'padding_same', b'pad_beg', b'pad_end', b'has_bias', b'linear', b'c_out', b'side_out'])
    out['pool1_columns'] = np.array([b'kernel', b'stride', b'padding_same', b'pad_beg', b'pad_end', b'side_out'])
    out['logits_columns'] = np.array([b'kernel', b'stride', b'has_bias', b'linear', b'c_in', b'c_out', b'side_out'])
    out['postnorm_columns'] = np.array([b'relu', b'scale', b'epsilon', b'is_training'])
    # ---- PART B
    rng = np.random.default_rng(20260930)
    cases = []
    for stride, jn, ds in ((32, 17, 'h36m'), (16, 17, 'h36m'), (8, 19, 'many19'), (4, 17, 'h36m'), (16, 53, 'merged')):
        for centered in (True, False):
            s = 256 // stride
            n = 3
            logits = (rng.integers(-40, 41, (n, s, s, 8 * jn)) / 4.0).astype(np.float32)   # multiples of 1/4 in [-10, 10]: compressible
            # image 1: one-hot-like peaks at known voxels (KA1 / KA5): joint j at (w, h, d) = ((3j+1) % S, (5j+2) % S, j % 8)
            logits[1] = -30.0
            for j in range(jn):
                logits[1, (5 * j + 2) % s, (3 * j + 1) % s, (j % 8) * jn + j] = 30.0
            logits[2] = 0.25                                                         # uniform volume (KA2)
            # 'many19': a 19-joint head exported with the `merged` permutation (README.md:27-28); 'merged': 19 of the 53 head joints
            perm = perms['merged' if ds == 'many19' else ds]
            # The reference's lines take the dtype of their input (decode_heatmap casts its fp32 linspace to inp.dtype, tfu.py:482):
            # they are run on the fp32 logits widened to fp64, so that NumPy's naive fp32 accumulation over non-contiguous axes
            # (4e-6 relative at S = 32: 8e-3 mm) does not become the fixture's noise floor.  The graph itself runs them in fp32.
            a, calls = run_decode(logits.astype(np.float64), stride, jn, centered, 'NHWC', perm)
            b, calls_b = run_decode(logits.astype(np.float64), stride, jn, centered, 'NCHW', perm)
            f32, _ = run_decode(logits, stride, jn, centered, 'NCHW', perm)
            assert np.abs(f32['output'] - a['output']).max() < 3e-2 and f32['output'].dtype == np.float32
            assert calls == calls_b and all(np.array_equal(a[k], b[k]) for k in a), 'data format must not matter'
            assert calls[0] == dict(n_out=8 * jn, scope='MainPart', stride=stride, centered_stride=centered, resnet_name='resnet_v2_50')
            key = f'decode/s{stride}_{ds}_{"c" if centered else "n"}'
            out[f'{key}/logits'] = logits
            out[f'{key}/permutation'] = np.array(perm, np.int64)
            for k in ('coords01', 'coords3d_pred', 'rootrel', 'output', 'heatmap_pred_z'):
                out[f'{key}/{k}'] = a[k]
            cases.append(key)
    out['decode_cases'] = np.array([c.encode() for c in cases])
    out['versions'] = np.array([np.__version__])
    np.savez_compressed(OUT, **out)
    print(f'wrote {OUT} ({os.path.getsize(OUT) / 1024:.1f} KiB): {len(configs)} schedules, {len(cases)} decode cases')


if __name__ == '__main__':
    main()
