#!/usr/bin/env python3
"""The reference's own NETWORK-BUILDING code executed ON NUMBERS, in the build container (the only place /root/reference exists):

    python tests/golden/make_ref_forward.py        # writes tests/golden/ref_forward_v2.npz

make_ref_schedule.py runs the reference's graph construction (architectures.resnet -> resnet_v2_50/101 -> resnet_v2 -> stack_blocks_dense
-> bottleneck -> conv2d_same / max_pool2d_same / subsample / spatial_slice, cut out of their modules with `ast` and executed unmodified)
on a recording tape that computes shapes only.  Here the SAME lines run on a tape whose tensors carry VALUES: every `slim.conv2d`,
`slim.batch_norm`, `slim.max_pool2d`, `array_ops.pad`, slice and `+` the reference issues is evaluated in NumPy fp64, on the weights the
variable scope names at that point (`<scope>/weights`, `<scope>/biases`, `<scope>/BatchNorm/{gamma,beta,moving_mean,moving_variance}`,
`<scope>/{gamma,...}` for a bare batch_norm: tf.contrib.layers' naming, which tests/test_ref_schedule.py already holds the synthetic
parameter sets to) and on the arguments the reference's arg_scopes resolved.

What this pins: the DATAFLOW of the reference as numbers -- which tensor feeds which op with which weights, the explicit pads of
conv2d_same / max_pool2d_same against the 'SAME' / 'VALID' modes of the calls, strides and rates per unit, where the pre-activation
branches off, what the shortcut reads (sub-sampled, shifted by centered_stride), the postnorm and the logits -- for ResNet-50 at strides
4 / 8 / 16 / 32 with and without centered_stride and ResNet-101 at stride 8.  oracle/forward.py (and through it the HIP path) is held to
these outputs in fp64 (tests/test_ref_forward.py).

What stands in for TensorFlow (and is therefore NOT pinned): the five op kernels below, ~60 lines, written from TensorFlow's documented
semantics and independent of oracle/forward.py (NumPy tap loops, no torch):
  conv2d        'SAME': out = ceil(in / stride), pad_total = max((out - 1) stride + k_eff - in, 0), pad_before = pad_total // 2;
                'VALID': no pad; k_eff = k + (k - 1)(rate - 1); cross-correlation, HWIO weights; then the normalizer (inference
                batch norm) or the bias, then the activation -- tf.contrib.layers.conv2d's order
  batch_norm    inference: (x - moving_mean) * gamma / sqrt(moving_variance + epsilon) + beta, then the activation
  max_pool2d    'VALID' windows (the reference pads pool1 explicitly, with zeros); 'SAME' (subsample's 1 x 1 / stride pool): -inf padding
  pad / slice / add / cast
The crops are 64 x 64 (the reference's functions take the size from the tensor; a 256-pixel ResNet-101 in NumPy is minutes).
Stored: per configuration the fp64 logits of two crops (at stride 4: every second row / column of both crops + the FULL map of crop 0,
so that the HIP path's stride-4 poses -- the reference's default test stride, options.py:96 -- can be decoded from it: v2) and (sum, sum |x|) of EVERY op output in tape order with its scope name;
inputs and weights are regenerated from seeds by metro_pose3d_amd/synth.py (data, not source).
"""
from __future__ import annotations

import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, '..', '..'))

import numpy as np

import make_ref_schedule as S
from metro_pose3d_amd import synth

OUT = os.path.join(HERE, 'ref_forward_v2.npz')
SIDE = 64
N_CROPS = 2
IMAGE_SEED = 4242


# ---- the op kernels (NHWC, fp64) ------------------------------------------------------------------------------------------
def same_pads(n, k_eff, stride):
    out = -(-n // stride)
    total = max((out - 1) * stride + k_eff - n, 0)
    return total // 2, total - total // 2


def conv2d_np(x, w, stride, rate, padding):
    kh, kw, ci, co = w.shape
    assert x.shape[3] == ci
    keh, kew = kh + (kh - 1) * (rate - 1), kw + (kw - 1) * (rate - 1)
    if padding == 'SAME':
        (pt, pb), (pl, pr) = same_pads(x.shape[1], keh, stride), same_pads(x.shape[2], kew, stride)
        x = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    else:
        assert padding == 'VALID'
    ho, wo = (x.shape[1] - keh) // stride + 1, (x.shape[2] - kew) // stride + 1
    y = np.zeros((x.shape[0], ho, wo, co), np.float64)
    for i in range(kh):
        for j in range(kw):
            patch = x[:, i * rate: i * rate + (ho - 1) * stride + 1: stride, j * rate: j * rate + (wo - 1) * stride + 1: stride, :]
            y += patch @ w[i, j]
    return y


def batch_norm_np(x, p, prefix, epsilon):
    g, b = p[prefix + '/gamma'].astype(np.float64), p[prefix + '/beta'].astype(np.float64)
    m, v = p[prefix + '/moving_mean'].astype(np.float64), p[prefix + '/moving_variance'].astype(np.float64)
    return (x - m) * (g / np.sqrt(v + epsilon)) + b


def max_pool_np(x, k, stride, padding):
    if padding == 'SAME':          # TensorFlow's SAME max-pool ignores its padding (= -inf); subsample's 1x1 pool needs none
        (pt, pb), (pl, pr) = same_pads(x.shape[1], k, stride), same_pads(x.shape[2], k, stride)
        x = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)), constant_values=-np.inf)
    else:
        assert padding == 'VALID'
    ho, wo = (x.shape[1] - k) // stride + 1, (x.shape[2] - k) // stride + 1
    y = np.full((x.shape[0], ho, wo, x.shape[3]), -np.inf)
    for i in range(k):
        for j in range(k):
            y = np.maximum(y, x[:, i: i + (ho - 1) * stride + 1: stride, j: j + (wo - 1) * stride + 1: stride, :])
    return y


def act(x, name):
    if name is None:
        return x
    assert name == 'relu', name
    return np.maximum(x, 0.0)


class ValueTape(S.Tape):
    """The tape of make_ref_schedule.py whose tensors also carry their value: emit() evaluates the op it records."""

    def __init__(self, params):
        super().__init__()
        self.params = params
        self.values = {}

    def emit(self, op, inputs, out_shape, **attrs):
        out = super().emit(op, inputs, out_shape, **attrs)
        x = [self.values[t.id] for t in inputs]
        p = self.params
        if op == 'conv2d':
            sc = attrs['scope']
            assert attrs['kernel'][0] == attrs['kernel'][1]
            y = conv2d_np(x[0], p[sc + '/weights'].astype(np.float64), attrs['stride'], attrs['rate'], attrs['padding'])
            if attrs['normalizer'] == 'batch_norm':
                y = batch_norm_np(y, p, sc + '/BatchNorm', attrs['normalizer_params']['epsilon'])
            else:
                assert attrs['normalizer'] is None
                if attrs['has_bias']:
                    y = y + p[sc + '/biases'].astype(np.float64)
            y = act(y, attrs['activation'])
        elif op == 'batch_norm':
            y = act(batch_norm_np(x[0], p, attrs['scope'], attrs['epsilon']), attrs['activation'])
        elif op == 'max_pool2d':
            y = max_pool_np(x[0], attrs['kernel'][0], attrs['stride'], attrs['padding'])
        elif op == 'pad':
            y = np.pad(x[0], [tuple(q) for q in attrs['paddings']])
        elif op == 'slice':
            b = attrs['begin']
            y = x[0][b[0]:, b[1]:, b[2]:, b[3]:]
        elif op == 'add':
            y = x[0] + x[1]
        elif op in ('cast', 'softmax_unused'):
            y = x[0]
        else:
            raise AssertionError(op)
        assert list(y.shape[1:]) == list(out_shape[1:]), (op, attrs.get('scope'), y.shape, out_shape)
        self.values[out.id] = y
        return out


def run(arch, stride, centered, n_out, params, images):
    tape = ValueTape(params)
    ar, _, _ = S.make_tape_namespaces(tape, 'NHWC')
    inp = tape.tensor([None, SIDE, SIDE, 3])
    tape.values[inp.id] = images.astype(np.float64)
    tape.scopes.append('MainPart')
    out = ar['resnet'](inp, n_out, stride=stride, centered_stride=centered, resnet_name=f'resnet_v2_{arch}')
    tape.scopes.pop()
    return tape, tape.values[out.id]


def main():
    if not os.path.isdir(S.REF):
        sys.exit('the reference tree is only present in the build container')
    out = {}
    images = synth.make_images(N_CROPS, SIDE, seed=IMAGE_SEED)
    cases = [(50, s, c, 17) for s in (32, 16, 8, 4) for c in (True, False)] + [(101, 8, True, 19)]
    keys = []
    for arch, stride, centered, joints in cases:
        n_out = 8 * joints
        params = synth.make_params(arch, n_out, 64, seed=arch + stride)
        tape, logits = run(arch, stride, centered, n_out, params, images)
        key = f'rn{arch}_s{stride}_{"c" if centered else "n"}_j{joints}'
        scopes, stats = [], []
        for o in tape.ops:
            v = tape.values[o['output']]
            scopes.append(f"{o['op']}:{o.get('scope', '')}".encode())
            stats.append([v.sum(), np.abs(v).sum(), v.size])
        # stride 4: every second heat-map row and column (the op statistics below still cover every element)
        out[f'{key}/logits'] = logits[:, ::2, ::2, :] if stride == 4 else logits
        if stride == 4:
            out[f'{key}/logits_crop0_full'] = logits[:1]
        out[f'{key}/op_names'] = np.array(scopes)
        out[f'{key}/op_stats'] = np.array(stats, np.float64)
        out[f'{key}/meta'] = np.array([arch, stride, int(centered), joints, arch + stride, SIDE, N_CROPS, IMAGE_SEED], np.int64)
        keys.append(key)
        print(key, logits.shape, float(np.abs(logits).max()), len(tape.ops), 'ops')
    out['cases'] = np.array([k.encode() for k in keys])
    out['meta_columns'] = np.array([b'arch', b'stride', b'centered', b'joints', b'param_seed', b'side', b'n_crops', b'image_seed'])
    np.savez_compressed(OUT, **out)
    print(f'wrote {OUT} ({os.path.getsize(OUT) / 1024:.1f} KiB)')


if __name__ == '__main__':
    main()
