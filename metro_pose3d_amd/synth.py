"""Seeded synthetic model parameters and input crops (there are no released weights offline).

The distributions are the ones SURVEY.md section 8(d) / BASELINE.md section 3 prescribe:
conv kernels N(0, sqrt(2/fan_in)), conv biases N(0, 0.01), BN gamma~U(0.5,1.5), beta~N(0,0.1),
moving_mean~N(0,0.1), moving_variance~U(0.5,1.5); images U[0,1).  Two documented deviations
keep the *synthetic* network in the numeric regime of a trained one (fp16 is the reference's
default compute dtype, options.py:73):
  * the residual-branch output conv (`conv3`) is damped by RES_GAIN so the residual stream
    does not double its variance at every one of the 16 / 33 units (which overflows fp16 in
    ResNet-101; a trained network's branches are small relative to the stream);
  * the `logits` kernel is rescaled by a per-spec constant (LOGIT_GAIN table, measured once
    with the fp64 oracle and rounded to 3 digits) so per-joint logit std is ~4: peaky but not
    one-hot heat-maps, which makes the soft-argmax non-trivial.

Tensors are keyed by TF-slim variable names and stored HWIO / fp32 like a TF checkpoint
(reference scopes: architectures.py:24, volumetric.py:158, resnet_v2.py:117-136,203-236).
Every tensor draws from its own generator seeded by (seed, crc32(name)), so the values do not
depend on generation order.
"""
from __future__ import annotations

import zlib
from typing import Dict, List, Tuple

import numpy as np

RES_GAIN = 0.25

# logits-kernel gain giving per-joint logit std ~= 4 on seed-1234 crops, measured with the fp64
# oracle (tests/golden/make_golden.py --calibrate) and frozen here so the generator is a pure
# function of its arguments on every machine.  Key: (arch, stride, base_width).
LOGIT_GAIN: Dict[Tuple[int, int, int], float] = {
    (50, 32, 64): 1.04, (50, 32, 16): 0.817, (50, 32, 8): 0.84,
    (50, 16, 64): 1.04, (50, 16, 16): 0.826, (50, 16, 8): 0.837,
    (50, 8, 64): 1.04, (50, 8, 16): 0.823, (50, 8, 8): 0.843,
    (50, 4, 64): 1.05, (50, 4, 16): 0.824, (50, 4, 8): 0.844,
    (101, 32, 64): 0.509, (101, 16, 64): 0.509,
    (101, 8, 64): 0.508, (101, 8, 16): 0.41, (101, 8, 8): 0.332,
    (101, 4, 64): 0.509, (101, 4, 16): 0.41, (101, 4, 8): 0.333,
}


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.default_rng([seed, zlib.crc32(name.encode())])


def block_layout(arch: int, base_width: int) -> List[Tuple[int, int]]:
    """[(bottleneck width, units)] of the four blocks (reference resnet_v2.py:282-287,304-309)."""
    return [(base_width, 3), (2 * base_width, 4), (4 * base_width, {50: 6, 101: 23}[arch]),
            (8 * base_width, 3)]


def make_params(arch: int, n_head_channels: int, base_width: int = 64, seed: int = 0,
                logit_gain: float = 1.0, res_gain: float = RES_GAIN) -> Dict[str, np.ndarray]:
    """res_gain: std multiplier of every conv3 (RES_GAIN in all fixtures and in the bench; 1.0 = the undamped He
    initialisation, used by the parity tests as a second, harsher fp16 regime)."""
    root = f'MainPart/resnet_v2_{arch}'
    p: Dict[str, np.ndarray] = {}

    def conv(name, kh, kw, cin, cout, bias, gain=1.0):
        g = _rng(seed, name + '/weights')
        std = np.sqrt(2.0 / (kh * kw * cin)) * gain
        p[name + '/weights'] = (g.standard_normal((kh, kw, cin, cout)) * std).astype(np.float32)
        if bias:
            gb = _rng(seed, name + '/biases')
            p[name + '/biases'] = (gb.standard_normal(cout) * 0.01).astype(np.float32)

    def bn(name, c):
        g = _rng(seed, name)
        p[name + '/gamma'] = g.uniform(0.5, 1.5, c).astype(np.float32)
        p[name + '/beta'] = (g.standard_normal(c) * 0.1).astype(np.float32)
        p[name + '/moving_mean'] = (g.standard_normal(c) * 0.1).astype(np.float32)
        p[name + '/moving_variance'] = g.uniform(0.5, 1.5, c).astype(np.float32)

    conv(root + '/conv1', 7, 7, 3, base_width, bias=True)
    cin = base_width
    for b, (cb, n_units) in enumerate(block_layout(arch, base_width), start=1):
        cout = 4 * cb
        for u in range(1, n_units + 1):
            s = f'{root}/block{b}/unit_{u}/bottleneck_v2'
            bn(s + '/preact', cin)
            if cin != cout:
                conv(s + '/shortcut', 1, 1, cin, cout, bias=True)
            conv(s + '/conv1', 1, 1, cin, cb, bias=False)
            bn(s + '/conv1/BatchNorm', cb)
            conv(s + '/conv2', 3, 3, cb, cb, bias=False)
            bn(s + '/conv2/BatchNorm', cb)
            conv(s + '/conv3', 1, 1, cb, cout, bias=True, gain=res_gain)
            cin = cout
    bn(root + '/postnorm', cin)
    conv(root + '/logits', 1, 1, cin, n_head_channels, bias=True, gain=logit_gain)
    return p


def make_images(n: int, side: int = 256, seed: int = 1234) -> np.ndarray:
    """[n, side, side, 3] fp32 in [0,1): the input contract of reference inference.py:17-18."""
    g = np.random.default_rng([seed, 0x696D67])
    return g.random((n, side, side, 3), dtype=np.float32)


def logit_gain_for(arch: int, stride: int, base_width: int = 64) -> float:
    return LOGIT_GAIN.get((arch, stride, base_width), 1.0)
