"""Known-answer tests KA1..KA12 that pin the CPU oracle.  No GPU.

The reference has no tests of its own for this path, so these analytical cases are what stands
between the oracle and an unnoticed misreading of the reference.
"""
import numpy as np
import pytest
import torch

from oracle import forward as OF
from oracle.spec import (OracleSpec, decode_constants, export_permutation, head_joint_info,
                         output_joint_info, schedule)


def _one_joint_logits(spec, fill=-60.0):
    j = head_joint_info(spec.dataset).n_joints
    s = spec.out_side
    return np.full((1, s, s, spec.depth * j), fill, np.float64), j, s


@pytest.mark.parametrize('stride', [32, 16, 8, 4])
def test_ka1_one_hot_volume(stride):
    """soft-argmax of a one-hot volume at (w,h,d) -> (w/(S-1), h/(S-1), d/7) (tfu.py:474-499)."""
    spec = OracleSpec(stride=stride)
    logits, j, s = _one_joint_logits(spec)
    rng = np.random.default_rng(stride)
    pos = [(rng.integers(s), rng.integers(s), rng.integers(spec.depth)) for _ in range(j)]
    for jj, (w, h, d) in enumerate(pos):
        logits[0, h, w, d * j + jj] = 80.0
    _, c01 = OF.soft_argmax01(torch.from_numpy(logits).permute(0, 3, 1, 2), j, spec.depth)
    exp = np.array([[w * float(np.float32(1) / np.float32(s - 1)), h * float(np.float32(1) / np.float32(s - 1)),
                     d * float(np.float32(1) / np.float32(7))] for w, h, d in pos])
    assert np.abs(c01[0].numpy() - exp).max() < 1e-6


@pytest.mark.parametrize('stride', [32, 16, 8, 4])
def test_ka2_uniform_logits_and_ka4_decode_constants(stride):
    spec = OracleSpec(stride=stride)
    lrc, half = decode_constants(spec)
    assert (lrc, half) == {32: (223, 16), 16: (239, 8), 8: (247, 4), 4: (251, 2)}[stride]   # KA4
    logits, j, s = _one_joint_logits(spec, fill=0.7)
    _, c01 = OF.soft_argmax01(torch.from_numpy(logits).permute(0, 3, 1, 2), j, spec.depth)
    assert np.abs(c01.numpy() - 0.5).max() < 1e-6                     # centre of the volume
    # mm before root-relative: ((0.5*lrc + half) * 2200/256, same, 1100)
    lrc_mm = (0.5 * lrc + half) * 2200.0 / 256
    mm = torch.cat([(c01[..., :2] * lrc + half) * 2200.0 / 256, c01[..., 2:] * 2200.0], -1).numpy()
    assert np.abs(mm[0, :, 0] - lrc_mm).max() < 1e-3 and np.abs(mm[0, :, 2] - 1100).max() < 1e-3
    out = OF.coords01_to_output(spec, c01).numpy()
    assert np.abs(out).max() < 1e-9                                    # all joints coincide with the root


def test_ka3_softmax_shift_invariance():
    spec = OracleSpec(stride=16)
    rng = np.random.default_rng(0)
    lg = rng.standard_normal((2, 16, 16, 136)) * 3
    a = OF.logits_to_output(spec, lg).numpy()
    b = OF.logits_to_output(spec, lg + 123.456).numpy()
    assert np.abs(a - b).max() < 1e-9


def test_ka5_channel_order_is_depth_major():
    """channel c = d*J + j (volumetric.py:231): a spike at channel d*J+j moves only joint j, to depth d."""
    spec = OracleSpec(stride=32)
    logits, j, s = _one_joint_logits(spec, fill=0.0)
    jj, d = 5, 6
    logits[0, 2, 3, d * j + jj] = 50.0
    _, c01 = OF.soft_argmax01(torch.from_numpy(logits).permute(0, 3, 1, 2), j, spec.depth)
    c = c01[0].numpy()
    others = np.delete(c, jj, axis=0)
    assert np.abs(others - 0.5).max() < 1e-6      # fp32 linspace: mean is 0.5 + 2e-8
    assert abs(c[jj, 2] - d / 7) < 1e-6 and abs(c[jj, 0] - 3 / 7) < 1e-6 and abs(c[jj, 1] - 2 / 7) < 1e-6


def test_ka6_padding_rules():
    """TF SAME vs explicit pads (resnet_utils.py:120-135)."""
    assert OF.tf_same_pads(64, 3, 2) == (0, 1)       # centered 3x3/2 on 64: window i covers in[2i..2i+2]
    assert OF.tf_same_pads(16, 3, 1) == (1, 1)
    assert OF.tf_same_pads(16, 5, 1) == (2, 2)       # rate 2
    assert OF.tf_same_pads(64, 17, 1) == (8, 8)      # rate 8
    x = torch.arange(64, dtype=torch.float64).reshape(1, 1, 1, 64).expand(1, 1, 64, 64).contiguous()
    w = torch.zeros(1, 1, 3, 3, dtype=torch.float64)
    w[0, 0, 1, 0] = 1.0                               # picks the LEFT tap of the middle row
    explicit = OF.conv2d_same(x, w, 2, 1, False)     # pads (1,1): left tap of out i = in[2i-1]
    centered = OF.conv2d_same(x, w, 2, 1, True)      # pads (0,1): left tap of out i = in[2i]
    assert explicit.shape[-1] == centered.shape[-1] == 32
    assert explicit[0, 0, 5, :4].tolist() == [0.0, 1.0, 3.0, 5.0]
    assert centered[0, 0, 5, :4].tolist() == [0.0, 2.0, 4.0, 6.0]
    stem = OF.conv2d_same(torch.zeros(1, 3, 256, 256, dtype=torch.float64),
                          torch.zeros(4, 3, 7, 7, dtype=torch.float64), 2, 1, False)
    assert stem.shape[-2:] == (128, 128)              # pads (3,3): (256+6-7)//2+1


def test_ka7_maxpool_pads_with_zero_not_minus_inf():
    x = -torch.rand(1, 2, 8, 8, dtype=torch.float64) - 0.1
    y = OF.max_pool2d_same_zeropad(x)
    assert y.shape[-2:] == (4, 4)
    assert (y[..., 0, :] == 0).all() and (y[..., :, 0] == 0).all()   # windows touching the pad
    assert (y[..., 1:, 1:] < 0).all()


def test_ka8_shifted_shortcut_picks_odd_pixels():
    """identity shortcut of a strided unit: x[1:,1:][::2,::2] when centered, x[::2,::2] otherwise."""
    from oracle.spec import Unit
    x = torch.arange(8 * 8, dtype=torch.float64).reshape(1, 1, 8, 8).expand(1, 4, 8, 8).contiguous()
    p = {}
    for bn in ('preact', 'conv1/BatchNorm', 'conv2/BatchNorm'):
        c = 4 if bn == 'preact' else 1
        p[f'u/{bn}/gamma'] = np.ones(c, np.float32)
        p[f'u/{bn}/beta'] = np.zeros(c, np.float32)
        p[f'u/{bn}/moving_mean'] = np.zeros(c, np.float32)
        p[f'u/{bn}/moving_variance'] = np.ones(c, np.float32)
    p['u/conv1/weights'] = np.zeros((1, 1, 4, 1), np.float32)
    p['u/conv2/weights'] = np.zeros((3, 3, 1, 1), np.float32)
    p['u/conv3/weights'] = np.zeros((1, 1, 1, 4), np.float32)
    p['u/conv3/biases'] = np.zeros(4, np.float32)
    for centered, first in ((True, 9.0), (False, 0.0)):
        u = Unit(1, 3, 4, 4, 1, 2, 1, centered, 8, 4)
        out = OF.bottleneck(x, p, 'u', u, None)
        assert out.shape == (1, 4, 4, 4)
        assert out[0, 0, 0, 0].item() == first            # x[1,1] = 9 vs x[0,0] = 0
        assert out[0, 0, 1, 1].item() == first + 2 * 8 + 2


@pytest.mark.parametrize('arch', [50, 101])
@pytest.mark.parametrize('stride', [32, 16, 8, 4])
def test_ka9_output_side(arch, stride):
    units = schedule(OracleSpec(arch=arch, stride=stride))
    assert units[-1].side_out == 256 // stride
    assert len(units) == (16 if arch == 50 else 33)


def test_ka10_joint_tables():
    """exported names/edges for h36m and merged; root row."""
    h = output_joint_info('h36m')
    assert h.names == 'pelv,rhip,rkne,rank,lhip,lkne,lank,tors,neck,head,htop,lsho,lelb,lwri,rsho,relb,rwri'.split(',')
    assert h.edges == [(10, 9), (9, 8), (8, 11), (11, 12), (12, 13), (8, 14), (14, 15), (15, 16), (8, 7), (7, 0),
                       (0, 4), (4, 5), (5, 6), (0, 1), (1, 2), (2, 3)]
    m = output_joint_info('merged')
    assert m.names == 'neck,nose,pelv,lsho,lelb,lwri,lhip,lkne,lank,rsho,relb,rwri,rhip,rkne,rank,leye,lear,reye,rear'.split(',')
    assert m.edges == [(1, 0), (0, 2), (0, 3), (3, 4), (4, 5), (0, 9), (9, 10), (10, 11), (2, 6), (6, 7), (7, 8),
                       (2, 12), (12, 13), (13, 14), (16, 15), (15, 1), (18, 17), (17, 1)]
    assert head_joint_info('merged').n_joints == 53 and head_joint_info('merged').names[-1] == 'pelv_tdpw'
    assert export_permutation('h36m')[0] == 16       # root (last head joint) goes to output row 0
    # the product's own tables must say the same
    from metro_pose3d_amd.joints import skeleton
    for ds in ('h36m', 'merged', 'many19'):
        sk, oj = skeleton(ds), output_joint_info(ds)
        assert list(sk.names) == oj.names and [tuple(e) for e in sk.edges] == oj.edges
        assert list(sk.permutation) == export_permutation(ds)
        assert list(sk.head_names) == head_joint_info(ds).names
    assert skeleton('h36m').edges_array().dtype == np.int64


def test_ka10_root_row_is_zero_for_h36m():
    spec = OracleSpec(stride=32)
    rng = np.random.default_rng(1)
    out = OF.logits_to_output(spec, rng.standard_normal((2, 8, 8, 136)) * 4).numpy()
    assert (out[:, 0, :] == 0).all() and np.abs(out[:, 1:, :]).min() > 0
    out_m = OF.logits_to_output(OracleSpec(stride=32, dataset='merged'), rng.standard_normal((1, 8, 8, 424)) * 4).numpy()
    assert np.abs(out_m[:, 2, :]).max() > 0          # merged: `pelv` row is NOT the root (root = joint 52)


def test_ka11_identity_batchnorm():
    """gamma=1, beta=0, mean=0, var=1-1e-5 -> pass-through (then ReLU)."""
    x = torch.randn(1, 3, 4, 4, dtype=torch.float64)
    p = {'b/gamma': np.ones(3), 'b/beta': np.zeros(3), 'b/moving_mean': np.zeros(3),
         'b/moving_variance': np.full(3, 1 - 1e-5)}
    assert torch.allclose(OF.batch_norm(x, p, 'b', False), x, atol=1e-15)
    assert torch.allclose(OF.batch_norm(x, p, 'b', True), torch.relu(x), atol=1e-15)


def test_ka12_schedule_table():
    """(stride, rate, centered) per unit for every model stride."""
    def nontrivial(arch, stride):
        return {(u.block, u.unit): (u.stride, u.rate, u.centered and u.stride == 2)
                for u in schedule(OracleSpec(arch=arch, stride=stride)) if (u.stride, u.rate) != (1, 1)}
    assert nontrivial(50, 32) == {(1, 3): (2, 1, False), (2, 4): (2, 1, False), (3, 6): (2, 1, True)}
    s16 = nontrivial(50, 16)
    assert s16 == {(1, 3): (2, 1, False), (2, 4): (2, 1, True), (4, 1): (1, 2, False), (4, 2): (1, 2, False),
                   (4, 3): (1, 2, False)}
    s8 = nontrivial(50, 8)
    assert s8[(1, 3)] == (2, 1, True) and (2, 4) not in s8
    assert all(s8[(3, u)] == (1, 2, False) for u in range(1, 7)) and all(s8[(4, u)] == (1, 4, False) for u in (1, 2, 3))
    s4 = nontrivial(50, 4)
    assert (1, 1) not in s4 and all(s4[(2, u)] == (1, 2, False) for u in range(1, 5))
    assert all(s4[(3, u)] == (1, 4, False) for u in range(1, 7)) and all(s4[(4, u)] == (1, 8, False) for u in (1, 2, 3))
    r101 = nontrivial(101, 8)
    assert all(r101[(3, u)] == (1, 2, False) for u in range(1, 24)) and len(r101) == 1 + 23 + 3
    # quirk: RN101 at stride 4 marks block3's last unit centered (c[-1]) but it runs with stride 1
    u = [u for u in schedule(OracleSpec(arch=101, stride=4)) if (u.block, u.unit) == (3, 23)][0]
    assert u.centered and u.stride == 1
    assert not any(u.centered for u in schedule(OracleSpec(arch=50, stride=4)))
    with pytest.raises(ValueError):
        schedule(OracleSpec(stride=6))


def test_linspace_is_fp32_like_tf():
    """tf.linspace(0., 1., n) in fp32: step = fp32(1/(n-1)); 7*step rounds to exactly 1.0."""
    step = np.float32(1.0) / np.float32(7)
    assert np.float32(7) * step == np.float32(1.0)
