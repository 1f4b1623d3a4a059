"""Single-kernel parity through the C ABI (`-m gpu`): HIP kernels vs fp64 references built from
the oracle's padding / pooling / soft-argmax restatements on the same seeded inputs."""
import ctypes as C
import zlib

import numpy as np
import pytest
import torch

from metro_pose3d_amd import ModelSpec, _lib
from metro_pose3d_amd._lib import check
from tests import helpers as H

pytestmark = pytest.mark.gpu

# (name, n, h_in, c_in, c_out, k, stride, dil, pad, h_out)
# pads follow reference resnet_utils.py:120-135: SAME (rate r -> pad r; centered stride-2 -> 0)
# or explicit (k_eff-1)//2.
CONV_CASES = [
    ('1x1', 2, 16, 64, 256, 1, 1, 1, 0, 16),
    ('1x1_ragged_m', 3, 7, 64, 64, 1, 1, 1, 0, 7),
    # persistent pipelined kernel (conv_pw64.hip; prologue / residual variants): ragged last tile, and more
    # tiles than resident blocks so every block walks several tiles through its double buffers
    ('1x1_pw64_ragged', 3, 7, 64, 256, 1, 1, 1, 0, 7),
    ('1x1_pw64_many_tiles', 25, 64, 64, 256, 1, 1, 1, 0, 64),
    ('1x1_pw128_halves_ragged', 3, 7, 128, 512, 1, 1, 1, 0, 7),
    ('1x1_pw128_many_tiles', 25, 32, 128, 512, 1, 1, 1, 0, 32),
    ('1x1_pw256_slabs_ragged', 3, 7, 256, 1024, 1, 1, 1, 0, 7),
    ('1x1_pw256_many_tiles', 40, 16, 256, 1024, 1, 1, 1, 0, 16),
    ('1x1_pw512_slabs_ragged', 3, 7, 512, 2048, 1, 1, 1, 0, 7),
    ('1x1_pw512_many_tiles', 40, 16, 512, 2048, 1, 1, 1, 0, 16),
    ('1x1_head136', 2, 16, 128, 136, 1, 1, 1, 0, 16),
    ('1x1_cin_tail', 2, 8, 24, 64, 1, 1, 1, 0, 8),
    ('3x3_s1', 2, 16, 64, 64, 3, 1, 1, 1, 16),
    ('3x3_rate2', 2, 16, 64, 128, 3, 1, 2, 2, 16),
    ('3x3_rate4', 1, 32, 32, 64, 3, 1, 4, 4, 32),
    ('3x3_rate8', 1, 16, 32, 64, 3, 1, 8, 8, 16),
    ('3x3_s2_explicit', 2, 32, 64, 64, 3, 2, 1, 1, 16),
    ('3x3_s2_centered', 2, 32, 64, 64, 3, 2, 1, 0, 16),
    ('3x3_big', 4, 32, 128, 256, 3, 1, 1, 1, 32),
    # stride-2 conv2 of a block's last unit: the real shapes of block1 / block2 with both pad rules (explicit 1 / centered 0).
    # (generic ring kernel: the tap-reuse kernel is stride 1 only)
    ('3x3_s2_slab_block1', 3, 64, 64, 64, 3, 2, 1, 1, 32),
    ('3x3_s2_slab_block1_centered', 3, 64, 64, 64, 3, 2, 1, 0, 32),
    ('3x3_s2_slab_block2', 5, 32, 128, 128, 3, 2, 1, 1, 16),
    ('3x3_s2_slab_block2_centered', 5, 32, 128, 128, 3, 2, 1, 0, 16),
    ('1x1_s2_shifted', 2, 16, 64, 128, 1, 2, 1, -1, 8),
    # tap-reuse slab kernel: tiles spanning several 8x8 images (ragged last tile), 64-wide maps,
    # c_in tail + c_out not a multiple of the tile, rate 2 on 16x16 maps (the stride-16 block4 shape)
    ('3x3_slab_8x8_ragged', 6, 8, 64, 128, 3, 1, 1, 1, 8),
    ('3x3_slab_64map', 1, 64, 64, 64, 3, 1, 1, 1, 64),
    ('3x3_slab_cin96_cout192', 2, 16, 96, 192, 3, 1, 1, 1, 16),
    ('3x3_slab_rate2_512', 3, 16, 128, 256, 3, 1, 2, 2, 16),
    ('3x3_slab_32map', 2, 32, 128, 128, 3, 1, 1, 1, 32),
    # RN101-s8 block3 conv2 at its per-GPU batch of 32 (rate 2 on 32x32 maps: 64 rows of halo, exactly 256 tiles of 128 cout x
    # 256 px): the 128-cout slab configuration with the 384-row slab (tests/test_kernel_coverage.py runs the full 256 -> 256 shape)
    ('3x3_slab_rate2_32map_r384', 32, 32, 128, 256, 3, 1, 2, 2, 32),
    # round 6, sub-grid pixel order (conv3x3_f16_slab<...>+subgrid): rates whose plain halo (rate x W rows) exceeds the slab --
    # rate 4 on 64-wide maps (16 x 16 sub-images), rate 8 on 64-wide maps (8 x 8 sub-images: four per 256-pixel tile), rate 4 on
    # 32-wide maps (the stride-8 block4 shape), rate 8 on a 16 x 16 map (2 x 2 sub-images: every tap but the centre leaves them)
    ('3x3_subgrid_rate4_64map', 2, 64, 64, 128, 3, 1, 4, 4, 64),
    ('3x3_subgrid_rate8_64map', 1, 64, 128, 192, 3, 1, 8, 8, 64),
    ('3x3_subgrid_rate4_32map', 3, 32, 64, 64, 3, 1, 4, 4, 32),
    ('3x3_subgrid_rate8_16map', 2, 16, 64, 128, 3, 1, 8, 8, 16),
    # persistent weight-resident 64 -> 64 kernel (conv3x3_c64.hip): blocks walk several 128-pixel tiles (image borders
    # inside a block's range, halo rows shared between consecutive tiles), 64- / 32- / 16-wide maps
    ('3x3_c64_many_tiles', 21, 64, 64, 64, 3, 1, 1, 1, 64),
    ('3x3_c64_32map', 5, 32, 64, 64, 3, 1, 1, 1, 32),
    ('3x3_c64_16map', 9, 16, 64, 64, 3, 1, 1, 1, 16),
]


def _mk(rng, n, h_in, c_in, c_out, k):
    x = rng.standard_normal((n, h_in, h_in, c_in)).astype(np.float32)
    w = (rng.standard_normal((c_out, k, k, c_in)) * np.sqrt(2.0 / (k * k * c_in))).astype(np.float32)
    b = (rng.standard_normal(c_out) * 0.1).astype(np.float32)
    return x, w, b


@pytest.mark.parametrize('case', CONV_CASES, ids=[c[0] for c in CONV_CASES])
@pytest.mark.parametrize('variant', ['plain', 'relu', 'prologue', 'residual', 'f32out'])
def test_conv_f16(lib, cuda, case, variant):
    name, n, h_in, c_in, c_out, k, stride, dil, pad, h_out = case
    if variant == 'prologue' and (k != 1 or pad > 0):
        pytest.skip('prologue is defined for un-padded 1x1 convs only')
    rng = np.random.default_rng(zlib.crc32(f'{name}/{variant}'.encode()))
    x, w, b = _mk(rng, n, h_in, c_in, c_out, k)
    x16, w16 = x.astype(np.float16), w.astype(np.float16)
    pro = res = None
    if variant == 'prologue':
        pro = (rng.uniform(0.5, 1.5, c_in).astype(np.float16), (rng.standard_normal(c_in) * 0.2).astype(np.float16))
    if variant == 'residual':
        res = rng.standard_normal((n, h_out, h_out, c_out)).astype(np.float16)
    d = H.conv_desc(n, h_in, c_in, h_out, c_out, k, stride, dil, pad, prologue=pro is not None,
                    relu=variant == 'relu', residual=res is not None, res_h=h_out,
                    out_dtype=_lib.METRO_F32 if variant == 'f32out' else _lib.METRO_F16,
                    in_dtype=_lib.METRO_F16)
    got = H.run_conv_f16(lib, cuda, d, x16, w16, b, pro, res).astype(np.float64)
    if pro is not None:
        # the kernel applies relu(x*s+b) with ONE fp16 rounding (v_pk_fma_f16): mirror it
        xin = np.maximum(np.float16(x16.astype(np.float64) * pro[0].astype(np.float64) + pro[1].astype(np.float64)), 0)
        ref = H.ref_conv_nhwc(xin.astype(np.float64), w16, b, stride, dil, pad, h_out, relu=False).numpy()
    else:
        ref = H.ref_conv_nhwc(x16, w16, b, stride, dil, pad, h_out, relu=variant == 'relu', res=res).numpy()
    assert np.isfinite(got).all()
    scale = np.abs(ref).max()
    tol = (2e-3 if d.out_dtype == _lib.METRO_F16 else 2e-5) * scale   # fp16 output rounding / fp32 accumulation
    assert np.abs(got - ref).max() <= tol, (np.abs(got - ref).max(), scale)


# (name, n, side, c_in, c_out, rate): 3x3 layers with enough 512-pixel tiles (>= 256 of 128 cout x 512 px) for the
# 512-pixel slab configuration -- the stride-16 net's conv2 shapes at batch 256, a half-image-per-tile map, a ragged
# last tile.  The fp64 reference is computed for a sample of images (first / last, both halves of a tile pair).
P512_CASES = [('block3_b256', 256, 16, 256, 256, 1), ('block4_b256_rate2', 256, 16, 512, 512, 2), ('block2_b256', 256, 32, 128, 128, 1),
              ('ragged_515', 515, 16, 64, 256, 1), ('cout_tail_320', 400, 16, 64, 320, 1)]


@pytest.mark.parametrize('case', P512_CASES, ids=[c[0] for c in P512_CASES])
@pytest.mark.parametrize('relu', [False, True], ids=['plain', 'relu'])
def test_conv3x3_slab_512px_tiles(lib, cuda, case, relu):
    name, n, side, c_in, c_out, rate = case
    rng = np.random.default_rng(zlib.crc32(f'p512/{name}/{relu}'.encode()))
    x16 = rng.standard_normal((n, side, side, c_in)).astype(np.float16)
    w16 = (rng.standard_normal((c_out, 3, 3, c_in)) * np.sqrt(2.0 / (9 * c_in))).astype(np.float16)
    b = (rng.standard_normal(c_out) * 0.1).astype(np.float32)
    d = H.conv_desc(n, side, c_in, side, c_out, 3, 1, rate, rate, relu=relu, in_dtype=_lib.METRO_F16)
    got = H.run_conv_f16(lib, cuda, d, x16, w16, b).astype(np.float64)
    assert np.isfinite(got).all()
    sample = sorted({0, 1, 2, n // 2, n // 2 + 1, n - 2, n - 1})
    ref = H.ref_conv_nhwc(x16[sample], w16, b, 1, rate, rate, side, relu=relu).numpy()
    tol = 2e-3 * np.abs(ref).max()
    assert np.abs(got[sample] - ref).max() <= tol, (np.abs(got[sample] - ref).max(), np.abs(ref).max())
    # race screen (hand-counted vmcnt over a 4-deep ring with a rotating slot): the same launch again, with an unrelated
    # memory-heavy kernel in between to perturb the timing, gives the same bits
    junk = torch.empty(192 << 20, dtype=torch.uint8, device=cuda)
    for it in range(4):
        if it % 2 == 0:
            junk.fill_(it + 1)
        again = H.run_conv_f16(lib, cuda, d, x16, w16, b).astype(np.float64)
        assert np.array_equal(got, again), f'repeat {it} differs'


@pytest.mark.parametrize('shape', [(2, 8, 64, 128), (5, 16, 64, 256), (3, 11, 64, 256), (5, 16, 128, 512)],
                         ids=['ring', 'pw64', 'pw64_ragged', 'pw128'])
@pytest.mark.parametrize('res_stride,res_offset', [(2, 0), (2, 1)])
def test_conv_f16_strided_residual(lib, cuda, res_stride, res_offset, shape):
    """identity shortcut of a strided unit: even pixels (non-centered) or odd pixels
    (centered, x[1:,1:][::2,::2]) -- reference resnet_v2.py:113-121, resnet_utils.py:76-79 (KA8).
    64->256 and 128->512 run in the persistent kernel (sub-sampled shortcut rows gathered by its LDS-DMA)."""
    rng = np.random.default_rng(5 + res_offset)
    n, h, c_in, c_out = shape
    x, w, b = _mk(rng, n, h, c_in, c_out, 1)
    res = rng.standard_normal((n, 2 * h, 2 * h, c_out)).astype(np.float16)
    d = H.conv_desc(n, h, c_in, h, c_out, 1, residual=True, res_h=2 * h, res_stride=res_stride,
                    res_offset=res_offset)
    got = H.run_conv_f16(lib, cuda, d, x, w, b, res=res).astype(np.float64)
    ref = H.ref_conv_nhwc(x.astype(np.float16), w.astype(np.float16), b, 1, 1, 0, h, res=res,
                          res_stride=res_stride, res_offset=res_offset).numpy()
    assert np.abs(got - ref).max() <= 2e-3 * np.abs(ref).max()


def _dev(a, cuda, dt):
    return torch.from_numpy(np.ascontiguousarray(a.astype(dt))).to(cuda)


@pytest.mark.parametrize('shape', [(3, 7, 64, 256, 64), (25, 64, 64, 256, 64), (2, 16, 256, 512, 128), (3, 9, 256, 512, 128),
                                   (40, 32, 256, 512, 128), (3, 9, 512, 1024, 256)],
                         ids=['block1_ragged', 'block1_many_tiles', 'block2', 'block2_ragged', 'block2_many_tiles', 'block3_ragged'])
def test_conv_f16_pair(lib, cuda, shape):
    """Projection shortcut + conv1 of a unit in one launch (reference resnet_v2.py:122-128): both outputs
    against the fp64 reference on the same fp16 operands.  64 -> 256+64 and (round 5) 256 -> 512+128 run in the persistent
    weight-resident kernel: a ragged last tile (243 pixels) and more tiles than blocks (40 960 pixels = 5 tiles per block)."""
    n, h, c_in, c_sc, cb = shape
    rng = np.random.default_rng(zlib.crc32(repr(shape).encode()))
    x, w, b = _mk(rng, n, h, c_in, c_sc + cb, 1)
    x16, w16 = x.astype(np.float16), w.astype(np.float16)
    sc = rng.uniform(0.5, 1.5, c_in).astype(np.float16)
    sh = (rng.standard_normal(c_in) * 0.2).astype(np.float16)
    d = H.conv_desc(n, h, c_in, h, c_sc + cb, 1, prologue=True, in_dtype=_lib.METRO_F16)
    out = torch.full((n, h, h, c_sc), float('nan'), dtype=torch.float16, device=cuda)
    out2 = torch.full((n, h, h, cb), float('nan'), dtype=torch.float16, device=cuda)
    tx, tw, tb, ts, tsh = _dev(x16, cuda, np.float16), _dev(w16, cuda, np.float16), _dev(b, cuda, np.float32), _dev(sc, cuda, np.float16), _dev(sh, cuda, np.float16)
    check(lib.metro_conv_f16_pair(C.byref(d), H.ptr(tx), H.ptr(tw), H.ptr(tb), H.ptr(ts), H.ptr(tsh), H.ptr(out), c_sc,
                                  H.ptr(out2), None), 'metro_conv_f16_pair')
    torch.cuda.synchronize()
    xin = np.maximum(np.float16(x16.astype(np.float64) * sc.astype(np.float64) + sh.astype(np.float64)), 0)
    ref = H.ref_conv_nhwc(xin.astype(np.float64), w16, b, 1, 1, 0, h).numpy()
    got, got2 = out.cpu().double().numpy(), out2.cpu().double().numpy()
    assert np.isfinite(got).all() and np.isfinite(got2).all()
    tol = 2e-3 * np.abs(ref).max()
    assert np.abs(got - ref[..., :c_sc]).max() <= tol
    assert np.abs(got2 - np.maximum(ref[..., c_sc:], 0)).max() <= tol


@pytest.mark.parametrize('shape', [(3, 7), (2, 8), (12, 64), (3, 7, 128), (2, 8, 128), (25, 32, 128)],
                         ids=['ragged', 'small', 'many_tiles', 'block2_ragged', 'block2_small', 'block2_many_tiles'])
def test_conv_f16_next(lib, cuda, shape):
    """conv3 + shortcut of unit u and conv1 of unit u+1 in one launch (reference resnet_v2.py:134-138 then
    :119,127-128): first output against the fp64 reference, second output against a restatement on the fp16
    first output the kernel itself stored (tight).  block1 shapes (64 -> 256, next conv1 256 -> 64) and block2 shapes
    (128 -> 512, next conv1 512 -> 128: all 512 channels of a 32-pixel tile in one block, W1' in registers,
    v_mfma_f32_16x16x32_f16 from the LDS tile)."""
    n, h = shape[:2]
    c_in = shape[2] if len(shape) > 2 else 64
    c_out, c2 = 4 * c_in, c_in
    rng = np.random.default_rng(zlib.crc32(repr(shape).encode()) + 1)
    x, w, b = _mk(rng, n, h, c_in, c_out, 1)
    res = rng.standard_normal((n, h, h, c_out)).astype(np.float16)
    w2 = (rng.standard_normal((c2, c_out)) * np.sqrt(2.0 / c_out)).astype(np.float16)
    b2 = (rng.standard_normal(c2) * 0.1).astype(np.float32)
    sc2 = rng.uniform(0.5, 1.5, c_out).astype(np.float16)
    sh2 = (rng.standard_normal(c_out) * 0.2).astype(np.float16)
    d = H.conv_desc(n, h, c_in, h, c_out, 1, residual=True, res_h=h, in_dtype=_lib.METRO_F16)
    out = torch.full((n, h, h, c_out), float('nan'), dtype=torch.float16, device=cuda)
    out2 = torch.full((n, h, h, c2), float('nan'), dtype=torch.float16, device=cuda)
    t = [_dev(x, cuda, np.float16), _dev(w, cuda, np.float16), _dev(b, cuda, np.float32), _dev(res, cuda, np.float16),
         _dev(w2, cuda, np.float16), _dev(b2, cuda, np.float32), _dev(sc2, cuda, np.float16), _dev(sh2, cuda, np.float16)]
    check(lib.metro_conv_f16_next(C.byref(d), H.ptr(t[0]), H.ptr(t[1]), H.ptr(t[2]), H.ptr(t[3]), H.ptr(out), H.ptr(t[4]),
                                  H.ptr(t[5]), H.ptr(t[6]), H.ptr(t[7]), H.ptr(out2), c2, None), 'metro_conv_f16_next')
    torch.cuda.synchronize()
    ref = H.ref_conv_nhwc(x.astype(np.float16), w.astype(np.float16), b, 1, 1, 0, h, res=res).numpy()
    got, got2 = out.cpu().double().numpy(), out2.cpu().double().numpy()
    assert np.isfinite(got).all() and np.isfinite(got2).all()
    assert np.abs(got - ref).max() <= 2e-3 * np.abs(ref).max()
    pre = np.maximum(np.float16(got * sc2.astype(np.float64) + sh2.astype(np.float64)), 0).astype(np.float64)
    want2 = np.maximum(pre @ w2.astype(np.float64).T + b2.astype(np.float64), 0)
    assert np.abs(got2 - want2).max() <= 2e-3 * np.abs(want2).max()


@pytest.mark.parametrize('shape', [(21, 64), (5, 32), (9, 16), (1, 64)], ids=['many_tiles_64map', '32map', '16map', 'one_image'])
def test_conv_f16_conv1_conv2(lib, cuda, shape):
    """conv1 (1x1 on relu(x * s + b), folded BN + ReLU) fused in FRONT of the weight-resident 3x3 (conv3x3_c64 PRE1; reference
    resnet_v2.py:119,127-132): every element against fp64 with ONE fp16 rounding of conv1's output (it lives in LDS as fp16).
    Image borders inside a block's tile range (taps outside the image read zeros of t1, NOT conv1 of a zero input), halo rows
    recomputed by both tiles that share them; repeated launches are bit-identical."""
    n, h = shape
    rng = np.random.default_rng(zlib.crc32(repr(('c1c2',) + shape).encode()))
    x = rng.standard_normal((n, h, h, 64)).astype(np.float16)
    w1 = (rng.standard_normal((64, 64)) * np.sqrt(2.0 / 64)).astype(np.float16)
    b1 = (rng.standard_normal(64) * 0.3 + 0.2).astype(np.float32)          # conv1 of a zero input is far from zero
    ps = rng.uniform(0.5, 1.5, 64).astype(np.float16)
    pb = (rng.standard_normal(64) * 0.2 + 0.3).astype(np.float16)
    w2 = (rng.standard_normal((64, 3, 3, 64)) * np.sqrt(2.0 / 576)).astype(np.float16)
    b2 = (rng.standard_normal(64) * 0.1).astype(np.float32)
    d = H.conv_desc(n, h, 64, h, 64, 3, 1, 1, 1, relu=True, in_dtype=_lib.METRO_F16)
    t = [_dev(x, cuda, np.float16), _dev(w1, cuda, np.float16), _dev(b1, cuda, np.float32), _dev(ps, cuda, np.float16),
         _dev(pb, cuda, np.float16), _dev(w2, cuda, np.float16), _dev(b2, cuda, np.float32)]
    outs = []
    for rep in range(3):
        out = torch.full((n, h, h, 64), float('nan'), dtype=torch.float16, device=cuda)
        check(lib.metro_conv_f16_conv1_conv2(C.byref(d), *[H.ptr(a) for a in t], H.ptr(out), None), 'metro_conv_f16_conv1_conv2')
        torch.cuda.synchronize()
        outs.append(out.cpu().numpy())
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    xin = np.maximum((x.astype(np.float64) * ps.astype(np.float64) + pb.astype(np.float64)).astype(np.float16).astype(np.float64), 0)
    t1 = np.maximum(xin @ w1.astype(np.float64).T + b1.astype(np.float64), 0).astype(np.float16)
    ref = H.ref_conv_nhwc(t1, w2, b2, 1, 1, 1, h, relu=True).numpy()
    got = outs[0].astype(np.float64)
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 2e-3 * np.abs(ref).max(), (np.abs(got - ref).max(), np.abs(ref).max())
    # and it IS the two separate launches: conv1 through the generic kernel, then the plain weight-resident 3x3 -- same bits
    d1 = H.conv_desc(n, h, 64, h, 64, 1, prologue=True, relu=True, in_dtype=_lib.METRO_F16)
    sep1 = H.run_conv_f16(lib, cuda, d1, x, w1.reshape(64, 1, 1, 64), b1, pro=(ps, pb))
    sep = H.run_conv_f16(lib, cuda, d, sep1, w2, b2)
    assert (sep == outs[0]).mean() > 0.995 and np.abs(sep.astype(np.float64) - got).max() <= 2e-3 * np.abs(ref).max()


@pytest.mark.parametrize('shape', [(3, 7), (2, 8), (12, 64)], ids=['ragged', 'small', 'many_tiles'])
def test_conv_f16_next_proj(lib, cuda, shape):
    """conv3 + the unit's PROJECTION shortcut computed in the launch + conv1 of the next unit (conv_pw64 PSC; reference
    resnet_v2.py:119,122-125,134-138 then :119,127-128 of unit u+1): first output against fp64 with one rounding per addend
    (fp16(conv3 + b) + fp16(shortcut conv + b), the fp16 Add of the graph), second output from the kernel's own first output;
    and the same bits as metro_conv_f16_next fed the separately computed shortcut tensor."""
    n, h = shape
    rng = np.random.default_rng(zlib.crc32(repr(('nextproj',) + shape).encode()))
    t2in, w3, b3 = _mk(rng, n, h, 64, 256, 1)
    xu = rng.standard_normal((n, h, h, 64)).astype(np.float16)
    wsc = (rng.standard_normal((256, 64)) * np.sqrt(2.0 / 64)).astype(np.float16)
    bsc = (rng.standard_normal(256) * 0.1).astype(np.float32)
    ps = rng.uniform(0.5, 1.5, 64).astype(np.float16)
    pb = (rng.standard_normal(64) * 0.2).astype(np.float16)
    w2 = (rng.standard_normal((64, 256)) * np.sqrt(2.0 / 256)).astype(np.float16)
    b2 = (rng.standard_normal(64) * 0.1).astype(np.float32)
    sc2 = rng.uniform(0.5, 1.5, 256).astype(np.float16)
    sh2 = (rng.standard_normal(256) * 0.2).astype(np.float16)
    d = H.conv_desc(n, h, 64, h, 256, 1, in_dtype=_lib.METRO_F16)
    f16, f32 = np.float16, np.float32
    t = [_dev(t2in, cuda, f16), _dev(w3, cuda, f16), _dev(b3, cuda, f32), _dev(xu, cuda, f16), _dev(wsc, cuda, f16), _dev(bsc, cuda, f32),
         _dev(ps, cuda, f16), _dev(pb, cuda, f16)]
    t_next = [_dev(w2, cuda, f16), _dev(b2, cuda, f32), _dev(sc2, cuda, f16), _dev(sh2, cuda, f16)]
    out = torch.full((n, h, h, 256), float('nan'), dtype=torch.float16, device=cuda)
    out2 = torch.full((n, h, h, 64), float('nan'), dtype=torch.float16, device=cuda)
    check(lib.metro_conv_f16_next_proj(C.byref(d), *[H.ptr(a) for a in t], H.ptr(out), *[H.ptr(a) for a in t_next], H.ptr(out2), 64, None),
          'metro_conv_f16_next_proj')
    torch.cuda.synchronize()
    got, got2 = out.cpu().double().numpy(), out2.cpu().double().numpy()
    assert np.isfinite(got).all() and np.isfinite(got2).all()
    conv3 = H.ref_conv_nhwc(t2in.astype(f16), w3.astype(f16), b3, 1, 1, 0, h).numpy().astype(f16).astype(np.float64)
    xin = np.maximum((xu.astype(np.float64) * ps.astype(np.float64) + pb.astype(np.float64)).astype(f16).astype(np.float64), 0)
    sc = (xin @ wsc.astype(np.float64).T + bsc.astype(np.float64)).astype(f16).astype(np.float64)
    ref = conv3 + sc
    assert np.abs(got - ref).max() <= 2e-3 * np.abs(ref).max()
    pre = np.maximum(np.float16(got * sc2.astype(np.float64) + sh2.astype(np.float64)), 0).astype(np.float64)
    want2 = np.maximum(pre @ w2.astype(np.float64).T + b2.astype(np.float64), 0)
    assert np.abs(got2 - want2).max() <= 2e-3 * np.abs(want2).max()
    # the shortcut as its own launch, then metro_conv_f16_next on that tensor: the same arithmetic, tensor for tensor
    dsc = H.conv_desc(n, h, 64, h, 256, 1, prologue=True, in_dtype=_lib.METRO_F16)
    sc_t = H.run_conv_f16(lib, cuda, dsc, xu, wsc.reshape(256, 1, 1, 64), bsc, pro=(ps, pb))
    dn = H.conv_desc(n, h, 64, h, 256, 1, residual=True, res_h=h, in_dtype=_lib.METRO_F16)
    o1 = torch.full_like(out, float('nan'))
    o2 = torch.full_like(out2, float('nan'))
    tsc = _dev(sc_t, cuda, f16)
    check(lib.metro_conv_f16_next(C.byref(dn), H.ptr(t[0]), H.ptr(t[1]), H.ptr(t[2]), H.ptr(tsc), H.ptr(o1), *[H.ptr(a) for a in t_next],
                                  H.ptr(o2), 64, None), 'metro_conv_f16_next')
    torch.cuda.synchronize()
    assert torch.equal(o1, out) and torch.equal(o2, out2), 'in-launch projection shortcut vs the shortcut tensor of a separate launch'


@pytest.mark.parametrize('n,side', [(2, 64), (3, 96), (9, 256), (33, 256)])
def test_stem_pool_f16(lib, cuda, n, side):
    """Stem 7x7/2 + zero-padded 3x3/2 max-pool in one launch against torch fp64 on the same fp16 operands
    (reference resnet_v2.py:219-224, resnet_utils.py:138-185); 9 x 256^2 = more patches than resident blocks.  At 256-pixel
    crops the fp32-input entry runs stem_pool_rows_kernel (conv pixels as MFMA rows, the pool in registers; 33 crops = more
    bands than a device holds): it must give the bits of the patch kernel -- same k order per output, one fp16 rounding."""
    rng = np.random.default_rng(n * 1000 + side)
    img = rng.uniform(-1, 1, (n, side, side, 3)).astype(np.float32)
    w = (rng.standard_normal((64, 7, 7, 3)) * np.sqrt(2.0 / 147)).astype(np.float16)      # [o][kh][kw][c]
    b = (rng.standard_normal(64) * 0.5).astype(np.float32)
    wp = np.zeros((64, 7, 8, 4), np.float16)
    wp[:, :, :7, :3] = w
    timg = torch.from_numpy(img).to(cuda)
    prepped = torch.empty((n, side + 6, side + 8, 4), dtype=torch.float16, device=cuda)
    check(lib.metro_prep_input_f16(H.ptr(timg), n, side, H.ptr(prepped), None), 'metro_prep_input_f16')
    out = torch.full((n, side // 4, side // 4, 64), float('nan'), dtype=torch.float16, device=cuda)
    tw, tb = _dev(wp, cuda, np.float16), _dev(b, cuda, np.float32)      # named: a temporary would be freed before the launch
    check(lib.metro_stem_pool_f16(H.ptr(prepped), H.ptr(tw), H.ptr(tb), H.ptr(out), n, side, None), 'metro_stem_pool_f16')
    torch.cuda.synchronize()
    xi = torch.from_numpy(img.astype(np.float16).astype(np.float64)).permute(0, 3, 1, 2)
    conv = torch.nn.functional.conv2d(torch.nn.functional.pad(xi, (3, 3, 3, 3)),
                                      torch.from_numpy(w.astype(np.float64)).permute(0, 3, 1, 2),
                                      torch.from_numpy(b.astype(np.float64)), stride=2).half().double()
    want = torch.nn.functional.max_pool2d(torch.nn.functional.pad(conv, (1, 1, 1, 1)), 3, 2).permute(0, 2, 3, 1).numpy()
    got = out.cpu().double().numpy()
    assert np.isfinite(got).all()
    assert np.abs(got - want).max() <= 2e-3 * np.abs(want).max()
    assert (got == want).mean() > 0.98
    # reading the fp32 crops directly (cast + border inside the kernel) gives the same bits
    out2 = torch.full_like(out, float('nan'))
    check(lib.metro_stem_pool_f32in(H.ptr(timg), H.ptr(tw), H.ptr(tb), H.ptr(out2), n, side, None), 'metro_stem_pool_f32in')
    torch.cuda.synchronize()
    assert torch.equal(out, out2)


def test_fused_entry_points_reject_unsupported_shapes(lib, cuda):
    d = H.conv_desc(1, 8, 64, 8, 128, 1, residual=True, res_h=8, in_dtype=_lib.METRO_F16)     # c_out != 256
    p = C.c_void_p(256)
    assert lib.metro_conv_f16_next(C.byref(d), p, p, p, p, p, p, p, p, p, p, 64, None) == -1
    assert b'conv_f16_next' in lib.metro_last_error()
    d = H.conv_desc(1, 8, 64, 8, 320, 1, prologue=True, in_dtype=_lib.METRO_F16)
    assert lib.metro_conv_f16_pair(C.byref(d), p, p, p, p, p, p, 100, p, None) == -1        # split % 256
    assert lib.metro_stem_pool_f16(p, p, p, p, 1, 100, None) == -1                            # side % 32


@pytest.mark.parametrize('case', CONV_CASES, ids=[c[0] for c in CONV_CASES])
@pytest.mark.parametrize('variant', ['plain', 'relu_residual', 'prologue'])
@pytest.mark.parametrize('store', ['f32', 'f64'])
def test_conv_f64acc(lib, cuda, case, variant, store):
    name, n, h_in, c_in, c_out, k, stride, dil, pad, h_out = case
    if variant == 'prologue' and (k != 1 or pad > 0):
        pytest.skip('prologue is defined for un-padded 1x1 convs only')
    rng = np.random.default_rng(zlib.crc32(f'{name}/{variant}/64'.encode()))
    x, w, b = _mk(rng, n, h_in, c_in, c_out, k)
    w = w.astype(np.float64) * (1 + 1e-9)    # genuinely fp64 weights
    pro = res = None
    if variant == 'prologue':
        pro = (rng.uniform(0.5, 1.5, c_in), rng.standard_normal(c_in) * 0.2)
    if variant == 'relu_residual':
        res = rng.standard_normal((n, h_out, h_out, c_out)).astype(np.float32)
    d = H.conv_desc(n, h_in, c_in, h_out, c_out, k, stride, dil, pad, prologue=pro is not None,
                    relu=variant == 'relu_residual', residual=res is not None, res_h=h_out,
                    out_dtype=_lib.METRO_F32 if store == 'f32' else _lib.METRO_F64)
    got = H.run_conv_f64acc(lib, cuda, d, x, w, b.astype(np.float64), pro, res).astype(np.float64)
    ref = H.ref_conv_nhwc(x, w, b.astype(np.float64), stride, dil, pad, h_out, pro=pro,
                          relu=variant == 'relu_residual', res=res).numpy()
    # fp64 accumulation, one rounding to fp32: at most 1 ulp of the result (plus reordering ~1e-15)
    err = np.abs(got - ref)
    if store == 'f32':
        ulp = np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64)
        assert (err <= 0.5001 * ulp + 1e-12).all(), (err / ulp).max()
    else:
        assert err.max() <= 1e-12 * max(np.abs(ref).max(), 1.0), err.max()


@pytest.mark.parametrize('case', CONV_CASES, ids=[c[0] for c in CONV_CASES])
@pytest.mark.parametrize('variant', ['plain', 'relu_residual', 'prologue'])
def test_conv_f32m(lib, cuda, case, variant):
    """The fp32-matrix-core kernel of the F32M mode (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulation in ascending
    k) on every conv shape, against fp64 on the same fp32 operands: what is left is fp32 accumulation, <= sqrt(K) ulps of the
    layer maximum (bar 2e-5 relative: K <= 4608)."""
    name, n, h_in, c_in, c_out, k, stride, dil, pad, h_out = case
    if variant == 'prologue' and (k != 1 or pad > 0):
        pytest.skip('prologue is defined for un-padded 1x1 convs only')
    rng = np.random.default_rng(zlib.crc32(f'{name}/{variant}/32m'.encode()))
    x, w, b = _mk(rng, n, h_in, c_in, c_out, k)
    pro = res = None
    if variant == 'prologue':
        pro = (rng.uniform(0.5, 1.5, c_in).astype(np.float32), (rng.standard_normal(c_in) * 0.2).astype(np.float32))
    if variant == 'relu_residual':
        res = rng.standard_normal((n, h_out, h_out, c_out)).astype(np.float32)
    d = H.conv_desc(n, h_in, c_in, h_out, c_out, k, stride, dil, pad, prologue=pro is not None, relu=variant == 'relu_residual',
                    residual=res is not None, res_h=h_out, out_dtype=_lib.METRO_F32, in_dtype=_lib.METRO_F32)
    t = [_dev(x, cuda, np.float32), _dev(w, cuda, np.float32), _dev(b, cuda, np.float32)]
    ts = _dev(pro[0], cuda, np.float32) if pro else None
    tsh = _dev(pro[1], cuda, np.float32) if pro else None
    tr = _dev(res, cuda, np.float32) if res is not None else None
    out = torch.full((n, h_out, h_out, c_out), float('nan'), dtype=torch.float32, device=cuda)
    check(lib.metro_conv_f32m(C.byref(d), H.ptr(t[0]), H.ptr(t[1]), H.ptr(t[2]), H.ptr(ts), H.ptr(tsh), H.ptr(tr), H.ptr(out), None),
          'metro_conv_f32m')
    torch.cuda.synchronize()
    got = out.cpu().double().numpy()
    ref = H.ref_conv_nhwc(x, w, b, stride, dil, pad, h_out, pro=pro, relu=variant == 'relu_residual', res=res).numpy()
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max(), (np.abs(got - ref).max(), np.abs(ref).max())


def test_conv_f32m_stem_3ch(lib, cuda):
    """7x7/2 stem on the raw 3-channel image with TF explicit pad 3 (resnet_utils.py:125-135), fp32 matrix cores."""
    from oracle.forward import conv2d_same
    rng = np.random.default_rng(17)
    x = rng.random((2, 64, 64, 3)).astype(np.float32)
    w_hwio = (rng.standard_normal((7, 7, 3, 16)) * 0.1).astype(np.float32)
    b = (rng.standard_normal(16) * 0.01).astype(np.float32)
    d = H.conv_desc(2, 64, 3, 32, 16, 7, stride=2, pad=3, out_dtype=_lib.METRO_F32, in_dtype=_lib.METRO_F32)
    t = [_dev(x, cuda, np.float32), _dev(w_hwio.transpose(3, 0, 1, 2), cuda, np.float32), _dev(b, cuda, np.float32)]
    out = torch.full((2, 32, 32, 16), float('nan'), dtype=torch.float32, device=cuda)
    check(lib.metro_conv_f32m(C.byref(d), H.ptr(t[0]), H.ptr(t[1]), H.ptr(t[2]), None, None, None, H.ptr(out), None), 'metro_conv_f32m')
    torch.cuda.synchronize()
    ref = conv2d_same(torch.from_numpy(x).double().permute(0, 3, 1, 2), torch.from_numpy(w_hwio).double().permute(3, 2, 0, 1), 2, 1,
                      False).permute(0, 2, 3, 1).numpy() + b
    assert np.abs(out.cpu().double().numpy() - ref).max() <= 2e-6 * np.abs(ref).max()


def test_conv_f64acc_stem_3ch(lib, cuda):
    """7x7/2 stem on the raw 3-channel image with TF explicit pad 3 (resnet_utils.py:125-135)."""
    from oracle.forward import conv2d_same
    rng = np.random.default_rng(7)
    x = rng.random((2, 64, 64, 3)).astype(np.float32)
    w_hwio = (rng.standard_normal((7, 7, 3, 16)) * 0.1)
    b = rng.standard_normal(16) * 0.01
    d = H.conv_desc(2, 64, 3, 32, 16, 7, stride=2, pad=3, out_dtype=_lib.METRO_F32)
    got = H.run_conv_f64acc(lib, cuda, d, x, w_hwio.transpose(3, 0, 1, 2), b)
    ref = conv2d_same(torch.from_numpy(x).double().permute(0, 3, 1, 2),
                      torch.from_numpy(w_hwio).permute(3, 2, 0, 1), 2, 1, False).permute(0, 2, 3, 1).numpy() + b
    assert np.abs(got - ref).max() <= 1e-6 * np.abs(ref).max()


def test_stem_f16_via_bordered_image(lib, cuda):
    """prep_input + 7x1-tap conv over the bordered 4-channel image == conv2d_same 7x7/2 pad 3."""
    from oracle.forward import conv2d_same
    rng = np.random.default_rng(8)
    n, side, co = 2, 64, 64
    x = rng.random((n, side, side, 3)).astype(np.float32)
    w_hwio = (rng.standard_normal((7, 7, 3, co)) * 0.1).astype(np.float32)
    b = (rng.standard_normal(co) * 0.01).astype(np.float32)
    tx = torch.from_numpy(x).to(cuda)
    prep = torch.full((n, side + 6, side + 8, 4), float('nan'), dtype=torch.float16, device=cuda)
    check(lib.metro_prep_input_f16(H.ptr(tx), n, side, H.ptr(prep), C.c_void_p(0)), 'prep')
    torch.cuda.synchronize()
    p = prep.cpu().numpy()
    exp = np.zeros((n, side + 6, side + 8, 4), np.float16)
    exp[:, 3:3 + side, 3:3 + side, :3] = x.astype(np.float16)
    assert np.array_equal(p, exp)
    packed = np.zeros((co, 7, 8, 4), np.float16)
    packed[:, :, :7, :3] = w_hwio.transpose(3, 0, 1, 2).astype(np.float16)
    d = H.conv_desc(n, side + 6, 32, side // 2, co, 0, stride=2, pad=0, w_in=side + 8, in_pix_stride=4,
                    kh=7, kw=1, in_dtype=_lib.METRO_F16)
    got = H.run_conv_f16(lib, cuda, d, p, packed.reshape(co, 7, 1, 32), b).astype(np.float64)
    ref = conv2d_same(torch.from_numpy(x.astype(np.float16)).double().permute(0, 3, 1, 2),
                      torch.from_numpy(w_hwio.astype(np.float16)).double().permute(3, 2, 0, 1), 2, 1,
                      False).permute(0, 2, 3, 1).numpy() + b
    assert np.abs(got - ref).max() <= 2e-3 * np.abs(ref).max()


@pytest.mark.parametrize('dtype', ['f16', 'f32', 'f64'])
def test_maxpool_zeropad(lib, cuda, dtype):
    from oracle.forward import max_pool2d_same_zeropad
    rng = np.random.default_rng(9)
    tdt = {'f16': torch.float16, 'f32': torch.float32, 'f64': torch.float64}[dtype]
    x = torch.from_numpy(rng.standard_normal((3, 32, 32, 64)).astype(np.float32)).to(tdt)
    x[1] = -x[1].abs() - 0.5                       # KA7: all-negative image
    out = torch.full((3, 16, 16, 64), float('nan'), dtype=tdt, device=cuda)
    xd = x.to(cuda)
    check(lib.metro_maxpool3x3s2_zeropad(H.ptr(xd), H.ptr(out), 3, 32, 32, 64,
                                         {'f16': _lib.METRO_F16, 'f32': _lib.METRO_F32, 'f64': _lib.METRO_F64}[dtype],
                                         C.c_void_p(0)), 'maxpool')
    torch.cuda.synchronize()
    ref = max_pool2d_same_zeropad(x.double().permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    got = out.cpu().double()
    assert torch.equal(got, ref)                   # max is exact in any dtype
    assert (got[1, 0, :, :] == 0).all() and (got[1, :, 0, :] == 0).all()   # border windows saw the pad
    assert (got[1, 1:, 1:, :] < 0).all()


SA_SPECS = [ModelSpec(50, 32, 'h36m'), ModelSpec(50, 16, 'h36m'), ModelSpec(50, 16, 'many19'),
            ModelSpec(101, 8, 'merged'), ModelSpec(50, 4, 'h36m')]


@pytest.mark.parametrize('spec', SA_SPECS, ids=lambda s: f'rn{s.arch}-s{s.stride}-{s.dataset}')
@pytest.mark.parametrize('precise', [0, 1, 2])
def test_softargmax_random(lib, cuda, spec, precise):
    from oracle.forward import logits_to_output
    rng = np.random.default_rng(spec.stride)
    n, s, c = 3, spec.heatmap_side, spec.n_head_channels
    logits = (rng.standard_normal((n, s, s, c)) * 4).astype(np.float32)
    got = H.run_softargmax(lib, cuda, spec, logits, precise)
    ref = logits_to_output(H.oracle_spec(spec), logits).numpy()
    tol = 1e-3 if precise else 2e-3       # mm; fast mode: fp32 accumulation of 8*S*S terms (measured 2-7e-4 on the nets, <= 1.5e-3 on these N(0,4) logits)
    assert np.abs(got - ref).max() <= tol, np.abs(got - ref).max()


def test_softargmax_known_answers(lib, cuda):
    """KA1 one-hot, KA2 uniform, KA3 shift invariance, KA5 channel order."""
    spec = ModelSpec(50, 16, 'h36m')
    sk = spec.skeleton
    s, dd, j = spec.heatmap_side, spec.depth, sk.n_head
    lrc, half = 239, 8
    mm = lambda c01: (c01 * lrc + half) * 2200.0 / 256
    logits = np.full((2, s, s, dd * j), -50.0, np.float32)
    pos = {}
    rng = np.random.default_rng(3)
    for jj in range(j):
        h, w, d_ = rng.integers(0, s), rng.integers(0, s), rng.integers(0, dd)
        pos[jj] = (w, h, d_)
        logits[0, h, w, d_ * j + jj] = 60.0                # channel = d*J + j (volumetric.py:231)
    logits[1] = 1.25                                        # uniform
    for precise in (0, 1, 2):
        got = H.run_softargmax(lib, cuda, spec, logits, precise)
        head = np.array([[mm(pos[jj][0] / (s - 1)), mm(pos[jj][1] / (s - 1)), pos[jj][2] / (dd - 1) * 2200.0]
                         for jj in range(j)])
        exp0 = (head - head[-1])[list(sk.permutation)]
        assert np.abs(got[0] - exp0).max() < 1e-2
        assert np.abs(got[1]).max() < 1e-3                 # all joints at the centre -> root-relative 0
        shifted = H.run_softargmax(lib, cuda, spec, logits + 7.5, precise)
        assert np.abs(shifted - got).max() < 1e-3
    assert (got[:, 0, :] == 0).all()                        # h36m: output row 0 is the root (KA10)


def test_softargmax_online_rescale_branch(lib, cuda):
    """Monotonically increasing logits force the running-max rescale on every step."""
    from oracle.forward import logits_to_output
    spec = ModelSpec(50, 8, 'h36m')
    s, c = spec.heatmap_side, spec.n_head_channels
    base = np.arange(s * s, dtype=np.float32).reshape(1, s, s, 1) * 0.05
    logits = np.broadcast_to(base, (1, s, s, c)).copy()
    logits += np.random.default_rng(0).standard_normal(logits.shape).astype(np.float32) * 0.01
    for precise in (0, 1, 2):
        got = H.run_softargmax(lib, cuda, spec, logits, precise)
        ref = logits_to_output(H.oracle_spec(spec), logits).numpy()
        assert np.abs(got - ref).max() <= (1e-3 if precise else 2e-3)


def test_error_reporting(lib):
    d = H.conv_desc(1, 8, 12, 8, 64, 1)     # c_in not a multiple of 8
    st = lib.metro_conv_f16(C.byref(d), C.c_void_p(256), C.c_void_p(256), C.c_void_p(256), None, None, None,
                            C.c_void_p(256), None)
    assert st == -1 and b'multiple of 8' in lib.metro_last_error()


# ---- 256 x 256 x 64 GEMM kernels: conv_gemm4w.hip (product) + the two experimental forms (libmetro_experimental.so) --------
# (name, images of 16x16, c_in, c_out, variant)
G8_CASES = [('k128', 2, 128, 256, 'plain'), ('k256_relu', 3, 256, 512, 'relu'), ('k512_pro', 2, 512, 256, 'prologue'),
            ('k1024_pro_relu', 5, 1024, 512, 'prologue+relu'), ('k512_res', 3, 512, 768, 'residual'),
            ('k2048_pro', 2, 2048, 256, 'prologue'), ('pair_k512', 3, 512, 1280, 'pair'), ('pair_k256', 2, 256, 512, 'pair'),
            ('pair512_k1024', 2, 1024, 2560, 'pair512'),        # block4/unit_1: shortcut 2048 + conv1 512 (gemm4w only)
            # round 6, the batch-64 shapes of block4: 128 whole tiles -> 256 HALF tiles (256 cout x 128 px) in conv_gemm4w; the pair as
            # 512 whole + 256 half tiles in one grid (the other kernels run their own tiling of the same shape: same bits)
            ('k2048_pro_half_tiles', 64, 2048, 512, 'prologue'), ('pair512_k1024_mixed_tiles', 64, 1024, 2560, 'pair512')]


@pytest.mark.parametrize('case', G8_CASES, ids=[c[0] for c in G8_CASES])
@pytest.mark.parametrize('kernel', ['gemm8p', 'gemm4w', 'gemm4d', 'gemm4d_geo1', 'gemm4d_geo2'])
def test_conv_gemm_experimental(lib, cuda, case, kernel):
    """Every element against fp64 on the same fp16 operands: 2e-3 of the layer maximum (fp16 output rounding), for the
    plain / ReLU / pre-activation / shortcut epilogues and the fused shortcut+conv1 pair routing (reference
    resnet_v2.py:119-138), K from 2 to 32 tiles, several tiles per launch; repeated launches are bit-identical (the
    ring is ordered by hand-counted s_waitcnt vmcnt(4) + barriers between two wave groups a barrier apart)."""
    name, n, c_in, c_out, variant = case
    xlib = _lib.load_experimental()
    c2 = 512 if variant == 'pair512' else 256
    if variant == 'pair512':
        if kernel != 'gemm4w':
            pytest.skip('a 512-wide second output exists in conv_gemm4w only')
        variant = 'pair'
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    x, w, b = _mk(rng, n, 16, c_in, c_out, 1)
    x16, w16 = x.astype(np.float16), w.astype(np.float16)
    pro = res = None
    if 'prologue' in variant or variant == 'pair':
        pro = (rng.uniform(0.5, 1.5, c_in).astype(np.float16), (rng.standard_normal(c_in) * 0.3).astype(np.float16))
    if variant == 'residual':
        res = rng.standard_normal((n, 16, 16, c_out)).astype(np.float16)
    relu = 'relu' in variant
    split = c_out - c2 if variant == 'pair' else 0
    d = H.conv_desc(n, 16, c_in, 16, c_out, 1, prologue=pro is not None, relu=relu, residual=res is not None, res_h=16)
    tx = torch.from_numpy(x16).to(cuda)
    tw = torch.from_numpy(np.ascontiguousarray(w16.reshape(c_out, c_in))).to(cuda)
    tb = torch.from_numpy(b).to(cuda)
    ts = torch.from_numpy(pro[0]).to(cuda) if pro else None
    tsh = torch.from_numpy(pro[1]).to(cuda) if pro else None
    tr = torch.from_numpy(res).to(cuda) if res is not None else None
    c1 = split if split else c_out
    out = torch.full((n, 16, 16, c1), float('nan'), dtype=torch.float16, device=cuda)
    out2 = torch.full((n, 16, 16, c2), float('nan'), dtype=torch.float16, device=cuda) if split else None

    if '_geo' in kernel:                                     # conv_gemm4d.hip on 128 x 128 / 128 x 256 block tiles
        geo = int(kernel[-1])

        def entry(*args):
            return xlib.metro_conv_f16_gemm4d_geo(*args[:-1], geo, args[-1])
    else:
        # conv_gemm4w.hip (the product's kernel: four waves of 128 x 128, register-staged operands) or an experimental form
        entry = getattr(lib if kernel == 'gemm4w' else xlib, f'metro_conv_f16_{kernel}')

    def run():
        check(entry(C.byref(d), H.ptr(tx), H.ptr(tw), H.ptr(tb), H.ptr(ts), H.ptr(tsh), H.ptr(tr),
                    H.ptr(out), split, H.ptr(out2), C.c_void_p(0)), f'metro_conv_f16_{kernel}')
        torch.cuda.synchronize()
        return out.clone(), (out2.clone() if split else None)

    g1, g2 = run()
    xin = x16.astype(np.float64)
    if pro is not None:       # fp16 FMA + ReLU, one rounding (the kernel's v_pk_fma_f16)
        xin = np.maximum((xin * pro[0].astype(np.float64) + pro[1].astype(np.float64)).astype(np.float16).astype(np.float64), 0)
    y = xin.reshape(-1, c_in) @ w16.reshape(c_out, c_in).astype(np.float64).T + b.astype(np.float64)
    y = y.reshape(n, 16, 16, c_out)
    if split:
        want1, want2 = y[..., :split], np.maximum(y[..., split:], 0)
    else:
        want1, want2 = (np.maximum(y, 0) if relu else y), None
        if res is not None:
            want1 = want1.astype(np.float16).astype(np.float64) + res.astype(np.float64)   # fp16(conv + bias), then the fp16 Add
    for got, want in ((g1, want1), (g2, want2)):
        if want is None:
            continue
        got = got.cpu().double().numpy()
        assert np.isfinite(got).all()
        err = np.abs(got - want).max() / np.abs(want).max()
        assert err <= 2e-3, (name, err)
    junk = torch.empty(32 << 20, dtype=torch.uint8, device=cuda)
    for it in range(6):                               # race screen: bits must not depend on timing
        if it % 2:
            junk.fill_(it)
        h1, h2 = run()
        assert torch.equal(h1, g1) and (not split or torch.equal(h2, g2)), f'{name}: launch {it} differs'
    if kernel != 'gemm8p' and c2 == 256:              # same K order, one fp32 accumulator per output: the kernels give the same bits
        check(xlib.metro_conv_f16_gemm8p(C.byref(d), H.ptr(tx), H.ptr(tw), H.ptr(tb), H.ptr(ts), H.ptr(tsh), H.ptr(tr),
                                        H.ptr(out), split, H.ptr(out2), C.c_void_p(0)), 'metro_conv_f16_gemm8p')
        torch.cuda.synchronize()
        assert torch.equal(out, g1) and (not split or torch.equal(out2, g2)), f'{name}: {kernel} and gemm8p differ'


def test_conv_gemm4w_rejects_partial_tiles(lib, cuda):
    d = H.conv_desc(1, 8, 512, 8, 256, 1)             # 64 pixels: not a whole 256-pixel tile
    t = torch.zeros(1 << 20, dtype=torch.float16, device=cuda)
    st = lib.metro_conv_f16_gemm4w(C.byref(d), H.ptr(t), H.ptr(t), H.ptr(t.float()), None, None, None, H.ptr(t), 0, None, C.c_void_p(0))
    assert st == -2 and b'pixels' in lib.metro_last_error()
