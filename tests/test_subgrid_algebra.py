"""The identity behind conv3x3_f16_slab's sub-grid pixel order (round 6), on the CPU: a rate-d 3x3 convolution with SAME padding (reference
resnet_utils.conv2d_same, rate > 1: resnet_utils.py:120-123) on an H x W map equals d*d independent rate-1 SAME convolutions on the
(H/d) x (W/d) sub-images of the pixels with equal (y mod d, x mod d) -- and `pixel_of`, the one function the kernel uses to turn a position in
sub-image order into an NHWC pixel index, is a bijection that sends sub-image (n, sy, sx), row vy, column vx to pixel (n, vy*d + sy, vx*d + sx)."""
import numpy as np
import pytest


def pixel_of(g, h, w, d):
    """csrc/conv3x3_f16_slab.hip: slab_tile::pixel_of, restated."""
    gh, gw = h // d, w // d
    hw = gh * gw
    vimg, rem = divmod(g, hw)
    vy, vx = divmod(rem, gw)
    img, sub = divmod(vimg, d * d)
    sy, sx = divmod(sub, d)
    return (img * h + vy * d + sy) * w + vx * d + sx


def conv3x3_same(x, wt, rate):
    """x [n,h,w,c], wt [co,3,3,c]: zero padding `rate`, taps at +-rate (TF SAME for an odd dilated kernel at stride 1)."""
    n, h, w, c = x.shape
    xp = np.zeros((n, h + 2 * rate, w + 2 * rate, c))
    xp[:, rate:rate + h, rate:rate + w] = x
    out = np.zeros((n, h, w, wt.shape[0]))
    for r in range(3):
        for s in range(3):
            out += np.einsum('nhwc,oc->nhwo', xp[:, r * rate:r * rate + h, s * rate:s * rate + w], wt[:, r, s])
    return out


@pytest.mark.parametrize('h,w,d', [(16, 16, 2), (64, 64, 4), (64, 64, 8), (32, 32, 4), (16, 16, 8), (24, 40, 4)])
def test_pixel_of_is_the_subgrid_bijection(h, w, d):
    n = 3
    g = np.arange(n * h * w)
    p = np.array([pixel_of(int(i), h, w, d) for i in g])
    assert sorted(p.tolist()) == g.tolist()                                   # a permutation of the NHWC pixels
    gh, gw = h // d, w // d
    y, x = (p % (h * w)) // w, p % w
    vimg = g // (gh * gw)
    assert np.array_equal(p // (h * w), vimg // (d * d))                       # same image
    assert np.array_equal(y % d, (vimg % (d * d)) // d) and np.array_equal(x % d, (vimg % (d * d)) % d)   # the sub-image's residue class
    assert np.array_equal(y // d, (g % (gh * gw)) // gw) and np.array_equal(x // d, g % gw)               # row-major inside the sub-image


@pytest.mark.parametrize('h,d', [(16, 2), (16, 4), (16, 8), (24, 4)])
def test_dilated_conv_is_d_squared_plain_convs_on_the_subimages(h, d):
    rng = np.random.default_rng(h * 10 + d)
    n, c, co = 2, 5, 4
    x = rng.standard_normal((n, h, h, c))
    wt = rng.standard_normal((co, 3, 3, c))
    want = conv3x3_same(x, wt, d)
    # gather the sub-images in the kernel's order, run the RATE-1 convolution on them, scatter back through pixel_of
    gh = h // d
    perm = np.array([pixel_of(i, h, h, d) for i in range(n * h * h)])
    sub = x.reshape(n * h * h, c)[perm].reshape(n * d * d, gh, gh, c)
    got_sub = conv3x3_same(sub, wt, 1).reshape(n * h * h, co)
    got = np.empty((n * h * h, co))
    got[perm] = got_sub
    assert np.abs(got.reshape(n, h, h, co) - want).max() <= 1e-12 * np.abs(want).max()
