"""Every kernel INSTANTIATION a BASELINE configuration dispatches at its per-GPU batch, against the fp64 reference.

Which kernel a layer runs on depends on its shape and on the batch (tile counts against the 256 CUs): parity shown for a
layer at n = 1 says nothing about the instantiation the same layer gets at n = 32.  `metro_plan_layer_kernel` names the
instantiation (a dry run of the dispatch, no device needed), so this file can

  * (CPU) enumerate the ids of C1..C5 (+ the north star's batch 256) without a GPU, and
  * (`-m gpu`) for EVERY distinct id of every configuration take the first layer that dispatches it and run that layer's
    real shape at the configuration's real batch through the single-kernel C-ABI entry point, check that the entry
    point launched exactly the instantiation the plan names (`metro_last_kernel_id`), and compare with the fp64 reference
    on the same fp16 operands (2e-3 of the layer maximum = fp16 output rounding; reference resnet_v2.py:119-138,219-236,
    resnet_utils.py:82-135, volumetric.py:227-235).  The batch is PERIODIC (image i = image i mod 4): the reference is
    computed for one period and every other image must carry the same bits as its twin, so each tile position of the
    launch is held to the oracle at the cost of four images.
"""
import ctypes as C
import zlib

import numpy as np
import pytest
import torch

from metro_pose3d_amd import ModelSpec, _lib
from metro_pose3d_amd._lib import check
from metro_pose3d_amd.engine import Engine
from tests import helpers as H

# BASELINE.json configs at their per-GPU batch (configs[2..4] are sharded 8 ways: 512/8, 256/8, 128/8) + the north star's batch
CONFIGS = {
    'C1-rn50-s32-J17-b1': (ModelSpec(50, 32, 'h36m'), 1),
    'C2-rn50-s16-J17-b64': (ModelSpec(50, 16, 'h36m'), 64),
    'C2-rn50-s16-J17-b256': (ModelSpec(50, 16, 'h36m'), 256),
    'C3-rn50-s16-J19-b64': (ModelSpec(50, 16, 'many19'), 64),
    'C3-rn50-s16-J19-b256': (ModelSpec(50, 16, 'many19'), 256),       # 152 head channels: the 160 x 256 head (whole K per wave)
    'C4-rn101-s8-J19-b32': (ModelSpec(101, 8, 'many19'), 32),
    'C5-rn50-s4-J17-b16': (ModelSpec(50, 4, 'h36m'), 16),
    # not BASELINE configurations: head shapes the ring head kernel would otherwise never see in a test -- depth 4 (the generic
    # statistics path: five-joint batches are for depth 8) and 128-pixel tiles with 144 weight rows (17 joints at stride 8)
    'X-rn50-s16-J17-D4-b64': (ModelSpec(50, 16, 'h36m', depth=4), 64),
    'X-rn50-s8-J17-b32': (ModelSpec(50, 8, 'h36m'), 32),
    # estimate_pose's middle bucket: block4's conv1 on conv_gemm4w QUARTER tiles (256 cout x 64 px, round 6), block4's pair at 1.25 rounds
    'X-rn50-s16-J17-b32': (ModelSpec(50, 16, 'h36m'), 32),
    # the released `many_*` exports: the 53-joint `merged` head = 424 channels (reference data/datasets.py:142-154, main.py:119-127)
    # in three joint groups on the ring head (round 5), 64- and 128-pixel tiles
    'X-rn50-s16-merged53-b64': (ModelSpec(50, 16, 'merged'), 64),
    'X-rn101-s8-merged53-b32': (ModelSpec(101, 8, 'merged'), 32),
}
PERIOD = 4


def dispatch_table(spec, n):
    """[(layer index, MetroLayerInfo, kernel id)] of an f16 plan at batch n -- no GPU needed."""
    eng = Engine(spec, None, 'f16', max_batch=n)
    return list(zip(range(10 ** 6), eng.layer_infos(), eng.layer_kernels(n)))


def _first_layers_by_id():
    out = []
    for cname, (spec, n) in CONFIGS.items():
        seen = set()
        for i, li, kid in dispatch_table(spec, n):
            if kid not in seen:
                seen.add(kid)
                out.append(pytest.param(cname, i, kid, id=f'{cname}:{li.name.decode()}:{kid}'))
    return out


try:
    _CASES = _first_layers_by_id()
except Exception as e:  # noqa: BLE001  (library not built: the CPU test below reports it)
    _CASES = [pytest.param(None, -1, str(e), id='library-missing')]


# ---- CPU ---------------------------------------------------------------------------------------------------------------
def test_every_layer_names_its_kernel_without_a_gpu():
    ids = {}
    for cname, (spec, n) in CONFIGS.items():
        table = dispatch_table(spec, n)
        assert all(kid and '<' in kid or kid in ('conv3x3_c64', 'prep_input_f16') for _, _, kid in table), [k for _, _, k in table]
        ids[cname] = {li.name.decode(): kid for _, li, kid in table}
    # the choice depends on the batch: RN101-s8 block3 conv2 (256 ch, rate 2, 32x32) takes 128-cout x 256-px tiles with a
    # 64-row halo at batch 32 and 64-cout tiles at batch 1
    assert ids['C4-rn101-s8-J19-b32']['block3/unit_2/conv2'] == 'conv3x3_f16_slab<128x256,rows384,bufs2,tps1,kc64,ws3>'
    small = {li.name.decode(): kid for _, li, kid in dispatch_table(ModelSpec(101, 8, 'many19'), 1)}
    assert small['block3/unit_2/conv2'] != ids['C4-rn101-s8-J19-b32']['block3/unit_2/conv2']
    # round 6: the rate-4 / rate-8 3x3 layers of strides 4 and 8 run on the tap-reuse kernel in sub-grid pixel order (they fell to the ring kernel
    # before); block4's conv1 at 64 crops on conv_gemm4w HALF tiles, its shortcut + conv1 pair as whole + half tiles in one grid; at 32 crops
    # QUARTER tiles; RN101-s8's block3 conv1 at 32 crops on half tiles
    assert ids['C5-rn50-s4-J17-b16']['block4/unit_2/conv2'] == 'conv3x3_f16_slab<128x512,rows640,bufs2,tps1,kc32,ws4>+subgrid'
    assert ids['C5-rn50-s4-J17-b16']['block3/unit_2/conv2'].endswith('+subgrid') and ids['C4-rn101-s8-J19-b32']['block4/unit_2/conv2'].endswith('+subgrid')
    assert not any(k.startswith('conv_igemm_f16_dma') for n_, k in ids['C5-rn50-s4-J17-b16'].items() if n_.endswith('/conv2') and ('block3' in n_ or 'block4' in n_))
    assert '+subgrid' not in ids['C2-rn50-s16-J17-b64']['block4/unit_2/conv2']        # rate 2 on 16 x 16: the plain order (measured faster)
    assert ids['C2-rn50-s16-J17-b64']['block4/unit_2/conv1'] == 'conv_gemm4w<256x128,pro>'
    assert ids['C2-rn50-s16-J17-b64']['block4/unit_1/shortcut+conv1'] == 'conv_gemm4w<256x256,pro>+pair & conv_gemm4w<256x128,pro>+pair'
    assert ids['C2-rn50-s16-J17-b256']['block4/unit_2/conv1'] == 'conv_gemm4w<256x256,pro>'
    assert ids['C2-rn50-s16-J17-b256']['block4/unit_1/shortcut+conv1'] == 'conv_gemm4w<256x256,pro>+pair'
    assert ids['X-rn50-s16-J17-b32']['block4/unit_2/conv1'] == 'conv_gemm4w<256x64,pro>'
    assert ids['C4-rn101-s8-J19-b32']['block3/unit_2/conv1'] == 'conv_gemm4w<256x128,pro>'
    # the one-launch head and its finalize
    # (head_f16.hip: tile width by the number of tiles, weight rows by the head's channels, K-parts per wave group)
    assert ids['C1-rn50-s32-J17-b1']['logits'] == ids['C2-rn50-s16-J17-b64']['logits'] == 'head_f16<144x64,k4>'
    assert ids['C3-rn50-s16-J19-b64']['logits'] == 'head_f16<160x64,k4>'
    assert ids['C4-rn101-s8-J19-b32']['logits'] == 'head_f16<160x128,k4>'
    assert ids['C2-rn50-s16-J17-b256']['logits'] == ids['C5-rn50-s4-J17-b16']['logits'] == 'head_f16<144x256,k2>'
    assert ids['C3-rn50-s16-J19-b256']['logits'] == 'head_f16<160x256>'
    assert ids['X-rn50-s16-J17-D4-b64']['logits'] == 'head_f16<144x64,k4>' and ids['X-rn50-s8-J17-b32']['logits'] == 'head_f16<144x128,k4>'
    assert ids['X-rn50-s16-merged53-b64']['logits'] == 'head_f16<160x64,k4,g3>' and ids['X-rn101-s8-merged53-b32']['logits'] == 'head_f16<160x128,k4,g3>'
    assert ids['X-rn50-s16-merged53-b64']['softargmax'] == 'softargmax_finalize<acc32>'
    assert ids['C2-rn50-s16-J17-b64']['softargmax'] == 'softargmax_finalize<acc32>'
    # parity modes name their kernels too
    e64 = Engine(ModelSpec(50, 16, 'h36m'), None, 'f64', max_batch=2)
    k64 = e64.layer_kernels(2)
    assert k64[0].startswith('conv_igemm_f64acc<') and k64[-1] == 'softargmax_partial<acc64,logits64> & softargmax_finalize<acc64>'
    k32 = Engine(ModelSpec(50, 16, 'h36m'), None, 'f32m', max_batch=2).layer_kernels(2)
    assert k32[0] == 'conv_igemm_f32<64x128,bk32>' and k32[3].startswith('conv_igemm_f32<') and ',v4' in k32[3]
    assert k32[-1] == 'softargmax_partial<acc64,logits32> & softargmax_finalize<acc64>'


@pytest.mark.parametrize('nb', [3, 8], ids=['64-pixel-tiles', '256-pixel-tiles'])
def test_head_partials_slot_covers_large_heat_maps(lib, nb):
    """The one-launch head writes one fp32 record per (image, slab, joint): one per 64 pixels, and one per 32 pixels once the
    launch has >= 256 tiles of 256 pixels (head_f16<160x256>) -- more slabs than the two-launch path's cap of 64 once the heat
    map has > 4096 pixels (proc_side 384 at stride 4 = 96 x 96).  The partials slot (followed only by the 4-byte-per-image
    status words) must hold what the library itself says a launch at this batch writes."""
    spec = ModelSpec(50, 4, 'h36m', proc_side=384)
    eng = Engine(spec, None, 'f16', max_batch=nb)
    infos = eng.layer_infos()
    logits = next(li for li in infos if li.name == b'logits')
    kern = eng.layer_kernels(nb)[infos.index(logits)]
    assert kern == 'head_f16<144x256,k2>' if nb == 8 else kern.startswith('head_f16<') and 'x256' not in kern, kern
    side, j = 96, spec.skeleton.n_head
    after_logits = logits.out_offset + logits.out_bytes_per_image * nb
    need = lib.metro_head_f16_scratch_bytes(nb, side, j)
    assert need >= nb * (side * side // 32) * j * 5 * 4          # a record per 32 pixels
    status = -(-nb * 4 // 256) * 256
    assert eng.workspace_bytes - after_logits - status >= need, (eng.workspace_bytes - after_logits - status, need)


# ---- GPU ---------------------------------------------------------------------------------------------------------------
def _periodic(gen, n, shape, cuda, scale=1.0, relu=False):
    """fp16 [n, *shape] on the device with image i == image i % PERIOD."""
    p = min(PERIOD, n)
    base = torch.randn((p,) + tuple(shape), generator=gen, device=cuda, dtype=torch.float32) * scale
    if relu:
        base = base.clamp_min(0)
    base = base.half()
    reps = (n + p - 1) // p
    return base.repeat((reps,) + (1,) * len(shape))[:n].contiguous(), base.cpu().numpy()


def _assert_periodic(out, n, what):
    p = min(PERIOD, n)
    for i in range(p, n, p):
        k = min(p, n - i)
        assert torch.equal(out[i:i + k], out[:k]), f'{what}: images {i}..{i + k - 1} differ from their twins 0..{k - 1}'


def _noted(lib):
    return lib.metro_last_kernel_id().decode().split(' & ')


def _close(got, ref, what, tol=2e-3):
    got = np.asarray(got, np.float64)
    assert np.isfinite(got).all(), what
    err, scale = np.abs(got - ref).max(), np.abs(ref).max()
    assert err <= tol * scale, (what, err, scale)


@pytest.mark.gpu
@pytest.mark.parametrize('cname,layer,kid', _CASES)
def test_production_dispatch_against_fp64_reference(lib, cuda, cname, layer, kid):
    assert cname is not None, f'libmetro_hip.so could not be loaded at collection time: {kid}'
    spec, n = CONFIGS[cname]
    _, li, kid2 = dispatch_table(spec, n)[layer]
    assert kid2 == kid
    name = li.name.decode()
    gen = torch.Generator(device=cuda)
    gen.manual_seed(zlib.crc32(f'{cname}/{name}'.encode()))
    rng = np.random.default_rng(zlib.crc32(f'{cname}/{name}/w'.encode()))
    dev = lambda a, dt: torch.from_numpy(np.ascontiguousarray(np.asarray(a).astype(dt))).to(cuda)
    p = min(PERIOD, n)
    check(lib.metro_kernel_notes(1), 'metro_kernel_notes')
    try:
        if name == 'softargmax':
            pytest.skip('launched (and compared) together with the head: see the logits case of this configuration')
        if name == 'conv1+pool1':
            _stem(lib, cuda, li, n, gen, rng, dev, kid)
        elif name == 'logits' and kid.startswith('head_f16'):
            _head(lib, cuda, spec, li, n, gen, rng, dev, kid)
        else:
            _conv(lib, cuda, li, n, gen, rng, dev, kid, name)
    finally:
        lib.metro_kernel_notes(0)


def _stem(lib, cuda, li, n, gen, rng, dev, kid):
    side = 4 * li.h_out
    p = min(PERIOD, n)
    base = torch.rand((p, side, side, 3), generator=gen, device=cuda, dtype=torch.float32)
    img = base.repeat(((n + p - 1) // p, 1, 1, 1))[:n].contiguous()
    w = (rng.standard_normal((64, 7, 7, 3)) * np.sqrt(2.0 / 147)).astype(np.float16)
    b = (rng.standard_normal(64) * 0.5).astype(np.float32)
    wp = np.zeros((64, 7, 8, 4), np.float16)
    wp[:, :, :7, :3] = w
    tw, tb = dev(wp, np.float16), dev(b, np.float32)
    out = torch.full((n, side // 4, side // 4, 64), float('nan'), dtype=torch.float16, device=cuda)
    check(lib.metro_stem_pool_f32in(H.ptr(img), H.ptr(tw), H.ptr(tb), H.ptr(out), n, side, None), 'metro_stem_pool_f32in')
    torch.cuda.synchronize()
    assert _noted(lib) == [kid], (_noted(lib), kid)
    _assert_periodic(out, n, kid)
    xi = torch.from_numpy(base.cpu().numpy().astype(np.float16).astype(np.float64)).permute(0, 3, 1, 2)
    conv = torch.nn.functional.conv2d(torch.nn.functional.pad(xi, (3, 3, 3, 3)), torch.from_numpy(w.astype(np.float64)).permute(0, 3, 1, 2),
                                      torch.from_numpy(b.astype(np.float64)), stride=2).half().double()
    want = torch.nn.functional.max_pool2d(torch.nn.functional.pad(conv, (1, 1, 1, 1)), 3, 2).permute(0, 2, 3, 1).numpy()
    _close(out[:p].cpu().numpy(), want, kid)


def _head(lib, cuda, spec, li, n, gen, rng, dev, kid):
    from oracle.forward import logits_to_output
    side, k, c = li.h_in, li.c_in, li.c_out
    x, xb = _periodic(gen, n, (side, side, k), cuda)
    w = (rng.standard_normal((c, k)) * np.sqrt(2.0 / k) * 2.0).astype(np.float16)
    b = (rng.standard_normal(c) * 0.1).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, k).astype(np.float16)
    sh = (rng.standard_normal(k) * 0.2).astype(np.float16)
    tw, tb, ts, tsh = dev(w, np.float16), dev(b, np.float32), dev(sc, np.float16), dev(sh, np.float16)
    cs = spec.to_c(_lib.METRO_PREC_F16)
    scratch = torch.empty(lib.metro_head_f16_scratch_bytes(n, side, spec.skeleton.n_head), dtype=torch.uint8, device=cuda)
    logits = torch.full((n, side, side, c), float('nan'), dtype=torch.float32, device=cuda)
    poses = torch.full((n, spec.skeleton.n_out, 3), float('nan'), dtype=torch.float32, device=cuda)
    check(lib.metro_head_f16(H.ptr(x), H.ptr(tw), H.ptr(tb), H.ptr(ts), H.ptr(tsh), n, k, C.byref(cs), H.ptr(scratch), H.ptr(logits),
                             H.ptr(poses), None), 'metro_head_f16')
    torch.cuda.synchronize()
    assert _noted(lib) == [kid, 'softargmax_finalize<acc32>'], _noted(lib)
    _assert_periodic(logits, n, kid)
    _assert_periodic(poses, n, kid)
    p = xb.shape[0]
    xin = np.maximum((xb.astype(np.float64) * sc.astype(np.float64) + sh.astype(np.float64)).astype(np.float16).astype(np.float64), 0)
    ref = xin.reshape(-1, k) @ w.astype(np.float64).T + b.astype(np.float64)
    ref = ref.reshape(p, side, side, c)
    got = logits[:p].cpu().double().numpy()
    assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max(), np.abs(got - ref).max() / np.abs(ref).max()
    want = logits_to_output(H.oracle_spec(spec), ref).numpy()
    d = np.abs(poses[:p].cpu().numpy() - want).max()
    assert d <= 2e-3, f'{kid}: poses {d} mm from the exact soft-argmax of the exact logits'


def _conv(lib, cuda, li, n, gen, rng, dev, kid, name):
    pair = name.endswith('/shortcut+conv1')
    nxt = '/conv3+' in name
    c_in, c1 = li.c_in, li.c_out
    c_out = c1 + li.out2_channels if pair else c1
    k, h_in, h_out = li.kh, li.h_in, li.h_out
    assert li.kh == li.kw and li.h_in == li.w_in and li.pad_top == li.pad_left
    f32out = li.out_dtype == _lib.METRO_F32
    rebuilt = bool(li.fused_flags & _lib.FUSED_REBUILT_SHORTCUT)      # the shortcut is rebuilt in the launch: no residual tensor
    compact = bool(li.fused_flags & _lib.FUSED_COMPACT_SHORTCUT)      # the sub-sampled shortcut arrives as a compact tensor
    has_res = bool(li.has_residual) and not rebuilt
    res_stride, res_offset = (1, 0) if compact else (li.res_stride, li.res_offset)
    res_h = h_out if res_stride == 1 else 2 * h_out
    d = H.conv_desc(n, h_in, c_in, h_out, c_out, k, li.stride, li.dilation, li.pad_top, prologue=bool(li.has_prologue),
                    relu=bool(li.relu), residual=has_res, res_h=res_h, res_stride=res_stride,
                    res_offset=res_offset, out_dtype=_lib.METRO_F32 if f32out else _lib.METRO_F16, in_dtype=_lib.METRO_F16)
    x, xb = _periodic(gen, n, (h_in, h_in, c_in), cuda, relu=not li.has_prologue and k == 3)
    w = (rng.standard_normal((c_out, k, k, c_in)) * np.sqrt(2.0 / (k * k * c_in))).astype(np.float16)
    b = (rng.standard_normal(c_out) * 0.1).astype(np.float32)
    tw, tb = dev(w, np.float16), dev(b, np.float32)
    ts = tsh = tr = None
    pro = None
    if li.has_prologue:
        pro = (rng.uniform(0.5, 1.5, c_in).astype(np.float16), (rng.standard_normal(c_in) * 0.2).astype(np.float16))
        ts, tsh = dev(pro[0], np.float16), dev(pro[1], np.float16)
    rb = None
    if has_res:
        tr, rb = _periodic(gen, n, (res_h, res_h, c_out), cuda)
    out = torch.full((n, h_out, h_out, c1), float('nan'), dtype=torch.float32 if f32out else torch.float16, device=cuda)
    out2 = None
    if li.fused_flags & _lib.FUSED_CONV1_IN_FRONT:
        return _conv1_conv2(lib, cuda, li, n, d, x, xb, tw, tb, w, b, out, rng, dev, kid)
    psc = None
    if li.fused_flags & (_lib.FUSED_PROJECTION_SHORTCUT | _lib.FUSED_REBUILT_SHORTCUT):
        assert nxt, 'the in-launch projection shortcut exists in the conv3 + next conv1 launch only'
        cx = 64                                   # the unit's raw input (block1/unit_1: the pooled stem output)
        xs, xsb = _periodic(gen, n, (h_out, h_out, cx), cuda)
        wsc = (rng.standard_normal((c1, cx)) * np.sqrt(2.0 / cx)).astype(np.float16)
        bsc = (rng.standard_normal(c1) * 0.1).astype(np.float32)
        psc_s = rng.uniform(0.5, 1.5, cx).astype(np.float16)
        psc_b = (rng.standard_normal(cx) * 0.2).astype(np.float16)
        psc = [xs, dev(wsc, np.float16), dev(bsc, np.float32), dev(psc_s, np.float16), dev(psc_b, np.float16)]
    if pair:
        out2 = torch.full((n, h_out, h_out, li.out2_channels), float('nan'), dtype=torch.float16, device=cuda)
        check(lib.metro_conv_f16_pair(C.byref(d), H.ptr(x), H.ptr(tw), H.ptr(tb), H.ptr(ts), H.ptr(tsh), H.ptr(out), c1, H.ptr(out2), None),
              'metro_conv_f16_pair')
        if kid.startswith('conv_pw64<k256'):   # block2's pair in the weight-resident kernel and in the ring kernel it replaced: the same bits
            torch.cuda.synchronize()
            mine, mine2 = out.clone(), out2.clone()
            check(lib.metro_conv_b1_form(1), 'metro_conv_b1_form')
            try:
                check(lib.metro_conv_f16_pair(C.byref(d), H.ptr(x), H.ptr(tw), H.ptr(tb), H.ptr(ts), H.ptr(tsh), H.ptr(out), c1, H.ptr(out2), None),
                      'metro_conv_f16_pair (ring kernel)')
                torch.cuda.synchronize()
            finally:
                lib.metro_conv_b1_form(0)
            assert _noted(lib)[-1].startswith('conv_igemm_f16_dma<'), _noted(lib)
            assert torch.equal(out, mine) and torch.equal(out2, mine2), f'{kid}: differs from {_noted(lib)[-1]}'
            check(lib.metro_kernel_notes(1), 'metro_kernel_notes')
            check(lib.metro_conv_f16_pair(C.byref(d), H.ptr(x), H.ptr(tw), H.ptr(tb), H.ptr(ts), H.ptr(tsh), H.ptr(out), c1, H.ptr(out2), None),
                  'metro_conv_f16_pair')
    elif nxt:
        c2 = li.out2_channels
        w2 = (rng.standard_normal((c2, c1)) * np.sqrt(2.0 / c1)).astype(np.float16)
        b2 = (rng.standard_normal(c2) * 0.1).astype(np.float32)
        sc2 = rng.uniform(0.5, 1.5, c1).astype(np.float16)
        sh2 = (rng.standard_normal(c1) * 0.2).astype(np.float16)
        t2 = [dev(w2, np.float16), dev(b2, np.float32), dev(sc2, np.float16), dev(sh2, np.float16)]
        out2 = torch.full((n, h_out, h_out, c2), float('nan'), dtype=torch.float16, device=cuda)
        on_chip = bool(li.fused_flags & _lib.FUSED_OUT_ON_CHIP)
        prev = None
        if rebuilt:
            # block1/unit_2: x_1 = fp16(W3_prev . t2_prev + b) + fp16(Wsc . pre(x0) + bsc) is rebuilt in the launch.  The storing form
            # (what metro_forward_upto runs) gives `out`; the form the plan names must give the SAME second output, and its
            # sub-sampled copy must be those pixels of `out`
            tp, tpb = _periodic(gen, n, (h_out, h_out, 64), cuda, relu=True)
            w3p = (rng.standard_normal((c1, 64)) * np.sqrt(2.0 / 64)).astype(np.float16)
            b3p = (rng.standard_normal(c1) * 0.1).astype(np.float32)
            prev = [tp, dev(w3p, np.float16), dev(b3p, np.float32)]
            args = lambda o, osub, soff, o2: (C.byref(d), H.ptr(x), H.ptr(tw), H.ptr(tb), H.ptr(psc[0]), H.ptr(psc[1]), H.ptr(psc[2]), H.ptr(psc[3]),
                                              H.ptr(psc[4]), H.ptr(prev[0]), H.ptr(prev[1]), H.ptr(prev[2]), H.ptr(o), H.ptr(osub), soff, H.ptr(t2[0]),
                                              H.ptr(t2[1]), H.ptr(t2[2]), H.ptr(t2[3]), H.ptr(o2), c2, None)
            # the classic single-role kernel first (what metro_forward_upto runs: the whole sum stored) ...
            check(lib.metro_conv_b1_form(1), 'metro_conv_b1_form')
            try:
                check(lib.metro_conv_f16_next_rebuild(*args(out, None, 0, out2)), 'metro_conv_f16_next_rebuild (classic form)')
                torch.cuda.synchronize()
            finally:
                lib.metro_conv_b1_form(0)
            full, full2 = out.clone(), out2.clone()
            # ... then the form the plan names: same sum (or exactly its sub-sampled pixels), same second output, bit for bit
            check(lib.metro_kernel_notes(1), 'metro_kernel_notes')
            out.fill_(float('nan'))
            out2.fill_(float('nan'))
            if li.out_sub_offset >= 0:
                sub = torch.full((n, li.out_sub_side, li.out_sub_side, c1), float('nan'), dtype=torch.float16, device=cuda)
                check(lib.metro_conv_f16_next_rebuild(*args(None, sub, li.out_sub_off, out2)), 'metro_conv_f16_next_rebuild')
                torch.cuda.synchronize()
                o = li.out_sub_off
                assert torch.equal(sub, full[:, o::2, o::2][:, :li.out_sub_side, :li.out_sub_side]), f'{kid}: sub-sampled copy != pixels of the classic form\'s sum'
                out.copy_(full)
            else:
                check(lib.metro_conv_f16_next_rebuild(*args(out, None, 0, out2)), 'metro_conv_f16_next_rebuild')
                torch.cuda.synchronize()
                assert torch.equal(out, full), f'{kid}: the sum differs between the classic and the producer / consumer form'
            assert torch.equal(out2, full2), f'{kid}: second output differs between the classic and the producer / consumer form'
        elif psc is not None:
            pargs = lambda o, o2: (C.byref(d), H.ptr(x), H.ptr(tw), H.ptr(tb), H.ptr(psc[0]), H.ptr(psc[1]), H.ptr(psc[2]), H.ptr(psc[3]), H.ptr(psc[4]),
                                   H.ptr(o), H.ptr(t2[0]), H.ptr(t2[1]), H.ptr(t2[2]), H.ptr(t2[3]), H.ptr(o2), c2, None)
            if on_chip:           # the plan's form keeps the sum on chip: same second output as the classic storing form
                check(lib.metro_conv_b1_form(1), 'metro_conv_b1_form')
                try:
                    check(lib.metro_conv_f16_next_proj(*pargs(out, out2)), 'metro_conv_f16_next_proj (classic form)')
                    torch.cuda.synchronize()
                finally:
                    lib.metro_conv_b1_form(0)
                full2 = out2.clone()
                check(lib.metro_kernel_notes(1), 'metro_kernel_notes')
                out2.fill_(float('nan'))
                check(lib.metro_conv_f16_next_proj(*pargs(None, out2)), 'metro_conv_f16_next_proj (sum on chip)')
                torch.cuda.synchronize()
                assert torch.equal(out2, full2), f'{kid}: second output differs between the classic storing and the on-chip form'
            else:
                check(lib.metro_conv_f16_next_proj(*pargs(out, out2)), 'metro_conv_f16_next_proj')
        else:
            check(lib.metro_conv_f16_next(C.byref(d), H.ptr(x), H.ptr(tw), H.ptr(tb), H.ptr(tr), H.ptr(out), H.ptr(t2[0]), H.ptr(t2[1]),
                                          H.ptr(t2[2]), H.ptr(t2[3]), H.ptr(out2), c2, None), 'metro_conv_f16_next')
    else:
        check(lib.metro_conv_f16(C.byref(d), H.ptr(x), H.ptr(tw), H.ptr(tb), H.ptr(ts), H.ptr(tsh), H.ptr(tr), H.ptr(out), None),
              'metro_conv_f16')
        if kid.endswith('+subgrid'):
            # the tap-reuse kernel in sub-grid pixel order against the ring kernel these layers ran on until round 6 (the test
            # switch puts them back): two correct fp32 summation orders of the same products (chunk-major vs tap-major) --
            # rounding flips of the fp16 result only

            torch.cuda.synchronize()
            sub = out.clone()
            check(lib.metro_conv_b1_form(1), 'metro_conv_b1_form')
            try:
                check(lib.metro_conv_f16(C.byref(d), H.ptr(x), H.ptr(tw), H.ptr(tb), H.ptr(ts), H.ptr(tsh), H.ptr(tr), H.ptr(out), None),
                      'metro_conv_f16 (classic form)')
                torch.cuda.synchronize()
            finally:
                lib.metro_conv_b1_form(0)
            assert _noted(lib)[-1].startswith('conv_igemm_f16_dma<'), _noted(lib)
            same = (out == sub).float().mean().item()
            worst = (out.float() - sub.float()).abs().max().item()
            assert same >= 0.97 and worst <= 2.0 ** -9 * sub.float().abs().max().item(), (kid, same, worst)
            check(lib.metro_kernel_notes(1), 'metro_kernel_notes')
            check(lib.metro_conv_f16(C.byref(d), H.ptr(x), H.ptr(tw), H.ptr(tb), H.ptr(ts), H.ptr(tsh), H.ptr(tr), H.ptr(out), None), 'metro_conv_f16')
        if kid.startswith('conv_pws<'):        # the skewed kernel and conv_pw64's lock-step one: the same bits
            torch.cuda.synchronize()
            skewed = out.clone()
            noted = _noted(lib)
            check(lib.metro_conv_b1_form(1), 'metro_conv_b1_form')
            try:
                check(lib.metro_conv_f16(C.byref(d), H.ptr(x), H.ptr(tw), H.ptr(tb), H.ptr(ts), H.ptr(tsh), H.ptr(tr), H.ptr(out), None),
                      'metro_conv_f16 (classic form)')
                torch.cuda.synchronize()
            finally:
                lib.metro_conv_b1_form(0)
            assert _noted(lib)[-1].startswith('conv_pw64<'), _noted(lib)
            assert torch.equal(out, skewed), f'{kid}: differs from {_noted(lib)[-1]}'
            check(lib.metro_kernel_notes(1), 'metro_kernel_notes')
            check(lib.metro_conv_f16(C.byref(d), H.ptr(x), H.ptr(tw), H.ptr(tb), H.ptr(ts), H.ptr(tsh), H.ptr(tr), H.ptr(out), None), 'metro_conv_f16')
    torch.cuda.synchronize()
    assert _noted(lib) == kid.split(' & '), f'the entry point launched {_noted(lib)}, the plan names {kid}'     # (a layer may be two launches)
    _assert_periodic(out, n, kid)
    if out2 is not None:
        _assert_periodic(out2, n, kid + ' (second output)')
    p = xb.shape[0]
    xin = xb.astype(np.float64)
    if pro is not None:       # fp16 FMA + ReLU, one rounding (v_pk_fma_f16)
        xin = np.maximum((xin * pro[0].astype(np.float64) + pro[1].astype(np.float64)).astype(np.float16).astype(np.float64), 0)
    ref = H.ref_conv_nhwc(xin, w, b, li.stride, li.dilation, li.pad_top, h_out, relu=bool(li.relu) and not pair).numpy()
    got = out[:p].cpu().double().numpy()
    if pair:
        _close(got, ref[..., :c1], kid)
        _close(out2[:p].cpu().numpy(), np.maximum(ref[..., c1:], 0), kid + ' (second output)')
        return
    if rb is not None:        # fp16(conv + bias), then the fp16 Add of the (sub-sampled, shifted) shortcut
        r = rb.astype(np.float64)[:, res_offset::res_stride, res_offset::res_stride][:, :h_out, :h_out]
        ref = ref.astype(np.float16).astype(np.float64) + r
    if psc is not None:       # fp16(conv3 + bias) + fp16(Wsc . fp16(relu(x * s + b)) + bias_sc): the fp16 Add of resnet_v2.py:138
        xin_s = np.maximum((xsb.astype(np.float64) * psc_s.astype(np.float64) + psc_b.astype(np.float64)).astype(np.float16).astype(np.float64), 0)
        sc = (xin_s @ wsc.astype(np.float64).T + bsc.astype(np.float64)).astype(np.float16).astype(np.float64)
        if rebuilt:           # x_1 = fp16(fp16(W3_prev . t2_prev + b) + projection shortcut), then THIS unit's fp16 Add
            c3p = (tpb.astype(np.float64) @ w3p.astype(np.float64).T + b3p.astype(np.float64)).astype(np.float16).astype(np.float64)
            sc = (c3p + sc).astype(np.float16).astype(np.float64)
        ref = ref.astype(np.float16).astype(np.float64) + sc
    _close(got, ref, kid, tol=2e-5 if f32out else 2e-3)
    if nxt:
        pre = np.maximum((got * sc2.astype(np.float64) + sh2.astype(np.float64)).astype(np.float16).astype(np.float64), 0)
        want2 = np.maximum(pre @ w2.astype(np.float64).T + b2.astype(np.float64), 0)
        _close(out2[:p].cpu().numpy(), want2, kid + ' (second output)')


def _conv1_conv2(lib, cuda, li, n, d, x, xb, tw2, tb2, w2, b2, out, rng, dev, kid):
    """conv1 (1x1 on the pre-activated input, folded BN + ReLU) fused in front of the 3x3: t1 is rounded to fp16 once (in LDS)."""
    c = li.c_in
    w1 = (rng.standard_normal((li.c_out, c)) * np.sqrt(2.0 / c)).astype(np.float16)
    b1 = (rng.standard_normal(li.c_out) * 0.1).astype(np.float32)
    ps = rng.uniform(0.5, 1.5, c).astype(np.float16)
    pb = (rng.standard_normal(c) * 0.2).astype(np.float16)
    t = [dev(w1, np.float16), dev(b1, np.float32), dev(ps, np.float16), dev(pb, np.float16)]
    check(lib.metro_conv_f16_conv1_conv2(C.byref(d), H.ptr(x), H.ptr(t[0]), H.ptr(t[1]), H.ptr(t[2]), H.ptr(t[3]), H.ptr(tw2), H.ptr(tb2),
                                         H.ptr(out), None), 'metro_conv_f16_conv1_conv2')
    torch.cuda.synchronize()
    assert _noted(lib) == kid.split(' & '), f'the entry point launched {_noted(lib)}, the plan names {kid}'     # (a layer may be two launches)
    _assert_periodic(out, n, kid)
    xin = np.maximum((xb.astype(np.float64) * ps.astype(np.float64) + pb.astype(np.float64)).astype(np.float16).astype(np.float64), 0)
    t1 = np.maximum(xin @ w1.astype(np.float64).T + b1.astype(np.float64), 0).astype(np.float16)
    ref = H.ref_conv_nhwc(t1, w2, b2, 1, 1, 1, li.h_out, relu=True).numpy()
    _close(out[:xb.shape[0]].cpu().numpy(), ref, kid)
