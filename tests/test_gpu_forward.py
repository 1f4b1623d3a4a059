"""Whole-path parity on the GPU (`-m gpu`): metro_forward through the C ABI vs the fp64 oracle.

Bars (BASELINE.json north star): parity mode (precision='f64': fp32 images/weights in, fp32
poses out, fp64 arithmetic and storage inside) within 1e-3 mm of the oracle; the fp32-storage
mode sits at the fp32 rounding floor (~1e-3 mm) and is held to 5e-3 mm; fp16
mode is the reference's own default compute dtype (options.py:73): every launch is held to a
rounding flip against the fp16-faithful oracle in tests/test_f16_layerwise.py, and its poses to
the accuracy of that fp16 model of the graph.
"""
import numpy as np
import pytest
import torch

from metro_pose3d_amd import ModelSpec, _lib, synth
from metro_pose3d_amd.engine import Engine
from oracle import forward as OF
from tests import helpers as H

pytestmark = pytest.mark.gpu

TOL_PARITY_MM = 1e-3
TOL_F32_STORAGE_MM = 2e-2   # measured 1e-3..5e-3 on these nets: fp32 storage rounding, not a bug
TOY = [ModelSpec(50, 32, 'h36m', base_width=8), ModelSpec(50, 16, 'many19', base_width=8),
       ModelSpec(50, 8, 'h36m', base_width=8), ModelSpec(50, 4, 'h36m', base_width=8),
       ModelSpec(101, 8, 'merged', base_width=8), ModelSpec(101, 4, 'many19', base_width=8),
       ModelSpec(50, 16, 'h36m', base_width=16, centered_stride=False)]
_id = lambda s: f'rn{s.arch}-s{s.stride}-{s.dataset}-w{s.base_width}' + ('' if s.centered_stride else '-nc')


def _setup(spec, n, gain=3.0, seed=0):
    params = synth.make_params(spec.arch, spec.n_head_channels, spec.base_width, seed=seed, logit_gain=gain)
    images = synth.make_images(n, spec.proc_side)
    return params, images


@pytest.mark.parametrize('spec', TOY, ids=_id)
def test_toy_forward_parity_within_1e3_mm(cuda, spec):
    params, images = _setup(spec, 3)
    ref = OF.forward(H.oracle_spec(spec), params, images, torch.float64).numpy()
    x = torch.from_numpy(images).to(cuda)
    for prec, tol in (('f64', TOL_PARITY_MM), ('f32', TOL_F32_STORAGE_MM)):
        got = Engine(spec, params, prec, max_batch=4, device=cuda).forward(x).cpu().numpy()
        assert np.isfinite(got).all()
        assert np.abs(got - ref).max() <= tol, (prec, np.abs(got - ref).max())
    # the fp32-matrix-core mode accumulates in fp32 like the reference's fp32 graph: held to the distance the CPU fp32
    # restatement of the same graph has to exact math (two samples of the same rounding noise: 3x on the maximum of 51-57 values)
    got = Engine(spec, params, 'f32m', max_batch=4, device=cuda).forward(x).cpu().numpy()
    cpu32 = OF.forward(H.oracle_spec(spec), params, images, torch.float32).numpy().astype(np.float64)
    e, ec = np.abs(got - ref).max(), np.abs(cpu32 - ref).max()
    assert np.isfinite(got).all() and e <= max(3.0 * ec, 2e-3), ('f32m', e, ec)


@pytest.mark.parametrize('spec', TOY[:3], ids=_id)
def test_toy_layerwise(cuda, spec):
    """Every layer output of both precision modes against the oracle's intermediate tensors."""
    params, images = _setup(spec, 2)
    col = {}
    OF.forward(H.oracle_spec(spec), params, images, torch.float64, col)
    x = torch.from_numpy(images).to(cuda)
    for prec, rel in (('f64', 1e-11), ('f32', 2e-6), ('f32m', 2e-5), ('f16', 3e-2)):
        eng = Engine(spec, params, prec, max_batch=2, device=cuda)
        for i, li in enumerate(eng.layer_infos()):
            name = li.name.decode()
            key = {'logits': 'logits', 'conv1': 'conv1', 'pool1': 'pool1', 'conv1+pool1': 'pool1'}.get(name)
            if key is None and (name.endswith('/conv3') or '/conv3+' in name):
                key = name.split('/conv3')[0]                # unit output = shortcut + conv3
            elif key is None and name.endswith('/conv1+conv2'):
                key = name[:-len('conv1+conv2')] + 'conv2'   # conv1 fused in front of conv2 (block1/unit_1 at full width)
            elif key is None and name.endswith(('/conv1', '/conv2')):
                key = name
            if key is None or key not in col:
                continue
            got = eng.forward_upto(x, i).cpu().double().numpy()
            ref = col[key].permute(0, 2, 3, 1).numpy()
            err = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30)
            assert err <= rel, (prec, name, err)


def test_fused_stem_conv_pool(cuda):
    """Stem 7x7/2 conv + zero-padded 3x3/2 max-pool in one persistent launch (full base width): against a
    torch fp64 restatement on the same fp16-rounded operands (reference resnet_v2.py:219-224,
    resnet_utils.py:138-185).  9 images = 576 patches: more than the resident blocks, so blocks walk several
    patches through their double-buffered windows."""
    spec = ModelSpec(50, 32, 'h36m')
    params, images = _setup(spec, 9)
    x = torch.from_numpy(images).to(cuda)
    eng = Engine(spec, params, 'f16', max_batch=9, device=cuda)
    names = [li.name.decode() for li in eng.layer_infos()]
    assert 'conv1+pool1' in names and 'pool1' not in names
    got = eng.forward_upto(x, names.index('conv1+pool1')).cpu().double()
    root = 'MainPart/resnet_v2_50/conv1'
    w = torch.from_numpy(params[root + '/weights'].astype(np.float16).astype(np.float64)).permute(3, 2, 0, 1)   # HWIO -> OIHW
    b = torch.from_numpy(params[root + '/biases'].astype(np.float32).astype(np.float64))
    xi = torch.from_numpy(images.astype(np.float16).astype(np.float64)).permute(0, 3, 1, 2)
    conv = torch.nn.functional.conv2d(torch.nn.functional.pad(xi, (3, 3, 3, 3)), w, b, stride=2)
    conv = conv.half().double()                                   # the fp16 tensor the unfused path stores
    want = torch.nn.functional.max_pool2d(torch.nn.functional.pad(conv, (1, 1, 1, 1)), 3, 2).permute(0, 2, 3, 1)
    assert got.shape == want.shape == (9, 64, 64, 64)
    err = (got - want).abs().max().item() / want.abs().max().item()
    assert err <= 2e-3, err                                       # one fp16 ulp of the largest value
    assert (got == want).double().mean().item() > 0.98           # almost all elements bit-identical


def test_fused_launch_second_outputs(cuda):
    """Launches with two output tensors (projection shortcut + conv1 pairs; block1 conv3 + the NEXT unit's
    conv1 computed from the LDS-resident tile): both tensors against the oracle's intermediates, and the
    second GEMM against a torch restatement on the kernel's own fp16 unit output (tight)."""
    spec = ModelSpec(50, 32, 'h36m')                         # full width: block1 has 256-channel rows
    params, images = _setup(spec, 3)
    col = {}
    OF.forward(H.oracle_spec(spec), params, images, torch.float64, col)
    x = torch.from_numpy(images).to(cuda)
    eng = Engine(spec, params, 'f16', max_batch=3, device=cuda)
    seen = set()
    for i, li in enumerate(eng.layer_infos()):
        name = li.name.decode()
        if li.out2_offset < 0:
            continue
        got2 = eng.forward_upto(x, i, second=True).cpu().double().numpy()
        if '/conv3+' in name:
            unit, nxt = name.split('/conv3+')
            key2 = f'{unit.split("/")[0]}/{nxt}'
            seen.add('conv3+conv1')
            # tight: the second GEMM on the fp16 unit output the same launch stored
            xo = eng.forward_upto(x, i).cpu().double()
            sc = f'MainPart/resnet_v2_50/{key2[:-len("/conv1")]}/bottleneck_v2'
            bn = lambda s, e=1e-5: (params[s + '/gamma'] / np.sqrt(params[s + '/moving_variance'] + e),
                                    params[s + '/beta'], params[s + '/moving_mean'])
            g, b, mu = bn(sc + '/preact')
            pre = torch.relu(xo.half().double() * torch.from_numpy((g).astype(np.float16)).double()
                             + torch.from_numpy((b - mu * g).astype(np.float16)).double())
            pre = pre.half().double()          # the kernel applies the prologue in fp16
            g1, b1, mu1 = bn(sc + '/conv1/BatchNorm')
            w = (params[sc + '/conv1/weights'][0, 0] * g1[None, :]).astype(np.float16).astype(np.float64)   # [cin, cb]
            want = torch.relu(pre @ torch.from_numpy(w) + torch.from_numpy((b1 - mu1 * g1).astype(np.float32)).double())
            err = (torch.from_numpy(got2) - want).abs().max().item() / max(want.abs().max().item(), 1e-30)
            assert err <= 4e-3, (name, err)
        elif name.endswith('/conv1+conv2'):                 # conv1 in front of the 3x3: its output is a dump-only second tensor
            key2 = name[:-len('conv1+conv2')] + 'conv1'
            seen.add('conv1+conv2')
        else:
            key2 = name.replace('shortcut+conv1', 'conv1')
            seen.add('shortcut+conv1')
        ref2 = col[key2].permute(0, 2, 3, 1).numpy()
        err = np.abs(got2 - ref2).max() / max(np.abs(ref2).max(), 1e-30)
        assert err <= 3e-2, (name, 'second output', err)
    assert seen == {'conv3+conv1', 'shortcut+conv1', 'conv1+conv2'}


def test_full_rn50_s16_all_modes(cuda):
    spec = ModelSpec(50, 16, 'h36m')
    params, images = _setup(spec, 2, gain=synth.logit_gain_for(50, 16))
    ref = OF.forward(H.oracle_spec(spec), params, images, torch.float64).numpy()
    x = torch.from_numpy(images).to(cuda)
    got64 = Engine(spec, params, 'f64', max_batch=2, device=cuda).forward(x).cpu().numpy()
    assert np.abs(got64 - ref).max() <= TOL_PARITY_MM, np.abs(got64 - ref).max()
    got32 = Engine(spec, params, 'f32', max_batch=2, device=cuda).forward(x).cpu().numpy()
    assert np.abs(got32 - ref).max() <= TOL_F32_STORAGE_MM, np.abs(got32 - ref).max()
    # the fp32-speed parity mode (fp32 matrix cores, fp32 storage: the arithmetic of the reference's fp32 graph) is as close to
    # exact math as the CPU fp32 restatement of the graph is: two correct fp32 implementations, two samples of the same rounding
    # noise (maximum of 51 values per crop within 2.5x, mean within 2x)
    got32m = Engine(spec, params, 'f32m', max_batch=2, device=cuda).forward(x).cpu().numpy()
    cpu32 = OF.forward(H.oracle_spec(spec), params, images, torch.float32).numpy().astype(np.float64)
    e32m, ecpu = np.abs(got32m - ref), np.abs(cpu32 - ref)
    print(f'\nf32m mode: max {e32m.max():.2e} mean {e32m.mean():.2e} mm; CPU fp32 restatement: max {ecpu.max():.2e} mean {ecpu.mean():.2e} mm')
    assert e32m.max() <= max(2.5 * ecpu.max(), 2e-3) and e32m.mean() <= max(2.0 * ecpu.mean(), 5e-4), (e32m.max(), ecpu.max(), e32m.mean(), ecpu.mean())
    got16 = Engine(spec, params, 'f16', max_batch=2, device=cuda).forward(x).cpu().numpy()
    assert np.isfinite(got16).all()
    # fp16 activations (rel 5e-4 per layer over 50+ layers) cost a few mm; the f16 mode must be as accurate as the
    # one-rounding-per-tensor fp16 model of the graph (per launch: tests/test_f16_layerwise.py)
    from oracle import f16emu
    emu = f16emu.forward(H.oracle_spec(spec), params, images).numpy()
    e16, eemu = np.abs(got16 - ref), np.abs(emu - ref)
    assert e16.max() <= 2.5 * eemu.max() and e16.mean() <= 2.0 * eemu.mean(), (e16.max(), eemu.max(), e16.mean(), eemu.mean())


def test_batch_independence_bit_exact(cuda):
    """No op crosses the batch dimension: a batch and its halves give identical bits."""
    spec = ModelSpec(50, 16, 'h36m', base_width=16)
    params, images = _setup(spec, 6)
    x = torch.from_numpy(images).to(cuda)
    for prec in ('f16', 'f32', 'f32m', 'f64'):
        eng = Engine(spec, params, prec, max_batch=8, device=cuda)
        whole = eng.forward(x).clone()
        parts = torch.cat([eng.forward(x[:2]).clone(), eng.forward(x[2:]).clone()])
        assert torch.equal(whole, parts)


def test_batch_independence_full_width_persistent_kernels(cuda):
    """Full base width (the persistent kernels walk tiles of several images per block, the stem kernel patches):
    how tiles are dealt to blocks changes with the batch, the per-pixel arithmetic must not (the soft-argmax slab
    partition is batch independent for the same reason).  70 crops = more tiles than resident blocks in every
    persistent kernel; sub-batches of 1, 5 and 64 give the same bits."""
    spec = ModelSpec(50, 16, 'h36m')
    params, images = _setup(spec, 70, gain=synth.logit_gain_for(50, 16))
    x = torch.from_numpy(images).to(cuda)
    eng = Engine(spec, params, 'f16', max_batch=70, device=cuda)
    whole = eng.forward(x).clone()
    assert torch.isfinite(whole).all()
    parts = torch.cat([eng.forward(x[:1]).clone(), eng.forward(x[1:6]).clone(), eng.forward(x[6:]).clone()])
    assert torch.equal(whole, parts)


def test_large_batch_tile_shapes_agree_with_small_batches(cuda):
    """From 128 crops per call on, the 3x3 layers with >= 256 tiles of 512 pixels switch to those tiles (32-channel chunks: a
    different fp32 accumulation order), so a large batch is no longer BIT-identical to its halves; it must still be the same
    tensor to within rounding flips at the first such layer (same inputs up to there), and the same poses to within the
    distance of two fp16 realisations of one graph."""
    from tests.test_f16_layerwise import compare_fp16
    spec = ModelSpec(50, 16, 'h36m')
    n = 130
    params, images = _setup(spec, n, gain=synth.logit_gain_for(50, 16))
    x = torch.from_numpy(images).to(cuda)
    big = Engine(spec, params, 'f16', max_batch=n, device=cuda)
    small = Engine(spec, params, 'f16', max_batch=n // 2, device=cuda)
    names = [li.name.decode() for li in big.layer_infos()]
    halves = lambda f: torch.cat([f(x[:n // 2]).clone(), f(x[n // 2:]).clone()])
    i = names.index('block2/unit_1/conv2')
    before = big.forward_upto(x, i - 1)
    assert torch.equal(before, halves(lambda t: small.forward_upto(t, i - 1))), 'layers in front of the first 512-pixel-tile layer'
    a = big.forward_upto(x, i).cpu().double().numpy()
    b = halves(lambda t: small.forward_upto(t, i)).cpu().double().numpy()
    assert not np.array_equal(a, b), 'expected the 512-pixel tiles at this batch (dispatch changed? update this test)'
    compare_fp16(a, b, 'block2/unit_1/conv2, 512-pixel vs 256-pixel tiles')
    # the poses of BOTH dispatches are held to exact math by the accuracy criterion of the f16 mode (crops 0, 64, 65, 129: both halves'
    # edges), not to one another by a millimetre bound; every launch of the >= 128-crop dispatch is held to a rounding flip at its
    # real batch by tests/test_f16_layerwise.py (the batch-130 case) and tests/test_kernel_coverage.py
    sel = [0, n // 2 - 1, n // 2, n - 1]
    pa, pb = big.forward(x)[sel].cpu().numpy(), halves(small.forward)[sel].cpu().numpy()
    H.assert_as_accurate_as_fp16_model(spec, params, images[sel], pa, '130 crops in one call')
    H.assert_as_accurate_as_fp16_model(spec, params, images[sel], pb, '65 + 65 crops')


def test_two_devices_in_one_process(cuda):
    """One process may drive several GPUs (inference._engine_for caches one Engine per device): the per-kernel LDS
    opt-in and the persistent kernels' grid caps are per DEVICE (metro_common.h: PerDeviceInt), so the second device
    runs the > 64 KiB-LDS kernels too.  Needs two visible GPUs."""
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two visible GPUs')
    spec = ModelSpec(50, 16, 'h36m')
    params, images = _setup(spec, 2, gain=synth.logit_gain_for(50, 16))
    outs = []
    for d in (0, 1):
        dev = torch.device('cuda', d)
        with torch.cuda.device(dev):
            outs.append(Engine(spec, params, 'f16', max_batch=2, device=dev).forward(torch.from_numpy(images).to(dev)).cpu())
    assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1])


def test_estimate_pose_boundary(cuda, tmp_path):
    """Same call shape as reference inference.py:31-43."""
    from metro_pose3d_amd import save_model
    from metro_pose3d_amd.inference import estimate_pose
    spec = ModelSpec(50, 32, 'h36m', base_width=8)
    params, images = _setup(spec, 2)
    path = str(tmp_path / 'toy.npz')
    save_model(path, spec, params)
    poses, edges, names = estimate_pose(images, path, precision='f64')
    assert poses.shape == (2, 17, 3) and poses.dtype == torch.float32 and poses.is_cuda
    assert edges.dtype == np.int64 and edges.shape == (16, 2)
    assert names[0] == b'pelv' and len(names) == 17
    ref = OF.forward(H.oracle_spec(spec), params, images, torch.float64).numpy()
    assert np.abs(poses.cpu().numpy() - ref).max() <= TOL_PARITY_MM
    with pytest.raises(ValueError):
        estimate_pose(images[:, :128], path)
    with pytest.raises(ValueError):
        estimate_pose(images.astype(np.float64), path)


def test_race_screen_repeated_runs_are_bit_identical(cuda):
    """The LDS-DMA rings are ordered by hand-counted s_waitcnt vmcnt(N) + one s_barrier per K step.
    A too-early read shows up as rare wrong tiles that come and go between launches, so: same
    input, many launches, interleaved with an unrelated memory-heavy kernel to perturb timing; every
    output must be bit-identical, for the full-size net (all tile configs incl. the slab kernel)."""
    spec = ModelSpec(50, 16, 'h36m')
    params, images = _setup(spec, 8, gain=synth.logit_gain_for(50, 16))
    x = torch.from_numpy(images).to(cuda)
    eng = Engine(spec, params, 'f16', max_batch=8, device=cuda)
    ref = eng.forward(x).clone()
    junk = torch.empty(64 << 20, dtype=torch.uint8, device=cuda)
    for i in range(40):
        if i % 3 == 0:
            junk.fill_(i)                       # uneven load on the memory system
        out = eng.forward(x)
        assert torch.equal(out, ref), f'launch {i} differs: max |d| {(out - ref).abs().max().item()}'
    # layer outputs too (a wrong tile can be averaged away by the soft-argmax)
    n_layers = len(eng.layer_infos())
    for li in sorted({0, 1, 5, 6, 20, 30, n_layers - 4, n_layers - 3, n_layers - 2}):
        a = eng.forward_upto(x, li)
        for _ in range(5):
            assert torch.equal(eng.forward_upto(x, li), a), f'layer {li} not deterministic'


def test_race_screen_batch64_persistent_kernels(cuda):
    """Same screen at the bench batch: every persistent block walks 4-16 tiles / 8 patches through its
    double-buffered LDS-DMA rings with counted waits; bits must not depend on timing."""
    spec = ModelSpec(50, 16, 'h36m')
    params, images = _setup(spec, 64, gain=synth.logit_gain_for(50, 16))
    x = torch.from_numpy(images).to(cuda)
    eng = Engine(spec, params, 'f16', max_batch=64, device=cuda)
    names = [li.name.decode() for li in eng.layer_infos()]
    picks = [i for i, n in enumerate(names) if n in ('conv1+pool1', 'block1/unit_1/conv1+conv2', 'block1/unit_1/conv3+unit_2/conv1',
                                                     'block2/unit_1/shortcut+conv1',
                                                     'block1/unit_3/conv3', 'block2/unit_2/conv3+unit_3/conv1', 'block3/unit_2/conv2',
                                                     'block3/unit_3/conv3', 'block4/unit_2/conv3', 'logits',
                                                     'block1/unit_2/conv2',          # conv3x3_c64: persistent, double-buffered slabs
                                                     'block4/unit_1/shortcut+conv1')]   # conv_gemm4w pair: register-staged K tiles
    assert len(picks) == 12
    ref = eng.forward(x).clone()
    refs = {i: eng.forward_upto(x, i).clone() for i in picks}
    seconds = {i: eng.forward_upto(x, i, second=True).clone() for i in picks if eng.layer_infos()[i].out2_offset >= 0}
    assert torch.isfinite(ref).all() and len(seconds) == 5
    junk = torch.empty(256 << 20, dtype=torch.uint8, device=cuda)
    for it in range(12):
        if it % 2 == 0:
            junk.fill_(it)
        assert torch.equal(eng.forward(x), ref), f'launch {it} differs'
        i = picks[it % len(picks)]
        assert torch.equal(eng.forward_upto(x, i), refs[i]), f'layer {names[i]} not deterministic'
        if i in seconds:
            assert torch.equal(eng.forward_upto(x, i, second=True), seconds[i]), f'layer {names[i]} (second output) not deterministic'


def test_hipgraph_replay_is_bit_identical(cuda, monkeypatch):
    """metro_plan_set_graph_max_batch: a captured forward replays to the same bits as plain launches."""
    spec = ModelSpec(50, 32, 'h36m', base_width=16)
    params, images = _setup(spec, 2)
    x = torch.from_numpy(images).to(cuda)
    monkeypatch.setenv('METRO_HIPGRAPH_MAX_BATCH', '0')
    ref = Engine(spec, params, 'f16', max_batch=2, device=cuda).forward(x).clone()
    monkeypatch.setenv('METRO_HIPGRAPH_MAX_BATCH', '4')
    eng = Engine(spec, params, 'f16', max_batch=2, device=cuda)
    out = torch.empty_like(ref)
    for _ in range(4):                        # eager, capture, replay, replay
        out.zero_()
        eng.forward(x, out=out)
        assert torch.equal(out, ref)


@pytest.mark.parametrize('arch,stride,dataset,n', [(101, 8, 'many19', 32), (50, 4, 'h36m', 16), (50, 16, 'many19', 64)],
                         ids=['C4-rn101-s8-J19-b32', 'C5-rn50-s4-J17-b16', 'C3-rn50-s16-J19-b64'])
def test_batch_independence_of_the_other_baseline_configs(cuda, arch, stride, dataset, n):
    """tests/test_f16_layerwise.py holds every launch of these configurations to the fp16 oracle at n = 1; at their
    per-GPU batch (BASELINE.json configs[2..4] sharded 8 ways) other kernel instantiations run (tests/test_kernel_coverage.py
    checks each of them on its own).  This transfers the n = 1 parity to the real batch END TO END: while every tile shape
    accumulates in the same order (stride 16: below 128 crops) the batch and its split 1 + (n - 1) must give the same BITS, layer
    outputs included; strides 4 and 8 change tile shape exactly at their shard size (below)."""
    spec = ModelSpec(arch, stride, dataset)
    params, images = _setup(spec, n, gain=synth.logit_gain_for(arch, stride))
    x = torch.from_numpy(images).to(cuda)
    eng = Engine(spec, params, 'f16', max_batch=n, device=cuda)
    kern_n, kern_1 = eng.layer_kernels(n), eng.layer_kernels(1)
    assert kern_n != kern_1, 'the dispatch at the real batch is expected to differ from the dispatch at n = 1'
    whole = eng.forward(x).clone()
    assert torch.isfinite(whole).all()
    parts = torch.cat([eng.forward(x[:1]).clone(), eng.forward(x[1:]).clone()])
    names = [li.name.decode() for li in eng.layer_infos()]
    i_head = names.index('logits')
    # round 6: the dilated 3x3 layers of strides 4 and 8 run on the tap-reuse kernel in sub-grid order and reach its 512-pixel
    # tiles (32-channel chunks: another fp32 summation order) exactly at these per-GPU batches -- 16 crops at stride 4, 32 at
    # stride 8 -- while n - 1 crops stay on 256-pixel tiles.  Up to the first such layer the bits must not depend on the batch;
    # AT it the two tile shapes see the same inputs and may differ by rounding flips only; behind it the two chains are two fp16
    # realisations of one graph, each held to exact math by the accuracy criterion of the f16 mode.
    kern_m = eng.layer_kernels(n - 1)
    chunk = lambda k: 'kc32' in k
    first = next((i for i in range(len(names)) if chunk(kern_n[i]) != chunk(kern_1[i]) or chunk(kern_n[i]) != chunk(kern_m[i])), None)
    if first is not None:
        from tests.test_f16_layerwise import compare_fp16
        split = lambda i: torch.cat([eng.forward_upto(x[:1], i), eng.forward_upto(x[1:], i)])
        assert torch.equal(eng.forward_upto(x, first - 1), split(first - 1)), f'layers in front of {names[first]} depend on the batch'
        a, b = eng.forward_upto(x, first).cpu().double().numpy(), split(first).cpu().double().numpy()
        compare_fp16(a, b, f'{names[first]}: {kern_n[first]} vs {kern_m[first]}')
        sel = [0, n - 1]
        H.assert_as_accurate_as_fp16_model(spec, params, images[sel], whole[sel].cpu().numpy(), f'{n} crops in one call')
        H.assert_as_accurate_as_fp16_model(spec, params, images[sel], parts[sel].cpu().numpy(), f'1 + {n - 1} crops')
        return
    if kern_n[i_head] != kern_1[i_head] or kern_n[i_head] != eng.layer_kernels(n - 1)[i_head]:
        # the head takes 256-pixel tiles once they give every CU a tile (RN50-s4 from 16 crops on): its fp32 logits are then
        # accumulated over K in one run instead of four K-quarters -- another fp32 summation order, so the POSES agree to fp32
        # rounding (soft-argmax of logits that differ by ~1e-6 relative), while every conv tensor in front still has the same bits
        a = eng.forward_upto(x, i_head - 1)
        b = torch.cat([eng.forward_upto(x[:1], i_head - 1), eng.forward_upto(x[1:], i_head - 1)])
        assert torch.equal(a, b), 'the residual stream in front of the head depends on the batch'
        assert (whole - parts).abs().max().item() <= 3e-3, (whole - parts).abs().max().item()    # ~10 fp32 ulps of a 1000 mm coordinate
        return
    if not torch.equal(whole, parts):            # name the first layer whose bits depend on the batch
        for i, li in enumerate(eng.layer_infos()):
            if li.kind == _lib.LAYER_SOFTARGMAX:
                continue
            a = eng.forward_upto(x, i)
            b = torch.cat([eng.forward_upto(x[:1], i), eng.forward_upto(x[1:], i)])
            assert torch.equal(a, b), f'{li.name.decode()}: {kern_n[i]} (n = {n}) vs {kern_1[i]} / {eng.layer_kernels(n - 1)[i]}'
    assert torch.equal(whole, parts)


def test_estimate_pose_plans_for_the_batch_it_is_given(cuda, tmp_path, monkeypatch):
    """The reference's placeholder is [None, 256, 256, 3] (main.py:109-111): a caller may pass any N.  256 crops go through ONE
    metro_forward(n = 256) -- the dispatch the north star's batch-256 figure is measured on (512-pixel 3x3 tiles, more
    layers on the 256 x 256 GEMM) -- not through four calls of 64; the result agrees with the chunked one to within the
    distance of two fp16 summation orders (test_large_batch_tile_shapes_agree_with_small_batches), and 300 crops are
    chunked 256 + 44."""
    from metro_pose3d_amd import inference as INF, save_model
    spec = ModelSpec(50, 16, 'h36m')
    params, images = _setup(spec, 300, gain=synth.logit_gain_for(50, 16))
    path = str(tmp_path / 'rn50s16.npz')
    save_model(path, spec, params)
    calls = []
    real = Engine.forward

    def counting(self, imgs, out=None):
        calls.append((self.max_batch, int(imgs.shape[0])))
        return real(self, imgs, out=out)

    monkeypatch.setattr(Engine, 'forward', counting)
    x = torch.from_numpy(images).to(cuda)
    p256, _, _ = INF.estimate_pose(x[:256], path)
    assert calls == [(256, 256)], calls
    calls.clear()
    p64 = torch.cat([INF.estimate_pose(x[i:i + 64], path)[0] for i in range(0, 256, 64)])
    assert calls == [(64, 64)] * 4, calls
    sel = [0, 63, 64, 255]
    assert torch.isfinite(p256).all() and torch.isfinite(p64).all()
    H.assert_as_accurate_as_fp16_model(spec, params, images[sel], p256[sel].cpu().numpy(), 'estimate_pose, 256 crops in one call')
    H.assert_as_accurate_as_fp16_model(spec, params, images[sel], p64[sel].cpu().numpy(), 'estimate_pose, 4 calls of 64 crops')
    calls.clear()
    p300, _, _ = INF.estimate_pose(x, path)
    assert calls == [(256, 256), (256, 44)], calls
    assert torch.equal(p300[:256], p256)
    calls.clear()
    INF.estimate_pose(x[:3], path)
    assert calls == [(8, 3)], calls
    assert len(INF._ENGINES) <= INF.MAX_CACHED_ENGINES


def _two_rank_worker(rank, world, port, spec_json, n, q):
    import os
    import torch.distributed as dist
    from metro_pose3d_amd.dist import shard_range, sharded_forward
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        spec = ModelSpec.from_json(spec_json)
        params, images = _setup(spec, n, gain=synth.logit_gain_for(spec.arch, spec.stride))
        dev = torch.device('cuda', 0)                      # BOTH ranks on the one GPU of the box
        b, e = shard_range(n, rank, world)
        eng = Engine(spec, params, 'f16', max_batch=max(e - b, 1), device=dev)
        # the real engine on this rank's shard; the pose all-gather runs on gloo (host tensors): RCCL needs one GPU per rank
        got = sharded_forward(lambda t: eng.forward(t.to(dev)).cpu(), torch.from_numpy(images))
        q.put((rank, got.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n', [16, 7], ids=['even', 'ragged'])
def test_two_ranks_real_engine_gather_equals_single_process(cuda, n):
    """Row (e) with the REAL engine: two processes (world 2), each running metro_forward on its contiguous shard of the batch,
    one all-gather of the poses -- bit-identical to the single-process run of the whole batch.  Both ranks share cuda:0 and
    the gather goes over gloo, which is what a 1-GPU box can show; the RCCL path is bench.py --gpus N."""
    import socket
    import torch.multiprocessing as mp
    spec = ModelSpec(50, 16, 'many19')
    params, images = _setup(spec, n, gain=synth.logit_gain_for(50, 16))
    want = Engine(spec, params, 'f16', max_batch=n, device=cuda).forward(torch.from_numpy(images).to(cuda)).cpu().numpy()
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, spec.to_json(), n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for r in range(2):
        assert res[r].shape == want.shape and np.array_equal(res[r], want), f'rank {r}: gathered poses differ from the single-process run'


def _estimate_pose_rank(rank, world, port, path, n, q, precision=None):
    import os
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    if precision:
        os.environ['METRO_PRECISION'] = precision
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from metro_pose3d_amd import inference as INF
        torch.cuda.set_device(0)                              # BOTH ranks on the one GPU of the box
        images = torch.from_numpy(synth.make_images(n))       # host tensor: only the rank's shard is uploaded
        poses, edges, names = INF.estimate_pose(images, path)
        q.put((rank, poses.cpu().numpy(), str(poses.device), [int(k[-1]) for k in INF._ENGINES]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n', [16, 7, 1], ids=['even', 'ragged', 'fewer-crops-than-ranks'])
def test_estimate_pose_shards_across_ranks(cuda, tmp_path, n):
    """Row (e) THROUGH THE BOUNDARY: two processes under an initialised process group call estimate_pose(images, model_path)
    with the same N crops; each forwards its contiguous shard with the real engine and every rank gets all N poses back,
    bit-identical to the single-process call (below 128 crops per call every tile shape sums in the same order).  Both ranks
    share cuda:0 and the all-gather runs over gloo -- what a 1-GPU box can show; under backend `nccl` the same code path gathers
    device tensors with RCCL (bench.py --gpus N)."""
    import socket
    import torch.multiprocessing as mp
    from metro_pose3d_amd import inference as INF, save_model
    spec = ModelSpec(50, 16, 'many19')
    params, images = _setup(spec, n, gain=synth.logit_gain_for(50, 16))
    path = str(tmp_path / 'rn50s16j19.npz')
    save_model(path, spec, params)
    want = INF.estimate_pose(torch.from_numpy(images).to(cuda), path)[0].cpu().numpy()
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_estimate_pose_rank, args=(r, 2, port, path, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r: (a, d, b) for r, a, d, b in (q.get(timeout=600) for _ in range(2))}
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for r in range(2):
        got, dev, buckets = res[r]
        assert dev == 'cuda:0' and got.shape == want.shape == (n, 19, 3)
        assert np.array_equal(got, want), f'rank {r}: sharded estimate_pose differs from the single-process call'
        assert buckets == [INF.batch_bucket(max(-(-n // 2) if r == 0 else n // 2, 1))]      # planned for the SHARD, not for N


def test_f64_mode_is_shard_invariant(cuda, tmp_path):
    """"results must be bit-identical to the 1-GPU run".  In the f16 throughput mode that holds below the dispatch
    thresholds only (tile shapes follow the crops per call); the PARITY mode runs one kernel configuration whatever the batch, so
    the same 300 crops give the same bits as one call (chunks of 256 + 44), as five calls of 60, and as two ranks of 150 under
    a process group (gloo on one GPU here; RCCL on a node)."""
    import socket
    import torch.multiprocessing as mp
    from metro_pose3d_amd import inference as INF, save_model
    n = 300
    spec = ModelSpec(50, 16, 'h36m')
    params, images = _setup(spec, n, gain=synth.logit_gain_for(50, 16))
    path = str(tmp_path / 'rn50s16.npz')
    save_model(path, spec, params)
    x = torch.from_numpy(images)
    one = INF.estimate_pose(x.to(cuda), path, precision='f64')[0].cpu().numpy()
    five = np.concatenate([INF.estimate_pose(x[i:i + 60].to(cuda), path, precision='f64')[0].cpu().numpy() for i in range(0, n, 60)])
    INF.clear_cache()
    torch.cuda.empty_cache()
    assert np.isfinite(one).all() and np.array_equal(one, five), 'f64 mode: 1 x (256 + 44) and 5 x 60 differ'
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_estimate_pose_rank, args=(r, 2, port, path, n, q, 'f64')) for r in range(2)]
    for p in procs:
        p.start()
    res = {r: a for r, a, _, _ in (q.get(timeout=900) for _ in range(2))}
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for r in range(2):
        assert np.array_equal(res[r], one), f'f64 mode: rank {r} of a 2-rank sharded call differs from the single-GPU call'


def test_fp16_overflow_is_reported_not_returned(cuda, tmp_path):
    """A checkpoint whose residual stream exceeds fp16's 65 504 (here: ResNet-101 with every conv3 at 3x its He initialisation;
    the reference keeps fp32 variables under fp16 compute for this reason, tfu.py:426-440) must not hand back NaN -- or worse,
    finite but wrong -- poses: the finalize launch flags the crops, metro_forward_status / estimate_pose raise and name the
    remedy, and the remedy (precision='f32m') works on the same file."""
    from metro_pose3d_amd import inference as INF, save_model
    spec = ModelSpec(101, 32, 'h36m')
    params = synth.make_params(101, spec.n_head_channels, 64, seed=0, logit_gain=1e-6, res_gain=3.0)
    images = torch.from_numpy(synth.make_images(3)).to(cuda)
    path = str(tmp_path / 'rn101_overflowing.npz')
    save_model(path, spec, params)
    eng = Engine(spec, params, 'f16', max_batch=4, device=cuda)
    poses = eng.forward(images)
    with pytest.raises(_lib.NonFiniteError, match='f32m'):
        eng.check_finite(3)
    with pytest.raises(_lib.NonFiniteError, match='f32m'):
        INF.estimate_pose(images, path)
    silent = INF.estimate_pose(images, path, check_finite=False)[0]                # the old behaviour, on request
    assert silent.shape == (3, 17, 3)
    ok = INF.estimate_pose(images, path, precision='f32m')[0]
    assert torch.isfinite(ok).all()
    # a healthy model passes the screen in every mode and the flag is rewritten by every forward
    good = synth.make_params(101, spec.n_head_channels, 64, seed=0, logit_gain=synth.logit_gain_for(101, 32))
    eng2 = Engine(spec, good, 'f16', max_batch=4, device=cuda)
    eng2.forward(images)
    eng2.check_finite(3)
    INF.clear_cache()
    assert len(INF._ENGINES) == 0
