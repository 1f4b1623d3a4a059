"""metro_stem_pool_f32in under two knob settings of the A/B library (tools/build_knobs.sh): same bits?  and the time of each.
    METRO_HIP_LIB=$PWD/metro_pose3d_amd/dbg/libmetro_knobs.so python tools/stem_ab.py [n]"""
import ctypes as C, os, subprocess, sys
import numpy as np
if len(sys.argv) > 2:            # child: run one setting, dump the output
    import torch
    sys.path.insert(0, '.')
    from metro_pose3d_amd import _lib
    from tests import helpers as H
    n = int(sys.argv[1]); lib = _lib.load(); dev = torch.device('cuda', 0)
    g = torch.Generator(device='cpu').manual_seed(0)
    x = torch.rand((n, 256, 256, 3), generator=g).to(dev)
    w = (torch.randn((64, 7, 8, 4), generator=g) * 0.05).half(); w[:, :, 7, :] = 0; w[:, :, :, 3] = 0; w = w.to(dev)
    b = torch.randn(64, generator=g).to(dev)
    out = torch.full((n, 64, 64, 64), float('nan'), dtype=torch.float16, device=dev)
    fn = lambda: lib.metro_stem_pool_f32in(H.ptr(x), H.ptr(w), H.ptr(b), H.ptr(out), n, 256, C.c_void_p(0))
    lib.metro_kernel_notes(1); assert fn() == 0, lib.metro_last_error(); kid = lib.metro_last_kernel_id().decode(); lib.metro_kernel_notes(0)
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    o = out.cpu().numpy()
    np.save(sys.argv[2], o.view(np.uint16))
    print(f'{kid}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us  finite {np.isfinite(o.astype(np.float32)).all()}')
    sys.exit(0)
n = sys.argv[1] if len(sys.argv) > 1 else '64'
for rows in ('0', os.environ.get('STEM_AB_ROWS', '1')):
    env = dict(os.environ, METRO_STEM_ROWS=rows)
    print(subprocess.run([sys.executable, __file__, n, f'/tmp/stem_rows{rows}.npy'], env=env, capture_output=True, text=True).stdout.strip().split('\n')[-1])
a, b = np.load('/tmp/stem_rows0.npy'), np.load('/tmp/stem_rows%s.npy' % os.environ.get('STEM_AB_ROWS', '1'))
diff = a != b
print('same bits' if not diff.any() else f'DIFFER: {diff.sum()} of {diff.size}; first at {np.argwhere(diff)[:8].tolist()}')
