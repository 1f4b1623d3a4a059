"""block3's 1x1 layers at batch 64 (one 128 x 128 tile per CU) through metro_conv_f16, operands cold (a 512 MB buffer is touched between
launches) and warm (back to back): what a knock-out build of conv_igemm_f16_dma.hip changes (tools/build_dbg_variants.sh)."""
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, '.')
from metro_pose3d_amd import _lib
from tests import helpers as H
lib = _lib.load(); xlib = _lib.load_experimental(); dev = torch.device('cuda', 0)
def run(name, n, h, c_in, c_out, pro, res, relu=False, reps=30):
    g = torch.Generator(device='cpu').manual_seed(0)
    x = torch.randn((n, h, h, c_in), generator=g).half().to(dev)
    w = (torch.randn((c_out, 1, 1, c_in), generator=g) * 0.05).half().to(dev)
    b = torch.zeros(c_out, dtype=torch.float32, device=dev)
    sc = torch.ones(c_in, dtype=torch.float16, device=dev) if pro else None
    sh = torch.zeros(c_in, dtype=torch.float16, device=dev) if pro else None
    r = torch.randn((n, h, h, c_out), generator=g).half().to(dev) if res else None
    out = torch.empty((n, h, h, c_out), dtype=torch.float16, device=dev)
    d = H.conv_desc(n, h, c_in, h, c_out, 1, prologue=pro, relu=relu, residual=res, res_h=h, in_dtype=_lib.METRO_F16)
    f = lambda: lib.metro_conv_f16(C.byref(d), H.ptr(x), H.ptr(w), H.ptr(b), H.ptr(sc), H.ptr(sh), H.ptr(r), H.ptr(out), None)
    lib.metro_kernel_notes(1); f(); kid = lib.metro_last_kernel_id().decode(); lib.metro_kernel_notes(0)
    f(); torch.cuda.synchronize(); ref = out.clone()
    junk = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    gf = 2.0 * n * h * h * c_in * c_out / 1e9
    def timeit(fn):
        res_us = []
        for cold in (True, False):
            e0 = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]; e1 = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
            for i in range(reps):
                if cold: junk.add_(1)
                e0[i].record(); fn(); e1[i].record()
            torch.cuda.synchronize()
            res_us.append(float(np.median([a.elapsed_time(b_) for a, b_ in zip(e0, e1)][5:])) * 1e3)
        return res_us
    res_us = timeit(f)
    print('%-30s cold %6.1f us  warm %6.1f us  (%4.0f / %4.0f TF)  %s' % (name, res_us[0], res_us[1], gf / res_us[0] * 1e3, gf / res_us[1] * 1e3, kid))
    if not hasattr(xlib, 'metro_conv_f16_gemm4d_geo'): return
    for geo in (0, 1, 2):
        out.fill_(float('nan'))
        g = lambda: xlib.metro_conv_f16_gemm4d_geo(C.byref(d), H.ptr(x), H.ptr(w), H.ptr(b), H.ptr(sc), H.ptr(sh), H.ptr(r), H.ptr(out), 0, None, geo, None)
        if g() != 0: print('%-30s gemm4d geo %d: %s' % ('', geo, lib.metro_last_error().decode()[:60])); continue
        torch.cuda.synchronize(); same = torch.equal(out, ref)
        res_us = timeit(g)
        print('%-30s cold %6.1f us  warm %6.1f us  (%4.0f / %4.0f TF)  gemm4d geo %d  bits %s' % ('', res_us[0], res_us[1], gf / res_us[0] * 1e3, gf / res_us[1] * 1e3, geo, 'same' if same else 'DIFFER'))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
run('b3 conv1 1024->256 pro relu', n, 16, 1024, 256, True, False, True)
run('b4 conv1 2048->512 pro relu', n, 16, 2048, 512, True, False, True)
run('b4 u1 conv1 1024->512 pro relu', n, 16, 1024, 512, True, False, True)
run('b3 conv3 256->1024 +res', n, 16, 256, 1024, False, True)
run('b4 conv3 512->2048 +res', n, 16, 512, 2048, False, True)
run('b4 shortcut 1024->2048 pro', n, 16, 1024, 2048, True, False)
run('b2 conv3+.. 128->512 +res', n, 32, 128, 512, False, True)
