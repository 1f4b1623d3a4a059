"""metro_stem_pool_f32in back to back at the given batch sizes (METRO_HIP_LIB=... for knock-out builds: results garbage, times not)."""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, '.')
from metro_pose3d_amd import _lib
from tests import helpers as H
lib = _lib.load(); dev = torch.device('cuda', 0)
res = []
for n in [int(v) for v in sys.argv[1:]] or [64]:
    x = torch.rand((n, 256, 256, 3), dtype=torch.float32, device=dev)
    w = (torch.randn((64, 7, 8, 4), device=dev) * 0.05).half()
    b = torch.zeros(64, dtype=torch.float32, device=dev)
    out = torch.empty((n, 64, 64, 64), dtype=torch.float16, device=dev)
    fn = lambda: lib.metro_stem_pool_f32in(H.ptr(x), H.ptr(w), H.ptr(b), H.ptr(out), n, 256, C.c_void_p(0))
    for _ in range(5): assert fn() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    res.append(f'n={n}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us')
print('  '.join(res))
