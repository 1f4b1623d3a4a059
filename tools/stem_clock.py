"""Reads the per-interval cycle sums of a METRO_DBG_SP2_CLOCK build of the rows stem (tools/build_dbg_variants.sh stem_pool_f16.hip SP2_CLOCK)."""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, '.')
from metro_pose3d_amd import _lib
from tests import helpers as H
lib = _lib.load(); dev = torch.device('cuda', 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
x = torch.rand((n, 256, 256, 3), dtype=torch.float32, device=dev)
w = (torch.randn((64, 7, 8, 4), device=dev) * 0.05).half()
b = torch.zeros(64, dtype=torch.float32, device=dev)
out = torch.zeros((n, 64, 64, 64), dtype=torch.float16, device=dev)
for _ in range(3):
    assert lib.metro_stem_pool_f32in(H.ptr(x), H.ptr(w), H.ptr(b), H.ptr(out), n, 256, C.c_void_p(0)) == 0
torch.cuda.synchronize()
d = out.view(-1)[:32].cpu().numpy().view(np.int64)
rows = int(d[7])
names = ['', 'wait+barrier', 'tile+staging reads, requests', 'conv(2Y), conv(2Y+1)+hmax(2Y)', 'hmax(2Y+1), stores, window writes', 'vmax+tile writes']
print(f'n={n}: block 1, wave 0: {rows} pooled rows, {d[6]} cycles in the kernel; per iteration: ' +
      ', '.join(f'{names[i]} {d[i] / rows:.0f}' for i in range(1, 6)) + f'; sum {sum(d[1:6]) / rows:.0f}')

nb = n * 8
blk = out.view(-1)[32:32 + nb * 16].cpu().numpy().view(np.int64).reshape(nb, 4)
t0 = blk[:, 0].min()
start, end = (blk[:, 0] - t0) / 100.0, (blk[:, 1] - t0) / 100.0        # us
cu = (blk[:, 3] & 15) * 1000 + ((blk[:, 2] >> 13) & 7) * 100 + ((blk[:, 2] >> 12) & 1) * 50 + ((blk[:, 2] >> 8) & 15)     # xcc, se, sh, cu
print(f'blocks {nb}: start min/median/max {start.min():.1f}/{np.median(start):.1f}/{start.max():.1f} us, end median/max {np.median(end):.1f}/{end.max():.1f} us, '
      f'duration median {np.median(end - start):.1f} us; distinct CUs {len(set(cu.tolist()))}, blocks per CU max {np.bincount(np.unique(cu, return_inverse=True)[1]).max()}')
late = start > 5
print(f'blocks starting later than 5 us: {late.sum()}')
pairs = {}
for b, c in enumerate(cu.tolist()):
    pairs.setdefault(c, []).append(b)
ex = [v for v in pairs.values() if len(v) == 2][:12]
print('co-resident block ids (first CUs):', ex)
d = np.array([abs(v[1] - v[0]) for v in pairs.values() if len(v) == 2])
if len(d): print('id distance of co-resident pairs: ', np.unique(d, return_counts=True))
wid = blk[:, 2] & 15
simd = (blk[:, 2] >> 4) & 3
print('wave slot ids of wave 0 in blocks 0..7:', wid[:8].tolist(), ' in blocks 256..263:', wid[256:264].tolist() if nb > 263 else '', ' simd:', simd[:4].tolist(), simd[256:260].tolist() if nb > 263 else '')
print('slot id histogram:', np.bincount(wid).tolist())
