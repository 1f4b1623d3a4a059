"""Bone-length head on the GPU against the oracle (scipy): agreement of the z offset and kernel time."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from metro_pose3d_amd import ModelSpec, heads as MH
from oracle import heads as OH
from tests.test_heads import _problem
cuda = torch.device('cuda', 0)
spec = ModelSpec(50, 16, 'h36m')
rng = np.random.default_rng(0)
n = 2048
ji, p, c01, inv_k, bones = _problem(rng, spec, n)
target = bones.mean(axis=0)
t0 = time.perf_counter(); ref, zref = OH.backproject_bone_lengths(c01, inv_k, target, ji.edges, spec.stride); t_cpu = time.perf_counter() - t0
c = torch.from_numpy(c01).to(cuda)
got, z = MH.backproject_bone_lengths(c, inv_k, target, spec)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
k = torch.from_numpy(inv_k).to(cuda)
e0.record()
for _ in range(20): MH.backproject_bone_lengths(c, k, target, spec)
e1.record(); torch.cuda.synchronize()
dz = np.abs(z.cpu().numpy().astype(np.float64) - zref.astype(np.float64))
print('poses', n, 'max |dz| mm', dz.max(), 'exactly equal fp32', float((dz == 0).mean()), 'max |dcoord| mm', np.abs(got.cpu().numpy() - ref).max())
print('oracle (scipy, 1 thread): %.1f poses/s; GPU call incl. host glue: %.0f us per %d poses' % (n / t_cpu, e0.elapsed_time(e1) / 20 * 1e3, n))
