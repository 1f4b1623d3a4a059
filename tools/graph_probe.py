import sys, time, os, torch
sys.path.insert(0, '.')
from metro_pose3d_amd import ModelSpec, synth
from metro_pose3d_amd.engine import Engine
dev = torch.device('cuda', 0)
spec = ModelSpec(50, 16, 'h36m')
params = synth.make_params(50, spec.n_head_channels, 64, seed=0, logit_gain=1.04)
for b in (1, 2, 8):
    x = torch.from_numpy(synth.make_images(b)).to(dev)
    res = {}
    for g in (0, 8):
        os.environ['METRO_HIPGRAPH_MAX_BATCH'] = str(g)
        e = Engine(spec, params, 'f16', b, dev)
        out = torch.empty((b, 17, 3), device=dev)
        for _ in range(20): e.forward(x, out=out)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(300): e.forward(x, out=out)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 300 * 1e3
        res[g] = (dt, out.clone())
    print('batch %d: eager %.3f ms   graph %.3f ms   identical %s' % (b, res[0][0], res[8][0], torch.equal(res[0][1], res[8][1])))
