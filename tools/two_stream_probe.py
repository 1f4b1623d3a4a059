import sys, time, torch
sys.path.insert(0, '.')
from metro_pose3d_amd import ModelSpec, synth
from metro_pose3d_amd.engine import Engine
dev = torch.device('cuda', 0)
spec = ModelSpec(50, 16, 'h36m')
params = synth.make_params(50, spec.n_head_channels, 64, seed=0, logit_gain=1.04)
x = torch.from_numpy(synth.make_images(64)).to(dev)
def bench(fn, n=30, w=10):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
e64 = Engine(spec, params, 'f16', 64, dev)
print('1 stream  x64      %.3f ms' % bench(lambda: e64.forward(x)))
for parts in (2, 4):
    engs = [Engine(spec, params, 'f16', 64 // parts, dev) for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    xs = list(x.chunk(parts))
    def run():
        for e, s, xi in zip(engs, streams, xs):
            with torch.cuda.stream(s):
                e.forward(xi)
    print('%d streams x%d       %.3f ms' % (parts, 64 // parts, bench(run)))
    def run_seq():
        for e, xi in zip(engs, xs): e.forward(xi)
    print('1 stream  %dx%d (seq) %.3f ms' % (parts, 64 // parts, bench(run_seq)))
