#!/usr/bin/env python3
"""Two half-batches on two streams, each persistent kernel capped at half the CUs (METRO_CU_LIMIT in the knobs build),
the second stream started half a period late: does a memory-bound phase of one half overlap a compute-bound phase of
the other?   METRO_HIP_LIB=.../libmetro_knobs.so METRO_CU_LIMIT=128 python tools/partition_probe.py"""
import os, sys, time, torch
sys.path.insert(0, '.')
from metro_pose3d_amd import ModelSpec, synth
from metro_pose3d_amd.engine import Engine
dev = torch.device('cuda', 0)
spec = ModelSpec(50, 16, 'h36m')
params = synth.make_params(50, spec.n_head_channels, 64, seed=0, logit_gain=1.04)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = torch.from_numpy(synth.make_images(B)).to(dev)
parts = 2
engs = [Engine(spec, params, 'f16', B // parts, dev) for _ in range(parts)]
streams = [torch.cuda.Stream() for _ in range(parts)]
xs = list(x.chunk(parts))
outs = [torch.empty((B // parts, 17, 3), device=dev) for _ in range(parts)]
nl = len(engs[0].layer_infos())
def run(steps, offset):
    torch.cuda.synchronize()
    t = time.perf_counter()
    if offset:   # stream 1 starts after stream 0 has run its first `offset` layers
        with torch.cuda.stream(streams[0]):
            engs[0].forward_upto(xs[0], offset)
        streams[1].wait_stream(streams[0])
    for _ in range(steps):
        for e, s, xi, o in zip(engs, streams, xs, outs):
            with torch.cuda.stream(s):
                e.forward(xi, out=o)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / steps * 1e3
print('CU limit', os.environ.get('METRO_CU_LIMIT'), 'batch', B)
for off in (0, 20, 21, 30):
    run(5, off)
    print(f'  2 streams x{B // parts}, offset {off:2d} layers: {run(40, off):.3f} ms per full batch', flush=True)
