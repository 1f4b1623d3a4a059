import torch, time
x = torch.zeros(64, device='cuda')
for n in (1000,):
    for _ in range(100): x.add_(1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): x.add_(1)
    e1.record(); torch.cuda.synchronize()
    print('tiny dependent kernels: %.2f us each' % (e0.elapsed_time(e1) / n * 1e3))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): x.add_(1)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(1000): x.add_(1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
g.replay(); torch.cuda.synchronize()
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
print('same in a graph: %.2f us each' % (e0.elapsed_time(e1) / 1000 * 1e3))
