"""Write-only / copy / read-only device-memory rates (torch elementwise kernels) at activation-tensor sizes."""
import torch, numpy as np
dev = torch.device('cuda', 0)
def t(f, reps=20):
    e0 = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]; e1 = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
    for i in range(reps):
        e0[i].record(); f(); e1[i].record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in zip(e0, e1)][3:])) * 1e3
for mb in (34, 134, 537, 2147):
    n = mb * 1000 * 1000 // 2
    x = torch.empty(n, dtype=torch.float16, device=dev).normal_(); y = torch.empty_like(x)
    rows = []
    us = t(lambda: x.fill_(1.0)); rows.append(('fill (W)', mb / us))
    us = t(lambda: y.copy_(x)); rows.append(('copy (R+W)', 2 * mb / us))
    us = t(lambda: torch.relu_(x)); rows.append(('relu_ in place (R+W same)', 2 * mb / us))
    us = t(lambda: x.sum()); rows.append(('sum (R)', mb / us))
    print('%5d MB: ' % mb + '   '.join('%s %.2f TB/s' % r for r in rows))
