import sys; sys.path.insert(0, '.')
import bench, torch
import torch.distributed as dist
dev = torch.device('cuda', 0)
for a in ((4, 'h36m', 128, 20, 3, 'c5'), (16, 'h36m', 64, 50, 5, 'c2'), (16, 'h36m', 2048, 20, 3, 'c2b2048')):
    r = bench.softargmax_hbm_leg(dev, dist, *a)
    print(a[-1], r['us_per_call_median'], r['roofline']['achieved'], r['roofline']['frac'])
