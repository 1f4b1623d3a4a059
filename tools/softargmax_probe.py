#!/usr/bin/env python3
"""The stand-alone soft-argmax legs of bench.py on their own (for `rocprofv3 --kernel-trace --stats -- python tools/softargmax_probe.py`):
configs[4] volume x 128 crops, configs[1] volume x 64 and x 2048 crops."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device('cuda', 0)
for stride, crops, steps, what in ((4, 128, 20, 'configs[4] volume, 128 crops'), (16, 64, 50, 'configs[1] volume, 64 crops'), (16, 2048, 20, 'configs[1] volume, 2048 crops')):
    r = bench.softargmax_hbm_leg(dev, None, stride, 'h36m', crops, steps, 3, what)
    print(json.dumps({'workload': r['workload'], 'us': r['us_per_call_median'], 'roofline': {k: r['roofline'][k] for k in ('achieved', 'frac', 'algorithmic_bytes_per_call')}}))
