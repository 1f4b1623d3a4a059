#!/usr/bin/env python3
"""Poses of one forward on seeded synthetic data -> .npy (bit-comparison of two library builds / knob settings):
    METRO_HIP_LIB=... [KNOB=V] python tools/dump_poses.py --batch 64 --out a.npy [--arch 50 --stride 16 --dataset h36m]
    python tools/dump_poses.py --compare a.npy b.npy"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--arch', type=int, default=50)
    ap.add_argument('--stride', type=int, default=16)
    ap.add_argument('--dataset', default='h36m')
    ap.add_argument('--out')
    ap.add_argument('--compare', nargs=2)
    a = ap.parse_args()
    if a.compare:
        x, y = np.load(a.compare[0]), np.load(a.compare[1])
        same = x.shape == y.shape and np.array_equal(x.view(np.uint32), y.view(np.uint32))
        print(f'{a.compare[0]} vs {a.compare[1]}: {"BIT-IDENTICAL" if same else "DIFFERENT"}'
              + ('' if same else f' (max |d| {np.abs(x - y).max():.4g} mm, {np.mean(x != y) * 100:.1f} % of the values)'))
        sys.exit(0 if same else 1)
    import torch
    from metro_pose3d_amd import ModelSpec, synth
    from metro_pose3d_amd.engine import Engine
    spec = ModelSpec(a.arch, a.stride, a.dataset)
    params = synth.make_params(spec.arch, spec.n_head_channels, spec.base_width, seed=0, logit_gain=synth.logit_gain_for(spec.arch, spec.stride))
    dev = torch.device('cuda', 0)
    eng = Engine(spec, params, 'f16', max_batch=a.batch, device=dev)
    img = torch.from_numpy(synth.make_images(a.batch, spec.proc_side, seed=1234)).to(dev)
    out = eng.forward(img)
    torch.cuda.synchronize()
    np.save(a.out, out.cpu().numpy())
    print(f'{a.out}: {tuple(out.shape)} finite={bool(torch.isfinite(out).all())}')


if __name__ == '__main__':
    main()
