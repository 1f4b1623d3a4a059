#!/usr/bin/env python3
"""Generates tests/golden/*.npz with the fp64 CPU oracle, in the build container.

    python tests/golden/make_golden.py --calibrate   # prints the LOGIT_GAIN table for synth.py
    python tests/golden/make_golden.py               # writes golden_poses_v1.npz
    python tests/golden/make_golden.py --f16         # writes golden_f16emu_v1.npz (fp16-faithful oracle, oracle/f16emu.py)

The reference itself (TensorFlow 1.13 + a frozen .pb) cannot run here, so these vectors pin the
ORACLE (and through it the HIP path), not TensorFlow: "parity unpinned", see oracle/__init__.py.
Weights are not stored: they are regenerated from seeds by metro_pose3d_amd/synth.py, and a CRC
of every generated tensor is stored so a drifting generator is detected instead of silently
producing different expectations.  Inputs are stored by seed + CRC as well.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from metro_pose3d_amd import ModelSpec, synth  # noqa: E402
from oracle import f16emu  # noqa: E402
from oracle import forward as OF  # noqa: E402
from oracle.spec import OracleSpec, head_joint_info  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden_poses_v1.npz')
OUT_F16 = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden_f16emu_v1.npz')

# (case name, ModelSpec, batch)
CASES = [
    ('rn50-s32-h36m', ModelSpec(50, 32, 'h36m'), 2),          # BASELINE configs[0] shape
    ('rn50-s16-h36m', ModelSpec(50, 16, 'h36m'), 2),          # configs[1]
    ('rn50-s16-many19', ModelSpec(50, 16, 'many19'), 2),      # configs[2]
    ('rn101-s8-many19', ModelSpec(101, 8, 'many19'), 1),      # configs[3]
    ('rn50-s4-h36m', ModelSpec(50, 4, 'h36m'), 1),            # configs[4]
    ('rn50-s16-merged53', ModelSpec(50, 16, 'merged'), 1),    # J_head = 53 != J_out = 19
    ('toy-rn50-s32-w8', ModelSpec(50, 32, 'h36m', base_width=8), 3),
    ('toy-rn50-s16-w8', ModelSpec(50, 16, 'many19', base_width=8), 3),
    ('toy-rn50-s8-w8', ModelSpec(50, 8, 'h36m', base_width=8), 3),
    ('toy-rn50-s4-w8', ModelSpec(50, 4, 'h36m', base_width=8), 2),
    ('toy-rn101-s8-w8', ModelSpec(101, 8, 'merged', base_width=8), 2),
    ('toy-rn101-s4-w8', ModelSpec(101, 4, 'many19', base_width=8), 2),
    ('toy-rn50-s16-w16-noncentered', ModelSpec(50, 16, 'h36m', base_width=16, centered_stride=False), 2),
]


def ospec(spec: ModelSpec) -> OracleSpec:
    return OracleSpec(arch=spec.arch, stride=spec.stride, dataset=spec.dataset, depth=spec.depth,
                      centered_stride=spec.centered_stride, proc_side=spec.proc_side,
                      box_size_mm=spec.box_size_mm, base_width=spec.base_width)


def gain_for(spec: ModelSpec) -> float:
    return synth.logit_gain_for(spec.arch, spec.stride, spec.base_width)


def params_crc(params) -> int:
    crc = 0
    for k in sorted(params):
        crc = zlib.crc32(k.encode(), crc)
        crc = zlib.crc32(np.ascontiguousarray(params[k]).tobytes(), crc)
    return crc


def calibrate():
    """logits-kernel gain for per-joint logit std ~= 4 (fp32 oracle pass on 2 seeded crops)."""
    table = {}
    for arch in (50, 101):
        for stride in (32, 16, 8, 4):
            for bw in (64, 16, 8):
                if bw != 64 and arch == 101 and stride in (32, 16):
                    continue
                spec = ModelSpec(arch, stride, 'h36m', base_width=bw)
                params = synth.make_params(arch, spec.n_head_channels, bw, seed=0, logit_gain=1.0)
                images = synth.make_images(2, spec.proc_side)
                col = {}
                with torch.no_grad():
                    OF.forward(ospec(spec), params, images, torch.float32, col)
                lg = col['logits']                       # [N, D*J, S, S]
                j = head_joint_info('h36m').n_joints
                per_joint = lg.reshape(lg.shape[0], spec.depth, j, -1).permute(0, 2, 1, 3).reshape(lg.shape[0] * j, -1)
                std = float(per_joint.std(dim=1).mean())
                table[(arch, stride, bw)] = float(f'{4.0 / std:.3g}')
                print(f'    ({arch}, {stride}, {bw}): {table[(arch, stride, bw)]},   # raw per-joint std {std:.3f}', flush=True)
    return table


def make_f16():
    """Poses of the fp16 arithmetic model (one rounding per stored tensor) for the same seeded cases, plus probes of
    its fp16 tensors: pins oracle/f16emu.py, which the GPU tests hold the benchmarked f16 mode to."""
    out = {}
    for name, spec, n in CASES:
        params = synth.make_params(spec.arch, spec.n_head_channels, spec.base_width, seed=0, logit_gain=gain_for(spec))
        images = synth.make_images(n, spec.proc_side)
        col = {}
        with torch.no_grad():
            poses = f16emu.forward(ospec(spec), params, images, col).numpy()
        out[name + '/poses_f16emu'] = poses
        for key in ('pool1', 'block1/unit_1', 'block2/unit_4', 'block4/unit_3', 'logits'):
            flat = col[key].reshape(-1)
            idx = np.linspace(0, flat.numel() - 1, 16).astype(np.int64)
            out[f'{name}/probe16/{key}'] = flat[idx].numpy()
        print(f'{name}: f16emu poses {poses.shape}', flush=True)
    np.savez_compressed(OUT_F16, **out)
    print(f'wrote {OUT_F16} ({os.path.getsize(OUT_F16) / 1024:.1f} KiB)')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--calibrate', action='store_true')
    ap.add_argument('--f16', action='store_true')
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 1)
    if args.calibrate:
        calibrate()
        return
    if args.f16:
        make_f16()
        return
    out = {}
    meta = {}
    for name, spec, n in CASES:
        params = synth.make_params(spec.arch, spec.n_head_channels, spec.base_width, seed=0,
                                   logit_gain=gain_for(spec))
        images = synth.make_images(n, spec.proc_side)
        col = {}
        with torch.no_grad():
            poses = OF.forward(ospec(spec), params, images, torch.float64, col).numpy()
        out[name + '/poses'] = poses
        out[name + '/coords01'] = col['coords01'].numpy()
        lg = col['logits']
        # a few probes of intermediate tensors (NCHW in the oracle): value at fixed positions
        probes = {}
        for key in ('conv1', 'pool1', 'block1/unit_1', 'block2/unit_4', 'block4/unit_3', 'logits'):
            t = col[key]
            flat = t.reshape(-1)
            idx = np.linspace(0, flat.numel() - 1, 16).astype(np.int64)
            probes[key] = flat[idx].numpy()
            out[f'{name}/probe/{key}'] = np.concatenate([probes[key], [float(t.mean()), float(t.abs().mean())]])
        meta[name] = {'spec': json.loads(spec.to_json()), 'batch': n, 'param_seed': 0, 'image_seed': 1234,
                      'logit_gain': gain_for(spec), 'params_crc32': params_crc(params),
                      'images_crc32': zlib.crc32(images.tobytes()),
                      'logit_std': float(lg.std())}
        print(f'{name}: poses {poses.shape}, |pose|max {np.abs(poses).max():.1f} mm, logit std {float(lg.std()):.2f}', flush=True)
    # stand-alone soft-argmax vectors (K6): seeded logits -> poses
    for name, spec in (('sa-rn50-s16-h36m', ModelSpec(50, 16, 'h36m')), ('sa-rn101-s8-merged', ModelSpec(101, 8, 'merged'))):
        rng = np.random.default_rng(77)
        lg = (rng.standard_normal((2, spec.heatmap_side, spec.heatmap_side, spec.n_head_channels)) * 4).astype(np.float32)
        out[name + '/poses'] = OF.logits_to_output(ospec(spec), lg).numpy()
        meta[name] = {'spec': json.loads(spec.to_json()), 'logits_seed': 77, 'logits_crc32': zlib.crc32(lg.tobytes())}
    out['__meta__'] = np.frombuffer(json.dumps(meta, sort_keys=True).encode(), dtype=np.uint8)
    np.savez_compressed(OUT, **out)
    print(f'wrote {OUT} ({os.path.getsize(OUT) / 1024:.1f} KiB)')


if __name__ == '__main__':
    main()
