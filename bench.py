#!/usr/bin/env python3
"""Throughput of the MeTRo inference hot path on MI355X: crops/s on synthetic 256x256 batches.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
torch.distributed.run with one rank per GPU (RCCL).  A "step" is one pass of the hot path over
one resident batch per GPU -- fp32 NHWC crops in HBM -> poses [B, Jout, 3] in HBM -- plus, for
N > 1, the one all-gather of the pose outputs.  Weak scaling: the per-GPU batch is fixed.

Workload (BASELINE.json configs[1], the config the metric is quoted on): ResNet-50, stride 16,
17 H36M joints, batch 64 per GPU, fp16 compute (the reference's default dtype, options.py:73),
seeded synthetic weights and crops (metro_pose3d_amd/synth.py).

One JSON line on stdout from rank 0, with
  roofline     -- the implicit-GEMM conv kernel (all conv launches of the forward): algorithmic
                  FLOPs (2*MACs) / summed launch durations from HIP events
                  recorded on the launch stream, against the 2.5 PFLOP/s dense fp16 MFMA peak;
  cpu_baseline -- this repo's CPU restatement of the same graph (oracle/, PyTorch CPU fp32, all
                  host cores) timed on a bounded sample in the same run.  It is NOT TensorFlow:
                  the reference's own CPU path cannot run here (no TF 1.13, no frozen .pb).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F16_DENSE_TFLOPS = 2500.0    # /opt/skills/guides/MI355X_MICROARCH.md, dense (no sparsity)
PEAK_HBM_GBS = 8000.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=64, help='crops per GPU per step')
    ap.add_argument('--arch', type=int, default=50)
    ap.add_argument('--stride', type=int, default=16)
    ap.add_argument('--dataset', type=str, default='h36m')
    ap.add_argument('--precision', type=str, default='f16', choices=['f16', 'f32', 'f64'])
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='budget of the CPU baseline leg (0 = skip)')
    ap.add_argument('--cpu-crops', type=int, default=8)
    ap.add_argument('--layer-report', type=str, default=None, help='write the per-layer table to this file')
    return ap.parse_args()


def cpu_baseline(spec, params, seconds: float, crops: int):
    """Oracle (CPU restatement) timed on host cores.  kind = 'port'."""
    from oracle import forward as OF
    from oracle.spec import OracleSpec
    from metro_pose3d_amd import synth
    ospec = OracleSpec(arch=spec.arch, stride=spec.stride, dataset=spec.dataset, depth=spec.depth,
                       centered_stride=spec.centered_stride, proc_side=spec.proc_side,
                       box_size_mm=spec.box_size_mm, base_width=spec.base_width)
    avail = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    images = synth.make_images(crops, spec.proc_side, seed=99)
    t_start = time.perf_counter()
    with torch.no_grad():
        # pick the thread count the host actually runs this graph fastest with (1 crop each)
        best = (float('inf'), 1)
        for threads in sorted({t for t in (8, 16, 32, 64, min(avail, 96)) if t <= avail}):
            torch.set_num_threads(threads)
            OF.forward(ospec, params, images[:1], torch.float32)    # warm-up (allocators, oneDNN)
            t0 = time.perf_counter()
            OF.forward(ospec, params, images[:1], torch.float32)
            dt = time.perf_counter() - t0
            if dt < best[0]:
                best = (dt, threads)
            if time.perf_counter() - t_start > seconds:
                break
        cores = best[1]
        torch.set_num_threads(cores)
        t0 = time.perf_counter()
        done = 0
        while True:
            OF.forward(ospec, params, images, torch.float32)
            done += crops
            el = time.perf_counter() - t0
            if el >= seconds or done >= 64 * crops:
                break
    return {'value': round(done / el, 3), 'unit': 'crops/s', 'cores': cores, 'kind': 'port',
            'sample': f'{done} crops ({done // crops} passes of {crops}) of the same RN{spec.arch}-s{spec.stride} '
                      f'graph in {el:.1f} s: oracle/forward.py, PyTorch-CPU fp32, {cores} threads (fastest of 8..{avail} on this host); '
                      'not TensorFlow (reference CPU path cannot run here)'}


def main():
    args = parse_args()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f'bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks', file=sys.stderr)
            sys.exit(2)
    if not torch.cuda.is_available():
        print('bench.py: no HIP device visible; the hot path has no CPU fallback', file=sys.stderr)
        sys.exit(2)
    import torch.distributed as dist
    # debug overrides (single-GPU boxes): METRO_BENCH_SAME_GPU=1 puts every rank on cuda:0,
    # METRO_BENCH_BACKEND=gloo swaps RCCL for gloo so the N>1 control flow can be exercised anywhere
    same_gpu = os.environ.get('METRO_BENCH_SAME_GPU') == '1'
    backend = os.environ.get('METRO_BENCH_BACKEND', 'nccl')
    device = torch.device('cuda', 0 if same_gpu else local_rank)
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from metro_pose3d_amd import ModelSpec, synth
    from metro_pose3d_amd.engine import Engine
    from metro_pose3d_amd import _lib

    spec = ModelSpec(args.arch, args.stride, args.dataset)
    params = synth.make_params(spec.arch, spec.n_head_channels, spec.base_width, seed=0,
                               logit_gain=synth.logit_gain_for(spec.arch, spec.stride))
    eng = Engine(spec, params, args.precision, max_batch=args.batch, device=device)
    b = args.batch
    # each rank gets ITS OWN crops (seeded by rank): weak scaling, global batch = b * world
    images = torch.from_numpy(synth.make_images(b, spec.proc_side, seed=1234 + rank)).to(device)
    jout = spec.skeleton.n_out
    local = torch.empty((b, jout, 3), dtype=torch.float32, device=device)
    gatherer = None
    if world > 1:
        from metro_pose3d_amd.dist import OverlappedPoseGather
        gatherer = OverlappedPoseGather(b, jout, world, device)
    step_no = [0]

    def step():
        if world > 1:
            # one ncclAllGather (RCCL over xGMI) per step, overlapped with the next step's forward
            i = step_no[0]
            eng.forward(images, out=gatherer.local_buffer(i))
            gatherer.submit(i)
            step_no[0] = i + 1
        else:
            eng.forward(images, out=local)

    for _ in range(args.warmup):
        step()
    if world > 1:
        gatherer.finish()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    if world > 1:
        gatherer.finish()                                     # every gather of the timed steps has completed
        local = gatherer.local[(step_no[0] - 1) % gatherer.depth]
    ev1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gpu_ms = ev0.elapsed_time(ev1)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if not torch.isfinite(local).all():
        print('bench.py: non-finite poses', file=sys.stderr)
        sys.exit(3)

    # ---- per-kernel durations: HIP events around every launch, on the launch stream ----------
    reps = 5
    layer_ms = eng.forward_timed(images, reps=reps)
    infos = eng.layer_infos()
    conv_ms_raw = sum(ms for ms, li in zip(layer_ms, infos) if li.kind == _lib.LAYER_CONV)
    # An event pair around EVERY launch also times the ~2-3 us dependent-launch boundary it creates, so
    # the per-layer sum (2.03 ms) exceeds the un-instrumented forward (1.83 ms, = the rocprofv3 kernel
    # total).  The conv share of the per-layer breakdown is therefore applied to the forward time
    # measured by ONE event pair around the timed steps on the same stream: this reproduces the
    # rocprofv3 kernel-trace average for the conv kernels (profiles/*_kernel_stats.csv) within ~1 %.
    conv_ms = conv_ms_raw * (gpu_ms / args.steps) / float(layer_ms.sum())
    conv_flops = sum(li.flops_per_image for li in infos if li.kind == _lib.LAYER_CONV) * b
    n_conv = sum(1 for li in infos if li.kind == _lib.LAYER_CONV)
    achieved_tflops = conv_flops / (conv_ms * 1e-3) / 1e12
    if args.layer_report and rank == 0:
        with open(args.layer_report, 'w') as f:
            f.write(f'# per-layer HIP-event timing, batch {b}, {args.precision}, mean of {reps} passes\n')
            f.write('layer\tkind\tms\tGFLOP\tTFLOP/s\tout_MB\n')
            for ms, li in zip(layer_ms, infos):
                gf = li.flops_per_image * b / 1e9
                f.write(f'{li.name.decode()}\t{li.kind}\t{ms:.4f}\t{gf:.3f}\t{(gf / ms if ms > 0 else 0):.1f}\t'
                        f'{li.out_bytes_per_image * b / 1e6:.2f}\n')
            f.write(f'TOTAL\t-\t{layer_ms.sum():.4f}\t{conv_flops / 1e9:.3f}\t{conv_flops / 1e9 / layer_ms.sum():.1f}\t-\n')

    # HBM traffic of the conv launches comes from a COMMITTED rocprofv3 PMC run of this same command
    # (counters cannot be collected from inside the process being profiled): profiles/*_pmc_traffic.json
    traffic = None
    traffic_note = None
    if rank == 0 and (args.arch, args.stride, args.dataset, b, args.precision) == (50, 16, 'h36m', 64, 'f16'):
        import glob
        files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_traffic.json')))
        if files:
            with open(files[-1]) as f:
                t = json.load(f)
            traffic = t['conv_hbm_bytes_per_forward']
            traffic_note = (f'HBM bytes per forward summed over the {t["conv_launches"]} conv launches, '
                            f'(2*FETCH_SIZE + WRITE_SIZE)*1024 from {os.path.basename(files[-1])} '
                            f'(rocprofv3 --pmc, separate passes; avg per launch {t["conv_hbm_bytes_per_launch_avg"]:.3e} B; '
                            f'see DESIGN.md section 4 for the algorithmic minimum)')
    if rank == 0:
        total_crops = b * world * args.steps
        ms_per_step = elapsed * 1e3 / args.steps
        value = total_crops / elapsed
        out = {
            'metric': 'crops/sec (256x256, RN50 stride16)' if (args.arch, args.stride) == (50, 16)
                      else f'crops/sec (256x256, RN{args.arch} stride{args.stride})',
            'value': round(value, 2), 'unit': 'crops/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None,
            'dtype': {'f16': 'f16', 'f32': 'f32 storage, f64 accumulate', 'f64': 'f64'}[args.precision],
            'data': 'synthetic (seeded random weights + uniform [0,1) crops; no released weights offline)',
            'config': {'workload': f'RN{args.arch}-s{args.stride}-J{spec.skeleton.n_head} {args.dataset}, '
                                   f'batch {b}/GPU, 256x256x3 fp32 NHWC in HBM -> poses [B,{jout},3] mm',
                       'global_batch': b * world, 'per_gpu_batch': b,
                       'parallelism': f'dp{world} (batch-sharded, one all-gather of poses)' if world > 1 else 'single GPU',
                       'gflop_per_crop': round(eng.flops_per_image / 1e9, 3)},
            'gpu_ms_per_step_events': round(gpu_ms / args.steps, 4),
            'whole_path_tflops': round(eng.flops_per_image * value / world / 1e12, 2),
            'roofline': {'bound': 'mfma', 'kernel': f'conv kernels: conv_igemm_f16_dma, conv3x3_f16_slab, conv_pw64 (weight-stationary), stem_pool_f16 ({n_conv} launches per forward)',
                         'achieved': round(achieved_tflops, 2), 'peak': PEAK_F16_DENSE_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': round(achieved_tflops / PEAK_F16_DENSE_TFLOPS, 4), 'traffic': traffic,
                         'traffic_note': traffic_note,
                         'ms_per_forward_in_kernel': round(conv_ms, 4),
                         'ms_per_forward_in_kernel_event_pairs_raw': round(conv_ms_raw, 4),
                         'algorithmic_gflop_per_forward': round(conv_flops / 1e9, 3)},
        }
        if world == 1 and args.cpu_seconds > 0:
            out['cpu_baseline'] = cpu_baseline(spec, params, args.cpu_seconds, args.cpu_crops)
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
