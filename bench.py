#!/usr/bin/env python3
"""Throughput of the MeTRo inference hot path on MI355X: crops/s on synthetic 256x256 batches.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
torch.distributed.run with one rank per GPU (RCCL).  A "step" is one pass of the hot path over
one resident batch per GPU -- fp32 NHWC crops in HBM -> poses [B, Jout, 3] in HBM -- plus, for
N > 1, the one all-gather of the pose outputs.  Weak scaling: the per-GPU batch is fixed.

Workload (BASELINE.json configs[1], the config the metric is quoted on): ResNet-50, stride 16,
17 H36M joints, batch 64 per GPU, fp16 compute (the reference's default dtype, options.py:73),
seeded synthetic weights and crops (metro_pose3d_amd/synth.py).  For N > 1 the default dataset is
the 19-joint head of BASELINE.json configs[2] (`many19`, 64 crops per GPU = batch 512 on 8 GPUs).

One JSON line on stdout from rank 0, with
  roofline     -- the conv launches of the forward: algorithmic FLOPs (2*MACs) / their
                  summed durations from HIP events recorded on the launch stream, against the
                  2.5 PFLOP/s dense fp16 MFMA peak; `traffic` = HBM bytes from the committed
                  rocprofv3 PMC run of the same kernels (refused when the kernels changed since);
  cpu_baseline -- this repo's CPU restatement of the same graph (oracle/, PyTorch CPU fp32) timed on
                  a bounded sample in the same run, best thread count and 1 thread.  It is NOT
                  TensorFlow: the reference's own CPU path cannot run here (no TF 1.13, no .pb);
  accuracy     -- max |dmm| of the benchmarked f16 mode and of the f64 parity mode against the fp64
                  oracle on a few crops (outside the timed region), next to the distance the fp16
                  arithmetic model of the graph (oracle/f16emu.py) has itself;
  parity_mode  -- crops/s of the f64 parity mode on the same batch (outside the timed region);
  b256         -- (N = 1) the same workload at batch 256, the batch BASELINE.json's north star
                  quotes its roofline target on: crops/s, ms/step and its own roofline object;
  merged53     -- (N = 1) RN50-s16 with the 53-joint head of the released many_* exports, batch 64 (next to c3_shard);
  c3_shard, c4_shard, c5_shard -- (N = 1) one GPU's shard of BASELINE.json configs[2..4] (RN50-s16-J19
                  batch 512/8, RN101-s8-J19 batch 256/8, RN50-s4-J17 batch 128/8), each with its roofline;
  boundary     -- 256 crops through `estimate_pose` itself (the drop-in call of reference inference.py:31-43, model file ->
                  poses), Python overhead included; N > 1: the same call on every rank with the global batch -- sharded by
                  image inside the call, one RCCL all-gather of the poses;
  softargmax_hbm -- (N = 1) the stand-alone soft-argmax on resident fp32 logits at the configs[4] and configs[1] volumes with
                  an HBM roofline object (`bound: "hbm"`), the second number SURVEY 8(d) asks for.
"""
from __future__ import annotations

import argparse
import glob
import hashlib
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
_PEAKS = {}                       # measured_peaks() result of this process (one measurement per run)


def measured_peaks():
    """the peaks "must be confirmed on the box ... and the confirmed values written next to every reported
    fraction".  tools/peak_probe.hip (a yardstick library, not part of the product): a register-resident
    v_mfma_f32_32x32x16_f16 loop on all CUs for >= 2 ms on zeros / N(0,1) / relu(N(0,1)) x He-weight operands (TFLOP/s and the
    shader clock held), and a read-only and a copy kernel on 1 GiB (beyond the 256 MiB Infinity Cache).  `frac` stays against
    the nominal 2 500 TFLOP/s / 8 000 GB/s; these are what THIS chip delivers to a kernel with nothing else in it."""
    if _PEAKS:
        return _PEAKS
    import ctypes as C
    path = os.path.join(ROOT, 'tools', 'libmetro_probe.so')
    try:
        lib = C.CDLL(path)
    except OSError as e:
        _PEAKS.update({'error': f'{path}: {e} (built by __graft_entry__.build() / tools/build_probe.sh)'})
        return _PEAKS
    lib.metro_probe_mfma_f16.argtypes = [C.c_int, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.metro_probe_hbm.argtypes = [C.c_int, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.metro_probe_last_error.restype = C.c_char_p
    torch.cuda.synchronize()
    rec = {'source': 'tools/peak_probe.hip, measured in this run on this GPU'}
    for kind, name in ((1, 'mfma_f16_random'), (2, 'mfma_f16_relu_x_he'), (0, 'mfma_f16_zeros')):
        tf, mhz, ms = C.c_double(), C.c_double(), C.c_double()
        if lib.metro_probe_mfma_f16(kind, 2.0, C.byref(tf), C.byref(mhz), C.byref(ms)) != 0:
            rec[name] = {'error': lib.metro_probe_last_error().decode()}
            continue
        rec[name] = {'tflops': round(tf.value, 1), 'sclk_mhz': round(mhz.value), 'ms': round(ms.value, 2)}
    # round 6: the same loop on the other instruction forms, on the operands the net multiplies (relu x He): a register-resident
    # 16x16x32 loop holds a higher clock -- inside the real kernels both forms measured SLOWER (NOTES_dead_ends.md, Round 6)
    if hasattr(lib, 'metro_probe_mfma_f16_variant'):
        lib.metro_probe_mfma_f16_variant.argtypes = [C.c_int, C.c_int, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        for variant, name in ((1, 'mfma_f16_16x16x32_relu_x_he'), (2, 'mfma_f16_32x32x16_agpr_acc_relu_x_he'), (3, 'mfma_f16_16x16x32_agpr_acc_relu_x_he')):
            tf, mhz, ms = C.c_double(), C.c_double(), C.c_double()
            if lib.metro_probe_mfma_f16_variant(2, variant, 2.0, C.byref(tf), C.byref(mhz), C.byref(ms)) != 0:
                rec[name] = {'error': lib.metro_probe_last_error().decode()}
                continue
            rec[name] = {'tflops': round(tf.value, 1), 'sclk_mhz': round(mhz.value), 'ms': round(ms.value, 2)}
    for kind, name in ((0, 'hbm_read'), (1, 'hbm_copy')):
        for size, tag in ((1 << 30, '_1GiB'), (64 << 20, '_64MiB')):
            tb, us = C.c_double(), C.c_double()
            if lib.metro_probe_hbm(kind, size, C.byref(tb), C.byref(us)) != 0:
                rec[name + tag] = {'error': lib.metro_probe_last_error().decode()}
                continue
            rec[name + tag] = {'gb_per_s': round(tb.value * 1e3, 1), 'us': round(us.value, 1)}
    rec['note'] = ('mfma_*: v_mfma_f32_32x32x16_f16 from registers only, 8 A x 8 B fragments rotating, 8 accumulators, 2 waves per SIMD '
                   'on every CU; random = N(0,1) fp16 operands (what bench data toggles like), zeros = the same instructions '
                   'on zero operands (the clock the power budget allows without data toggling).  hbm_*: 16-byte accesses, '
                   'read + written bytes counted; _64MiB sits inside the 256 MiB Infinity Cache.')
    _PEAKS.update(rec)
    return _PEAKS


def peak_measured_for(bound: str):
    """The measured-ceiling sub-object of a roofline record (None-safe: a missing probe library is reported, not fatal)."""
    pk = measured_peaks()
    if 'error' in pk:
        return {'error': pk['error']}
    if bound == 'mfma':
        # the forward is MFMA-bound in some launches and HBM-bound in others: both measured ceilings travel with its fraction
        return ({k: pk[k] for k in ('mfma_f16_random', 'mfma_f16_relu_x_he', 'mfma_f16_zeros', 'mfma_f16_16x16x32_relu_x_he',
                                    'mfma_f16_32x32x16_agpr_acc_relu_x_he', 'mfma_f16_16x16x32_agpr_acc_relu_x_he') if k in pk} | {'unit': 'TFLOP/s'} |
                {'hbm': {k: pk[k] for k in ('hbm_read_1GiB', 'hbm_copy_1GiB', 'hbm_read_64MiB', 'hbm_copy_64MiB') if k in pk} | {'unit': 'GB/s'},
                 'source': pk['source']})
    return {k: pk[k] for k in ('hbm_read_1GiB', 'hbm_copy_1GiB', 'hbm_read_64MiB', 'hbm_copy_64MiB') if k in pk} | {'unit': 'GB/s', 'source': pk['source']}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=64, help='crops per GPU per step')
    ap.add_argument('--arch', type=int, default=50)
    ap.add_argument('--stride', type=int, default=16)
    ap.add_argument('--dataset', type=str, default=None,
                    help='default: h36m (17 joints, configs[1]) on 1 GPU, many19 (19 joints, configs[2]) on N > 1')
    ap.add_argument('--precision', type=str, default='f16', choices=['f16', 'f32', 'f32m', 'f64'])
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='budget of the CPU baseline leg (0 = skip)')
    ap.add_argument('--cpu-crops', type=int, default=8)
    ap.add_argument('--layer-report', type=str, default=None, help='write the per-layer table to this file')
    ap.add_argument('--diag-zero-data', action='store_true',
                    help='DIAGNOSTIC, not a result: all-zero weights and crops (same instructions, no data toggling) -- how far '
                         'the step is from the power budget that sets the clock (MI355X_MICROARCH.md, DVFS give-back)')
    ap.add_argument('--no-extras', action='store_true',
                    help='skip the accuracy / parity-mode / batch-256 legs (profiling runs)')
    return ap.parse_args()


def kernels_sha16() -> str:
    """Hash of every source the library is built from (metro_pose3d_amd/build.py SOURCES + HEADERS, nothing else that may lie
    in csrc/): a committed PMC traffic file is only valid for these."""
    from metro_pose3d_amd.build import CSRC, HEADERS, SOURCES
    h = hashlib.sha256()
    for f in sorted(os.path.normpath(os.path.join(CSRC, x)) for x in list(SOURCES) + list(HEADERS)):
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def workload_key(arch: int, stride: int, dataset: str, batch: int) -> str:
    return f'rn{arch}-s{stride}-{dataset}-b{batch}'


def host_cpu():
    """(logical cores visible to this process, logical cores of the host, CPU model string)."""
    avail = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    model = 'unknown'
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.lower().startswith('model name'):
                    model = line.split(':', 1)[1].strip()
                    break
    except OSError:
        pass
    return avail, os.cpu_count() or avail, model


def oracle_spec(spec):
    from oracle.spec import OracleSpec
    return OracleSpec(arch=spec.arch, stride=spec.stride, dataset=spec.dataset, depth=spec.depth,
                      centered_stride=spec.centered_stride, proc_side=spec.proc_side,
                      box_size_mm=spec.box_size_mm, base_width=spec.base_width)


def cpu_baseline(spec, params, seconds: float, crops: int):
    """Oracle (CPU restatement) timed on host cores.  kind = 'port'."""
    from oracle import forward as OF
    from metro_pose3d_amd import synth
    ospec = oracle_spec(spec)
    avail, host_cores, cpu_model = host_cpu()
    images = synth.make_images(crops, spec.proc_side, seed=99)
    t_start = time.perf_counter()
    with torch.no_grad():
        # 1 thread: the reference pins 1 intra-op + 1 inter-op thread in its own sessions (helpers.py:123-125)
        torch.set_num_threads(1)
        OF.forward(ospec, params, images[:1], torch.float32)        # warm-up (allocators, oneDNN)
        t0 = time.perf_counter()
        n1 = 0
        while True:
            OF.forward(ospec, params, images[:1], torch.float32)
            n1 += 1
            if time.perf_counter() - t0 > min(3.0, seconds / 4) or n1 >= 4:
                break
        one_thread = n1 / (time.perf_counter() - t0)
        # the thread count the host actually runs this graph fastest with (1 crop each)
        best = (float('inf'), 1)
        sweep = {}
        # up to 96 threads: beyond that torch's CPU convolutions thrash (a single 256-thread pass of this graph took > 100 s on the
        # 256-thread EPYC of the GPU box, round 6) -- the sweep shows throughput FALLING from 16-32 threads on, which is the answer
        # to "what do all cores give"  without spending minutes of the bench on it
        for threads in sorted({t for t in (8, 16, 32, 64, min(avail, 96)) if t <= avail}):
            torch.set_num_threads(threads)
            OF.forward(ospec, params, images[:1], torch.float32)
            t0 = time.perf_counter()
            OF.forward(ospec, params, images[:1], torch.float32)
            dt = time.perf_counter() - t0
            sweep[threads] = round(1.0 / dt, 2)
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
            'host_logical_cores': host_cores, 'host_cores_visible_to_this_process': avail, 'host_cpu_model': cpu_model,
            'cores_note': '`cores` = the torch threads of the timed run (the fastest thread count tried); the host has '
                          f'{host_cores} logical cores ({avail} visible to this process)',
            'one_thread_crops_per_s': round(one_thread, 3),
            'single_crop_crops_per_s_by_threads': sweep,
            'most_threads_tried': max(sweep) if sweep else None,
            'all_cores_note': f'throughput of the oracle on single crops by thread count (above): it peaks at {cores} threads and FALLS '
                              f'beyond (torch-CPU convolutions stop scaling; with all {avail} threads a pass takes minutes), so `value` is '
                              f'the fastest count, not an all-core figure',
            'sample': f'{done} crops ({done // crops} passes of {crops}) of the same RN{spec.arch}-s{spec.stride} '
                      f'graph in {el:.1f} s: oracle/forward.py, PyTorch-CPU fp32, {cores} threads (fastest of 8..{avail} on this host: {cpu_model}); '
                      f'1 thread: {n1} single-crop passes; not TensorFlow (reference CPU path cannot run here)'}


def timed_steps(step, steps: int, device, world: int, dist, finish=None):
    """EXACTLY `steps` steps bracketed by barrier + synchronize; per-step HIP events on the launch stream."""
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(steps):
        step()
        evs[i + 1].record()
    if finish is not None:
        finish()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    per = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(steps)])
    gpu_ms = evs[0].elapsed_time(evs[steps])
    return elapsed, gpu_ms, per


def roofline_of(eng, images, gpu_ms_per_step: float, reps: int, layer_report=None, label=''):
    """Conv launches: algorithmic FLOPs / their share of the forward time (HIP events on the launch stream).

    An event pair around EVERY launch also times the ~2-3 us dependent-launch boundary it creates, so the per-layer sum
    exceeds the un-instrumented forward (= the rocprofv3 kernel total).  The conv SHARE of the per-layer breakdown is
    therefore applied to the forward time measured by the event pairs around whole steps on the same stream: this
    reproduces the rocprofv3 kernel-trace average of the conv kernels (profiles/*_kernel_stats.csv) within ~1 %."""
    from metro_pose3d_amd import _lib
    b = images.shape[0]
    layer_ms = eng.forward_timed(images, reps=reps)
    infos = eng.layer_infos()
    conv = [li.kind == _lib.LAYER_CONV for li in infos]
    conv_ms_raw = sum(ms for ms, c in zip(layer_ms, conv) if c)
    conv_ms = conv_ms_raw * gpu_ms_per_step / float(layer_ms.sum())
    conv_flops = sum(li.flops_per_image for li, c in zip(infos, conv) if c) * b
    n_conv = sum(conv)
    achieved = conv_flops / (conv_ms * 1e-3) / 1e12
    algo_bytes = sum(li.algo_act_bytes_per_image * b + li.algo_param_bytes for li, c in zip(infos, conv) if c)
    if layer_report:
        with open(layer_report, 'w') as f:
            f.write(f'# per-layer HIP-event timing, batch {b}, mean of {reps} passes {label}\n')
            f.write('layer\tkind\tms\tGFLOP\tTFLOP/s\tout_MB\talgo_MB\talgo_GB/s\n')
            for ms, li in zip(layer_ms, infos):
                gf = li.flops_per_image * b / 1e9
                ab = li.algo_act_bytes_per_image * b + li.algo_param_bytes
                f.write(f'{li.name.decode()}\t{li.kind}\t{ms:.4f}\t{gf:.3f}\t{(gf / ms if ms > 0 else 0):.1f}\t'
                        f'{li.out_bytes_per_image * b / 1e6:.2f}\t{ab / 1e6:.2f}\t{(ab / 1e6 / ms if ms > 0 else 0):.0f}\n')
            f.write(f'TOTAL\t-\t{layer_ms.sum():.4f}\t{conv_flops / 1e9:.3f}\t{conv_flops / 1e9 / layer_ms.sum():.1f}\t-\t{algo_bytes / 1e6:.1f}\t-\n')
    # the kernel families this forward dispatches at this batch (dry run of the plan's dispatch), with their time shares
    fam_ms = {}
    for ms, kid, c in zip(layer_ms, eng.layer_kernels(b), conv):
        if c:
            for one in kid.split(' & '):
                fam = one.split('<')[0]
                fam_ms[fam] = fam_ms.get(fam, 0.0) + ms / len(kid.split(' & '))
    fams = ', '.join(f'{k} {100 * v / conv_ms_raw:.0f}%' for k, v in sorted(fam_ms.items(), key=lambda kv: -kv[1]))
    pk = peak_measured_for('mfma')
    rnd = pk.get('mfma_f16_random', {}).get('tflops') if isinstance(pk, dict) else None
    return {'bound': 'mfma',
            'kernel': f'the {n_conv} conv launches of the forward ({len(infos)} launches per forward = {n_conv} conv + '
                      f'{len(infos) - n_conv} soft-argmax finalize; share of the conv time by kernel family: {fams})',
            'achieved': round(achieved, 2), 'peak': PEAK_F16_DENSE_TFLOPS, 'unit': 'TFLOP/s',
            'frac': round(achieved / PEAK_F16_DENSE_TFLOPS, 4), 'peak_measured': pk,
            'frac_of_measured_random_data_peak': round(achieved / rnd, 4) if rnd else None,
            'traffic': None, 'traffic_note': None,
            'algorithmic_min_bytes': int(algo_bytes),
            'algorithmic_min_bytes_note': 'every tensor each conv launch touches, counted once per launch (MetroLayerInfo.algo_*): '
                                          'what this launch set moves if nothing is re-read',
            'hbm_frac_at_algorithmic_bytes': round(algo_bytes / (conv_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
            'ms_per_forward_in_kernel': round(conv_ms, 4),
            'ms_per_forward_in_kernel_event_pairs_raw': round(conv_ms_raw, 4),
            'algorithmic_gflop_per_forward': round(conv_flops / 1e9, 3)}


def attach_traffic(roof: dict, workload) -> None:
    """roofline.traffic from the committed profiles/*_pmc_traffic.json of this workload (a workload_key(), or a batch of the
    default RN50-s16-h36m workload) that was collected for the kernel sources the library is built from; a file for other
    sources is refused (traffic null + a note).  Selection is by content, never by file time (arbitrary after a checkout)."""
    if isinstance(workload, int):
        workload = workload_key(50, 16, 'h36m', workload)
    files = []
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_traffic.json'))):
        with open(f) as fh:
            t = json.load(fh)
        if t.get('workload', workload_key(50, 16, 'h36m', int(t.get('batch', 64)))) == workload:
            files.append((f, t))
    if not files:
        return
    sha = kernels_sha16()
    match = [(f, t) for f, t in files if t.get('kernels_sha16') == sha]
    if match:
        path, t = match[-1]                       # several files for the same sources: the last by name
        roof['traffic'] = t['conv_hbm_bytes_per_forward']
        roof['traffic_note'] = (f'HBM bytes per forward summed over the {t["conv_launches"]} conv launches, '
                                f'(2*FETCH_SIZE + WRITE_SIZE)*1024 from {os.path.basename(path)} '
                                f'(rocprofv3 --pmc, separate passes, kernels {sha}; avg per launch '
                                f'{t["conv_hbm_bytes_per_launch_avg"]:.3e} B)')
    else:
        path, t = files[-1]
        roof['traffic_note'] = (f'{os.path.basename(path)} was collected for kernels {t.get("kernels_sha16")}, '
                                f'the library is now built from {sha}: stale, not reported (re-run profiles/collect.sh)')


def side_workload(device, dist, arch, stride, dataset, batch, steps, warmup, what, seed=1234):
    """One more workload on this GPU outside the main timed region (N = 1): its own engine, crops and roofline."""
    from metro_pose3d_amd import ModelSpec, synth
    from metro_pose3d_amd.engine import Engine
    spec = ModelSpec(arch, stride, dataset)
    params = synth.make_params(spec.arch, spec.n_head_channels, spec.base_width, seed=0,
                               logit_gain=synth.logit_gain_for(spec.arch, spec.stride))
    eng = Engine(spec, params, 'f16', max_batch=batch, device=device)
    img = torch.from_numpy(synth.make_images(batch, spec.proc_side, seed=seed)).to(device)
    out = torch.empty((batch, spec.skeleton.n_out, 3), dtype=torch.float32, device=device)
    for _ in range(warmup):
        eng.forward(img, out=out)
    el, gms, per = timed_steps(lambda: eng.forward(img, out=out), steps, device, 1, dist)
    roof = roofline_of(eng, img, gms / steps, reps=3)
    attach_traffic(roof, workload_key(arch, stride, dataset, batch))
    rec = {'workload': f'RN{arch}-s{stride}-J{spec.skeleton.n_head} {dataset}, batch {batch} on 1 GPU ({what})',
           'value': round(batch * steps / el, 2), 'unit': 'crops/s', 'steps': steps, 'warmup': warmup,
           'ms_per_step': round(el * 1e3 / steps, 4), 'gpu_ms_per_step_median': round(float(np.median(per)), 4),
           'gflop_per_crop': round(eng.flops_per_image / 1e9, 3), 'finite': bool(torch.isfinite(out).all()), 'roofline': roof}
    del eng, img, out
    torch.cuda.empty_cache()
    return rec


def boundary_leg(device, dist, spec, params, batch, steps, warmup):
    """`estimate_pose(images, model_path)` -- the drop-in call (reference inference.py:31-43) -- timed as a caller sees it:
    model file on disk (loaded and planned once, cached), crops resident on the GPU, poses on the GPU."""
    import tempfile
    from metro_pose3d_amd import save_model, synth
    from metro_pose3d_amd import inference as INF
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, 'model.npz')
        save_model(path, spec, params)
        img = torch.from_numpy(synth.make_images(batch, spec.proc_side, seed=1234)).to(device)
        for _ in range(warmup):
            poses, _, _ = INF.estimate_pose(img, path)
        el, gms, per = timed_steps(lambda: INF.estimate_pose(img, path), steps, device, 1, dist)
        ok = bool(torch.isfinite(poses).all())
        INF.clear_cache()
    return {'call': f'metro_pose3d_amd.inference.estimate_pose(images[{batch},256,256,3] on the GPU, model_path) -> poses on the GPU',
            'value': round(batch * steps / el, 2), 'unit': 'crops/s', 'steps': steps, 'ms_per_call': round(el * 1e3 / steps, 4),
            'gpu_ms_per_call_median': round(float(np.median(per)), 4), 'finite': ok,
            'note': 'one metro_forward(n = 256) per call (the engine is planned for the batch the call is given); includes the '
                    'Python argument checks, output allocation, the names/edges arrays and the non-finite screen of every call '
                    '(metro_forward_status: one stream synchronisation, like the reference\'s blocking sess.run)'}


def sharded_boundary_leg(device, dist, spec, params, per_gpu, world, rank, steps, warmup):
    """N > 1: the SAME drop-in call on every rank with the same global batch -- estimate_pose shards it by image, forwards the
    rank's shard and all-gathers the poses (RCCL under backend nccl).  Max over ranks, like the main timed region."""
    import tempfile
    from metro_pose3d_amd import save_model, synth
    from metro_pose3d_amd import inference as INF
    n = per_gpu * world
    td = tempfile.mkdtemp(prefix=f'metro_bench_r{rank}_')
    path = os.path.join(td, 'model.npz')
    save_model(path, spec, params)
    img = torch.cat([torch.from_numpy(synth.make_images(per_gpu, spec.proc_side, seed=1234 + r)) for r in range(world)]).to(device)
    for _ in range(warmup):
        poses, _, _ = INF.estimate_pose(img, path)
    el, gms, per = timed_steps(lambda: INF.estimate_pose(img, path), steps, device, world, dist)
    t = torch.tensor([el], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    el = float(t.item())
    ok = bool(torch.isfinite(poses).all()) and tuple(poses.shape) == (n, spec.skeleton.n_out, 3)
    INF.clear_cache()
    os.remove(path)
    os.rmdir(td)
    return {'call': f'estimate_pose(images[{n},256,256,3], model_path) on each of {world} ranks -> all {n} poses on every rank',
            'value': round(n * steps / el, 2), 'unit': 'crops/s', 'steps': steps, 'ms_per_call': round(el * 1e3 / steps, 4),
            'crops_per_rank': per_gpu, 'complete_and_finite': ok,
            'note': 'each rank forwards dist.shard_range(N, rank, world) and joins ONE all-gather of [N/G, Jout, 3] fp32; '
                    'not overlapped with the next call (the call returns the gathered poses), non-finite screen on'}


def softargmax_hbm_leg(device, dist, stride, dataset, crops, steps, warmup, what):
    """The HBM-bound piece of the path on its own: stand-alone metro_softargmax on resident fp32 logits
    [n, S, S, 8J] -> poses, i.e. tfu.softmax + decode_heatmap + heatmap_to_metric + root_relative + gather (reference
    volumetric.py:227-235, tfu.py:466-499).  Algorithmic bytes: 4*8*J*S*S read + 12*Jout written per crop."""
    import ctypes as C
    from metro_pose3d_amd import ModelSpec, _lib
    spec = ModelSpec(50, stride, dataset)
    lib = _lib.load()
    s, c, jo = spec.heatmap_side, spec.n_head_channels, spec.skeleton.n_out
    g = torch.Generator(device='cpu').manual_seed(5)
    logits = (torch.randn((crops, s, s, c), generator=g) * 4.0).to(device)
    scratch = torch.empty(lib.metro_softargmax_scratch_bytes(crops, s, spec.skeleton.n_head), dtype=torch.uint8, device=device)
    out = torch.empty((crops, jo, 3), dtype=torch.float32, device=device)
    cs = spec.to_c(0)
    stream = torch.cuda.current_stream(device).cuda_stream

    def step():
        _lib.check(lib.metro_softargmax(C.c_void_p(logits.data_ptr()), crops, C.byref(cs), 0, C.c_void_p(scratch.data_ptr()),
                                        C.c_void_p(out.data_ptr()), C.c_void_p(stream)), 'metro_softargmax')
    for _ in range(warmup):
        step()
    el, gms, per = timed_steps(step, steps, device, 1, dist)
    us = float(np.median(per)) * 1e3
    algo = crops * (4 * c * s * s + 12 * jo)
    gbs = algo / (us * 1e-6) / 1e9
    return {'workload': f'stand-alone soft-argmax, {what}: fp32 logits [{crops},{s},{s},{c}] = {crops * 4 * c * s * s / 1e6:.1f} MB -> poses [{crops},{jo},3]',
            'value': round(crops / (us * 1e-6), 1), 'unit': 'crops/s', 'us_per_call_median': round(us, 2), 'steps': steps,
            'finite': bool(torch.isfinite(out).all()),
            'roofline': {'bound': 'hbm', 'kernel': 'softargmax_partial<acc32,logits32> + softargmax_finalize<acc32> (both launches timed together)',
                         'achieved': round(gbs, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': round(gbs / PEAK_HBM_GBS, 4),
                         'peak_measured': peak_measured_for('hbm'), 'traffic': None,
                         'algorithmic_bytes_per_call': int(algo)}}


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

    dataset = args.dataset or ('h36m' if world == 1 else 'many19')
    spec = ModelSpec(args.arch, args.stride, dataset)
    params = synth.make_params(spec.arch, spec.n_head_channels, spec.base_width, seed=0,
                               logit_gain=synth.logit_gain_for(spec.arch, spec.stride))
    if args.diag_zero_data:
        params = {k: np.zeros_like(v) for k, v in params.items()}
    eng = Engine(spec, params, args.precision, max_batch=args.batch, device=device)
    b = args.batch
    # each rank gets ITS OWN crops (seeded by rank): weak scaling, global batch = b * world
    images_np = synth.make_images(b, spec.proc_side, seed=1234 + rank)
    if args.diag_zero_data:
        images_np = np.zeros_like(images_np)
    images = torch.from_numpy(images_np).to(device)
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
    elapsed, gpu_ms, per_step = timed_steps(step, args.steps, device, world, dist,
                                            finish=(gatherer.finish if world > 1 else None))
    if world > 1:
        local = gatherer.local[(step_no[0] - 1) % gatherer.depth]
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if not torch.isfinite(local).all():
        print('bench.py: non-finite poses', file=sys.stderr)
        sys.exit(3)

    roof = roofline_of(eng, images, gpu_ms / args.steps, reps=5,
                       layer_report=args.layer_report if rank == 0 else None, label=f'({args.precision})')

    # HBM traffic of the conv launches comes from a COMMITTED rocprofv3 PMC run of the same kernels (counters cannot be
    # collected from inside the process being profiled): profiles/*_pmc_traffic.json, valid only for the sources it names
    if rank == 0 and args.precision == 'f16' and not args.diag_zero_data:
        attach_traffic(roof, workload_key(args.arch, args.stride, dataset, b))

    out = None
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
            'dtype': {'f16': 'f16', 'f32': 'f32 storage, f64 accumulate', 'f32m': 'f32 (fp32 MFMA)', 'f64': 'f64'}[args.precision],
            'data': 'ALL-ZERO weights and crops: DIAGNOSTIC RUN, NOT A RESULT' if args.diag_zero_data else
                    'synthetic (seeded random weights + uniform [0,1) crops; no released weights offline)',
            'config': {'workload': f'RN{args.arch}-s{args.stride}-J{spec.skeleton.n_head} {dataset}, '
                                   f'batch {b}/GPU, 256x256x3 fp32 NHWC in HBM -> poses [B,{jout},3] mm',
                       'global_batch': b * world, 'per_gpu_batch': b,
                       'parallelism': f'dp{world} (batch-sharded, one all-gather of poses)' if world > 1 else 'single GPU',
                       'gflop_per_crop': round(eng.flops_per_image / 1e9, 3)},
            'gpu_ms_per_step_events': round(gpu_ms / args.steps, 4),
            'gpu_ms_per_step_median': round(float(np.median(per_step)), 4),
            'gpu_ms_per_step_min': round(float(per_step.min()), 4),
            'whole_path_tflops': round(eng.flops_per_image * value / world / 1e12, 2),
            'roofline': roof,
        }

    # ---- legs outside the timed region (rank 0, N = 1): accuracy, parity-mode throughput, batch 256 ----------
    if rank == 0 and world == 1 and not args.no_extras and args.precision == 'f16':
        from oracle import f16emu
        from oracle import forward as OF
        k = min(2, b)
        ospec = oracle_spec(spec)
        with torch.no_grad():
            exact = OF.forward(ospec, params, images_np[:k], torch.float64).numpy()
            emu = f16emu.forward(ospec, params, images_np[:k]).numpy()
        got16 = local[:k].cpu().numpy()
        eng64 = Engine(spec, params, 'f64', max_batch=b, device=device)
        out64 = torch.empty_like(local)
        eng64.forward(images, out=out64)
        torch.cuda.synchronize()
        got64 = out64[:k].cpu().numpy()
        out['max_abs_dmm'] = round(float(np.abs(got16 - exact).max()), 4)
        out['accuracy'] = {
            'crops': k,
            'f16_mode_max_abs_dmm_vs_fp64_oracle': round(float(np.abs(got16 - exact).max()), 4),
            'f16_mode_mean_abs_dmm_vs_fp64_oracle': round(float(np.abs(got16 - exact).mean()), 4),
            'fp16_model_of_the_graph_max_abs_dmm_vs_fp64_oracle': round(float(np.abs(emu - exact).max()), 4),
            'f64_parity_mode_max_abs_dmm_vs_fp64_oracle': float(f'{np.abs(got64 - exact).max():.3e}'),
            'note': 'oracle = oracle/forward.py (fp64 CPU restatement; its control flow and decode are held to the reference\'s own Python executed '
                    'in the build container (tests/golden/ref_schedule_v1.npz), its backbone to the same lines run ON NUMBERS with NumPy op kernels '
                    '(ref_forward_v2.npz, 1e-10); the arithmetic of TensorFlow\'s own kernels stays unpinned: no TF); '
                    'fp16 storage costs a few mm on this synthetic net in ANY implementation (oracle/f16emu.py is the '
                    'one-rounding-per-tensor model); the 1e-3 mm bar is met by the f64 parity mode'}
        psteps = 3
        _, pms, _ = timed_steps(lambda: eng64.forward(images, out=out64), psteps, device, 1, dist)
        out['parity_mode'] = {'precision': 'f64 (fp64 MFMA, fp64 storage)', 'crops_per_s': round(b * psteps / (pms * 1e-3), 1),
                              'ms_per_step': round(pms / psteps, 3), 'steps': psteps}
        del eng64
        # the parity mode at fp32 speed: fp32 matrix cores (v_mfma_f32_32x32x2_f32), fp32 storage -- the arithmetic of the
        # reference's own fp32 graph; its distance to the fp64 oracle next to the CPU fp32 restatement's distance (two correct
        # fp32 implementations differ from exact math by this much; the 1e-3 mm bar is met by the f64 mode only)
        eng32 = Engine(spec, params, 'f32m', max_batch=b, device=device)
        out32 = torch.empty_like(local)
        eng32.forward(images, out=out32)
        torch.cuda.synchronize()
        with torch.no_grad():
            cpu32 = OF.forward(ospec, params, images_np[:k], torch.float32).numpy().astype(np.float64)
        msteps = 5
        _, mms, _ = timed_steps(lambda: eng32.forward(images, out=out32), msteps, device, 1, dist)
        out['parity_mode_fp32'] = {'precision': 'f32m (fp32 MFMA v_mfma_f32_32x32x2_f32, fp32 storage, fp64 soft-argmax)',
                                   'crops_per_s': round(b * msteps / (mms * 1e-3), 1), 'ms_per_step': round(mms / msteps, 3), 'steps': msteps,
                                   'max_abs_dmm_vs_fp64_oracle': float(f'{np.abs(out32[:k].cpu().numpy() - exact).max():.3e}'),
                                   'cpu_fp32_restatement_max_abs_dmm_vs_fp64_oracle': float(f'{np.abs(cpu32 - exact).max():.3e}'),
                                   'tflops': round(eng32.flops_per_image * b * msteps / (mms * 1e-3) / 1e12, 1), 'peak_tflops': 157.3}
        del eng32
        if not args.diag_zero_data:
            # DVFS diagnostic (not a result): the SAME launches on all-zero weights and crops.  The chip clocks to its
            # power budget (MI355X_MICROARCH.md, "DVFS give-back"); the ratio says how much of the step time is the
            # clock the data's switching activity allows rather than stall cycles a better schedule could remove.
            engz = Engine(spec, {k_: np.zeros_like(v_) for k_, v_ in params.items()}, 'f16', max_batch=b, device=device)
            imgz = torch.zeros_like(images)
            for _ in range(3):
                engz.forward(imgz, out=local)
            zs = 20
            _, zms, zper = timed_steps(lambda: engz.forward(imgz, out=local), zs, device, 1, dist)
            out['power_diagnostic'] = {
                'zero_data_gpu_ms_per_step_median': round(float(np.median(zper)), 4), 'steps': zs,
                'ratio_to_timed_run': round(float(np.median(zper)) / float(np.median(per_step)), 3),
                'note': 'same binary, same launches, all-zero weights and crops: not a throughput claim; the MFMA-bound '
                        'launches run ~20 % faster on zeros (clock set by the power budget)'}
            del engz, imgz
        if b != 256 and (args.arch, args.stride, dataset) == (50, 16, 'h36m') and not args.diag_zero_data:
            out['b256'] = side_workload(device, dist, 50, 16, 'h36m', 256, 10, 3,
                                        'the batch the north star quotes its roofline target on')
            # the north-star batch INSIDE the main roofline object (the driver keeps `roofline`, not the sub-records)
            r256 = out['b256']['roofline']
            out['roofline']['b256'] = {
                'value': out['b256']['value'], 'unit': 'crops/s', 'ms_per_step': out['b256']['ms_per_step'],
                'achieved': r256['achieved'], 'frac': r256['frac'], 'traffic': r256['traffic'],
                'algorithmic_min_bytes': r256['algorithmic_min_bytes'], 'steps': out['b256']['steps'],
                'note': 'the same workload at batch 256 on this GPU (the batch BASELINE.json north_star quotes its MFMA target on), '
                        'timed outside the main region; full record: top-level `b256`'}
            # one GPU's shard of BASELINE.json configs[2..4] (the 8-GPU runs are the driver's; a shard is what a rank computes)
            out['c3_shard'] = side_workload(device, dist, 50, 16, 'many19', 64, 20, 3, 'configs[2]: batch 512 sharded over 8 GPUs')
            out['c4_shard'] = side_workload(device, dist, 101, 8, 'many19', 32, 10, 3, 'configs[3]: batch 256 sharded over 8 GPUs')
            out['c5_shard'] = side_workload(device, dist, 50, 4, 'h36m', 16, 10, 3, 'configs[4]: batch 128 sharded over 8 GPUs')
            # the released `many_*` exports carry the 53-joint `merged` head (424 channels; reference data/datasets.py:142-154,
            # main.py:119-127): since round 5 on the one-launch head too (three joint groups), logits on chip
            out['merged53'] = side_workload(device, dist, 50, 16, 'merged', 64, 20, 3, 'the 53-joint head of the released many_* exports (not a BASELINE config)')
            out['boundary'] = boundary_leg(device, dist, spec, params, 256, 10, 2)
            # the HBM-bound number SURVEY 8(d) asks for next to the MFMA one: the soft-argmax on its own
            out['softargmax_hbm'] = {'note': 'pure-read launch: 2 x FETCH only.  Yardsticks on the same 285 MB on this chip (tools/hbm_read_probe.py, '
                                             'profiles/r04_hbm_read_probe.txt): torch.sum 3.9 TB/s, torch.max 3.5 TB/s (the ROCm stack\'s own read-only '
                                             'reductions), copy 5.4 TB/s and exp 6.0 TB/s of read + write traffic; MI355X_MICROARCH.md: 6.29 TB/s float4 copy = '
                                             '79 % of the 8 TB/s this fraction is taken of',
                                     'c5_volume': softargmax_hbm_leg(device, dist, 4, 'h36m', 128, 20, 3, 'configs[4] volume (S = 64, J = 17), 128 crops'),
                                     'c2_volume': softargmax_hbm_leg(device, dist, 16, 'h36m', 64, 50, 5, 'configs[1] volume (S = 16, J = 17), 64 crops'),
                                     'c2_volume_b2048': softargmax_hbm_leg(device, dist, 16, 'h36m', 2048, 20, 3, 'configs[1] volume, 2048 crops (same bytes as the C5 case)')}
    if world > 1 and not args.no_extras and args.precision == 'f16':
        rec = sharded_boundary_leg(device, dist, spec, params, b, world, rank, 10, 2)      # collective: every rank takes part
        if rank == 0:
            out['boundary'] = rec
    if rank == 0:
        if world == 1 and args.cpu_seconds > 0:
            out['cpu_baseline'] = cpu_baseline(spec, params, args.cpu_seconds, args.cpu_crops)
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
