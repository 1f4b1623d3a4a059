"""Builds, in-tree with hipcc (gfx950 only):  `python -m metro_pose3d_amd.build`
  libmetro_hip.so            the product: everything metro_forward can reach + the per-kernel test entries (include/metro_hip.h)
  libmetro_experimental.so   kernels metro_forward never dispatches (csrc/experimental/, loaded by tools/ probes and one test)
  tools/libmetro_probe.so    measured-ceiling probes for bench.py (tools/peak_probe.hip: MFMA-only loop, HBM read / copy)"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB_PATH = os.path.join(HERE, 'libmetro_hip.so')
SOURCES = ['conv_igemm_f16_dma.hip', 'conv_gemm4w.hip', 'conv3x3_f16_slab.hip', 'conv3x3_c64.hip', 'conv_pw64.hip', 'conv_b1.hip', 'conv_pws.hip', 'head_f16.hip', 'stem_pool_f16.hip', 'conv_igemm_f64acc.hip', 'conv_igemm_f32.hip', 'pool_softargmax.hip', 'eval_metrics.hip', 'heads.hip', 'plan.cpp']
EXPERIMENTAL_LIB_PATH = os.path.join(HERE, 'libmetro_experimental.so')
EXPERIMENTAL_SOURCES = [os.path.join('experimental', f) for f in ('conv_gemm8p.hip', 'conv_gemm4d.hip', 'exp_abi.cpp')]
EXPERIMENTAL_HEADERS = [os.path.join('experimental', 'metro_experimental.h')]
PROBE_SRC = os.path.normpath(os.path.join(HERE, '..', 'tools', 'peak_probe.hip'))
PROBE_LIB_PATH = os.path.normpath(os.path.join(HERE, '..', 'tools', 'libmetro_probe.so'))
HEADERS = ['metro_common.h', os.path.join('..', '..', 'include', 'metro_hip.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-x', 'hip', '-Wall',
         '-Wno-unused-function'] + os.environ.get('METRO_EXTRA_HIPCC_FLAGS', '').split()


def _hipcc() -> str:
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', shutil.which('hipcc')):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found (looked at $HIPCC, /opt/rocm/bin/hipcc, PATH)')


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    hipcc = _hipcc()
    objdir = os.path.join(HERE, 'build')
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    xhdrs = hdrs + [os.path.join(CSRC, h) for h in EXPERIMENTAL_HEADERS]
    jobs = []

    def objects(sources, deps):
        objs = []
        for src in sources:
            s = os.path.join(CSRC, src)
            o = os.path.join(objdir, os.path.splitext(os.path.basename(src))[0] + '.o')
            objs.append(o)
            if force or _stale(o, [s] + deps):
                jobs.append([hipcc] + FLAGS + ['-c', s, '-o', o])
        return objs

    objs = objects(SOURCES, hdrs)
    n_product_jobs = len(jobs)
    # the experimental kernels and the bench yardstick are built best-effort: a problem there (or a tree shipped without
    # tools/ or csrc/experimental/) must not fail the build of the product library
    have_x = all(os.path.exists(os.path.join(CSRC, f)) for f in EXPERIMENTAL_SOURCES + EXPERIMENTAL_HEADERS)
    xobjs = objects(EXPERIMENTAL_SOURCES, xhdrs) if have_x else []
    x_jobs = jobs[n_product_jobs:]
    del jobs[n_product_jobs:]

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed ({r.returncode}): {" ".join(cmd)}\n{r.stdout}\n{r.stderr}')
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    def best_effort(what, fn):
        try:
            fn()
        except (RuntimeError, OSError) as e:
            print(f'metro_pose3d_amd.build: {what} not built (the product library is unaffected): {e}', file=sys.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
        x_done = list(ex.map(lambda c: best_effort('an experimental kernel', lambda: run(c)), x_jobs))
    del x_done
    if force or _stale(LIB_PATH, objs):
        run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB_PATH] + objs)
    if have_x and all(os.path.exists(o) for o in xobjs) and (force or _stale(EXPERIMENTAL_LIB_PATH, xobjs + [LIB_PATH])):
        # resolves set_error / note_kernel / validate_conv_desc / conv_gemm4w_shape_ok from the product library next to it
        best_effort('libmetro_experimental.so', lambda: run(
            [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', EXPERIMENTAL_LIB_PATH] + xobjs +
            ['-L' + HERE, '-l:libmetro_hip.so', '-Wl,-rpath,$ORIGIN']))
    if os.path.exists(PROBE_SRC) and (force or _stale(PROBE_LIB_PATH, [PROBE_SRC])):
        best_effort('tools/libmetro_probe.so', lambda: run(
            [hipcc, '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', PROBE_SRC, '-o', PROBE_LIB_PATH]))
    return LIB_PATH


if __name__ == '__main__':
    print(build_library(force='--force' in sys.argv, verbose=True))
