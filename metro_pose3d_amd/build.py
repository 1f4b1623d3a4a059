"""Builds libmetro_hip.so (gfx950 only) in-tree with hipcc.  `python -m metro_pose3d_amd.build`."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB_PATH = os.path.join(HERE, 'libmetro_hip.so')
SOURCES = ['conv_igemm_f16_dma.hip', 'conv_gemm8p.hip', 'conv_gemm4w.hip', 'conv_gemm4d.hip', 'conv3x3_f16_slab.hip', 'conv3x3_c64.hip', 'conv_pw64.hip', 'head_f16.hip', 'stem_pool_f16.hip', 'conv_igemm_f64acc.hip', 'conv_igemm_f32.hip', 'pool_softargmax.hip', 'eval_metrics.hip', 'heads.hip', 'plan.cpp']
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
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, os.path.splitext(src)[0] + '.o')
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([hipcc] + FLAGS + ['-c', s, '-o', o])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed ({r.returncode}): {" ".join(cmd)}\n{r.stdout}\n{r.stderr}')
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB_PATH, objs):
        run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB_PATH] + objs)
    return LIB_PATH


if __name__ == '__main__':
    print(build_library(force='--force' in sys.argv, verbose=True))
