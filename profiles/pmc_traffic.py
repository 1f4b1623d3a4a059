#!/usr/bin/env python3
"""Per-layer PMC table (profiles/pmc_table.py) -> HBM traffic summary JSON read by bench.py (`roofline.traffic`).

    python profiles/pmc_traffic.py profiles/r01d_pmc_layers.tsv r01d > profiles/r01d_pmc_traffic.json

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64 B per 128 B request for wide coalesced
reads, so bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(tsv, tag, batch=64, arch=50, stride=16, dataset='h36m'):
    from bench import kernels_sha16, workload_key
    from metro_pose3d_amd import ModelSpec, _lib
    from metro_pose3d_amd.engine import Engine
    infos = Engine(ModelSpec(arch, stride, dataset), None, 'f16', batch).layer_infos()
    algo = sum(li.algo_act_bytes_per_image * batch + li.algo_param_bytes for li in infos if li.kind == _lib.LAYER_CONV)
    rows = list(csv.DictReader(open(tsv), delimiter='\t'))
    conv = [r for r in rows if r['layer'].startswith(('conv1', 'block', 'logits'))]
    f = sum(float(r['FETCH_SIZE']) for r in conv)
    w = sum(float(r['WRITE_SIZE']) for r in conv)
    fa = sum(float(r['FETCH_SIZE']) for r in rows)
    wa = sum(float(r['WRITE_SIZE']) for r in rows)
    print(json.dumps({
        'source': f'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python bench.py --arch {arch} --stride {stride} '
                  f'--dataset {dataset} --batch {batch} --steps 2 --warmup 1 --cpu-seconds 0 --no-extras`, second forward pass; '
                  f'see profiles/{tag}_pmc_layers.tsv',
        'correction': 'FETCH_SIZE counts 64 B per 128 B request for wide coalesced reads on gfx950 (MI355X_MICROARCH.md, '
                      'HBM section): bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024',
        'workload': workload_key(arch, stride, dataset, batch),
        'batch': batch,
        'kernels_sha16': kernels_sha16(),
        'conv_launches': len(conv),
        'conv_fetch_size_kb_raw': f,
        'conv_write_size_kb': w,
        'conv_hbm_bytes_per_forward': (2 * f + w) * 1024,
        'conv_hbm_bytes_per_launch_avg': (2 * f + w) * 1024 / max(len(conv), 1),
        'all_kernels_hbm_bytes_per_forward': (2 * fa + wa) * 1024,
        'algorithmic_min_bytes_per_forward_this_launch_set': int(algo),
        'measured_over_algorithmic': round((2 * f + w) * 1024 / algo, 3),
    }, indent=1))


if __name__ == '__main__':
    a = sys.argv
    main(a[1], a[2], int(a[3]) if len(a) > 3 else 64, *((int(a[4]), int(a[5]), a[6]) if len(a) > 6 else ()))
