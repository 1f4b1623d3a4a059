#!/usr/bin/env python3
"""Joins rocprofv3 --pmc counter_collection CSVs (one dir per pass) into a per-dispatch table for
one forward pass and labels the dispatches with the plan's layer names.

    python profiles/pmc_table.py gpurun_out/pmc1 [forward_index] [arch stride dataset batch]
"""
import csv
import glob
import os
import sys
from collections import OrderedDict, defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load(pass_dir):
    f = glob.glob(os.path.join(pass_dir, '*_counter_collection.csv'))[0]
    disp = OrderedDict()
    for r in csv.DictReader(open(f)):
        d = int(r['Dispatch_Id'])
        e = disp.setdefault(d, {'name': r['Kernel_Name'], 'grid': int(r['Grid_Size']), 'wg': int(r['Workgroup_Size']),
                                'lds': int(r['LDS_Block_Size']), 'vgpr': int(r['VGPR_Count']),
                                'dur': (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, 'c': {}})
        e['c'][r['Counter_Name']] = e['c'].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    return [v for v in disp.values() if 'metro' in v['name']]


def main():
    base = sys.argv[1]
    fwd = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    from metro_pose3d_amd import ModelSpec
    from metro_pose3d_amd.engine import Engine
    arch, stride, dataset, batch = (int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], int(sys.argv[6])) if len(sys.argv) > 6 else (50, 16, 'h36m', 64)
    infos = Engine(ModelSpec(arch, stride, dataset), None, 'f16', batch).layer_infos()
    names = []
    for li in infos:
        names.append(li.name.decode())
    per = None
    merged = defaultdict(dict)
    meta = {}
    for p in sorted(os.listdir(base)):
        d = os.path.join(base, p)
        if not os.path.isdir(d) or not glob.glob(os.path.join(d, '*_counter_collection.csv')):
            continue
        rows = load(d)
        if per is None:
            # launches per forward = dispatches / forwards (one stem launch per forward); the soft-argmax layer is one
            # launch (finalize) behind the fused head, two (partial + finalize) otherwise
            n_fwd = sum(1 for r in rows if 'stem_pool' in r['name'] or 'prep_input' in r['name'])
            per = len(rows) // max(n_fwd, 1)
            if per == len(names) + 1:
                names.append('softargmax_fin')
            assert per == len(names), (per, len(names))
        rows = rows[fwd * per:(fwd + 1) * per]
        for i, r in enumerate(rows):
            merged[i].update(r['c'])
            meta[i] = r
    cols = sorted({k for m in merged.values() for k in m})
    print('\t'.join(['layer', 'grid_wg', 'wg', 'lds', 'vgpr', 'us'] + cols))
    for i in range(per):
        if i not in meta:
            continue
        m = meta[i]
        print('\t'.join([names[i], str(m['grid'] // m['wg']), str(m['wg']), str(m['lds']), str(m['vgpr']), f"{m['dur']:.1f}"] +
                        [f"{merged[i].get(c, float('nan')):.0f}" for c in cols]))


if __name__ == '__main__':
    main()
