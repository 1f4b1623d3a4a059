#!/usr/bin/env python3
"""Batch-independent cost of every launch of the forward: the `P` of a `T = P + n * t` fit.

    python tools/fixed_cost_table.py <layers_b64.tsv> <layers_b256.tsv> [<layers_b128.tsv>] [--event-us 6.0]

Inputs are `bench.py --batch N --layer-report FILE` tables (per-layer HIP-event pairs, mean of 5 passes) or the rocprofv3
per-layer tables of profiles/collect.sh (`*_pmc_layers.tsv`: kernel durations, no event pair: event_us = 0).  Per layer:
    t * 64 = (T256 - T64) / 3            the per-batch-64 slope between the two batches
    P      = T64 - t * 64 - event_us     what does not scale with the batch (event_us = what an event pair reads around a
                                         one-block kernel: the softargmax_finalize row, ~6 us; it is in every row)
The fit only means something where both batches run the SAME kernel instantiation (metro_plan_layer_kernel, a dry run: no GPU
needed); rows where the dispatch changes between the batches are marked `*` and their P is the cost of the batch-64 form over a
line through the batch-256 form -- an upper bound, not a prologue.  A third table (batch 128) is used as a linearity check.
Prints the table, sum P over the same-kernel rows and over all rows, and the three largest."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def read(path):
    rows = {}
    order = []
    batch = None
    col, scale = 2, 1e3                      # bench.py --layer-report: column `ms`
    for line in open(path):
        if line.startswith('#'):
            for tok in line.replace(',', ' ').split():
                if tok.isdigit() and batch is None:
                    batch = int(tok)
            continue
        f = line.rstrip('\n').split('\t')
        if f[0] == 'layer':
            if 'us' in f:                    # profiles/pmc_table.py: rocprofv3 kernel durations, column `us` (no event pair in them)
                col, scale = f.index('us'), 1.0
            continue
        if f[0] == 'TOTAL' or len(f) < 3:
            continue
        rows[f[0]] = float(f[col]) * scale   # us
        order.append(f[0])
    if batch is None:                        # pmc tables carry no header line: the batch is in the file name as collect.sh writes it
        batch = 256 if '_b256_' in os.path.basename(path) else 128 if '_b128_' in os.path.basename(path) else 64
    return batch, order, rows, scale == 1.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('b64')
    ap.add_argument('b256')
    ap.add_argument('b128', nargs='?')
    ap.add_argument('--event-us', type=float, default=None, help='default: the softargmax row of the batch-64 table')
    a = ap.parse_args()
    n1, order, t1, rocprof = read(a.b64)
    n2, _, t2, _ = read(a.b256)
    t3 = read(a.b128)[2] if a.b128 else None
    n3 = read(a.b128)[0] if a.b128 else None
    from metro_pose3d_amd import ModelSpec
    from metro_pose3d_amd.engine import Engine
    eng = Engine(ModelSpec(50, 16, 'h36m'), None, 'f16', max(n1, n2))
    names = [li.name.decode() for li in eng.layer_infos()]
    k1 = dict(zip(names, eng.layer_kernels(n1)))
    k2 = dict(zip(names, eng.layer_kernels(n2)))
    ev = a.event_us if a.event_us is not None else (0.0 if rocprof else t1.get('softargmax', 6.0))
    ratio = n2 / n1 - 1.0
    print(f'# T = P + n*t from batch {n1} and batch {n2}; event-pair floor {ev:.1f} us subtracted from every P')
    print('layer\tT%d_us\tT%d_us\tslope_us_per_%d\tP_us\tsame_kernel\tkernel_at_%d%s' % (n1, n2, n1, n1, '\tlin_err_us_b%d' % n3 if t3 else ''))
    tot_same = tot_all = 0.0
    recs = []
    for name in order:
        if name not in t2:
            continue
        slope = (t2[name] - t1[name]) / ratio
        p = t1[name] - slope - ev
        same = k1.get(name) == k2.get(name)
        tot_all += p
        if same:
            tot_same += p
        recs.append((p, name, same))
        extra = ''
        if t3 and name in t3:
            pred = t1[name] + slope * (n3 / n1 - 1.0)
            extra = f'\t{t3[name] - pred:+.1f}'
        print(f'{name}\t{t1[name]:.1f}\t{t2[name]:.1f}\t{slope:.1f}\t{p:.1f}\t{"yes" if same else "*"}\t{k1.get(name, "?")}{extra}')
    print(f'# sum P: {tot_same:.0f} us over the {sum(1 for r in recs if r[2])} launches whose kernel is the same at both batches, '
          f'{tot_all:.0f} us over all {len(recs)} (rows marked * change kernel: upper bounds)')
    top = sorted(recs, reverse=True)[:5]
    print('# largest: ' + '; '.join(f'{n} {p:.1f} us{"" if s else " (*)"}' for p, n, s in top))


if __name__ == '__main__':
    main()
