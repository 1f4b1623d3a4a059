#!/usr/bin/env python3
"""Per-launch roofline floors of a forward against the measured kernel times (made reproducible).

    python profiles/roofline_floors.py profiles/<tag>_pmc_layers.tsv [arch stride dataset batch] [--mfma-tflops 2500] [--hbm-gbs 6300]

For every launch of the plan:  floor = max(algorithmic FLOPs / MFMA peak, bytes / HBM rate), bytes = the launch's ALGORITHMIC
bytes (MetroLayerInfo.algo_*: every tensor the launch touches, once) -- the default since round 6: a floor taken from MEASURED
bytes rises with every wasted re-read and flatters the score.  The counter bytes of the table
((2 * FETCH_SIZE + WRITE_SIZE) * 1024, rocprofv3 --pmc, FETCH_SIZE doubled per MI355X_MICROARCH.md) are printed beside them
with their ratio, and `--measured-bytes` takes the floors from them instead; the last line always prints BOTH sums.  Prints the
table, the sums and Sigma floors / Sigma measured -- with the nominal peaks (2.5 PFLOP/s dense fp16, 6.3 TB/s achievable HBM) and, second line, with
the ceilings tools/peak_probe.hip measures on this chip (pass them: --mfma-tflops 1650 --hbm-gbs 5800).  No GPU needed: FLOPs and
algorithmic bytes come from a dry plan."""
import argparse
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('table')
    ap.add_argument('spec', nargs='*', default=None,
                    help='arch stride dataset batch of the table (default: read off its name as profiles/collect.sh writes them: '
                         '<tag>_pmc_layers.tsv = RN50-s16 h36m batch 64, _b256_ = batch 256, _c3_ / _c4_ / _c5_ = the shards)')
    ap.add_argument('--mfma-tflops', type=float, default=2500.0)
    ap.add_argument('--hbm-gbs', type=float, default=6300.0)
    ap.add_argument('--measured-bytes', action='store_true', help='floors from the counter bytes of the table instead of the algorithmic bytes')
    ap.add_argument('--algorithmic-bytes', action='store_true', help='(the default; kept for old command lines)')
    a = ap.parse_args()
    from metro_pose3d_amd import ModelSpec
    from metro_pose3d_amd.engine import Engine
    if not a.spec:        # FLOPs scale with the batch: a wrong default silently quarters the MFMA floors of a batch-256 table
        name = os.path.basename(a.table)
        a.spec = (['50', '16', 'h36m', '256'] if '_b256_' in name else ['50', '16', 'many19', '64'] if '_c3_' in name else
                  ['101', '8', 'many19', '32'] if '_c4_' in name else ['50', '4', 'h36m', '16'] if '_c5_' in name else ['50', '16', 'h36m', '64'])
    if len(a.spec) != 4:
        ap.error('spec = arch stride dataset batch')
    arch, stride, dataset, batch = int(a.spec[0]), int(a.spec[1]), a.spec[2], int(a.spec[3])
    infos = {li.name.decode(): li for li in Engine(ModelSpec(arch, stride, dataset), None, 'f16', batch).layer_infos()}
    rows = list(csv.DictReader(open(a.table), delimiter='\t'))
    tot = dict(us=0.0, floor=0.0, mf=0.0, hb=0.0, flops=0.0, bytes=0.0, algo=0.0, meas=0.0, floor_algo=0.0, floor_meas=0.0)
    print('layer\tus\tGFLOP\talgo_MB\tcounter_MB\tcounter/algo\tmfma_floor_us\thbm_floor_us\tfloor_us\tfloor/measured\tbound')
    for r in rows:
        name = r['layer']
        li = infos.get(name) or infos.get(name.replace('_fin', ''))
        us = float(r['us'])
        flops = li.flops_per_image * batch if li is not None else 0.0
        algo = (li.algo_act_bytes_per_image * batch + li.algo_param_bytes) if li is not None else 0.0
        meas = 0.0
        if r.get('FETCH_SIZE') and r.get('WRITE_SIZE'):
            meas = (2.0 * float(r['FETCH_SIZE']) + float(r['WRITE_SIZE'])) * 1024.0
        nbytes = meas if (a.measured_bytes and meas > 0) else algo
        mf = flops / (a.mfma_tflops * 1e12) * 1e6
        hb = nbytes / (a.hbm_gbs * 1e9) * 1e6
        fl = max(mf, hb)
        fl_algo = max(mf, algo / (a.hbm_gbs * 1e9) * 1e6)
        fl_meas = max(mf, (meas if meas > 0 else algo) / (a.hbm_gbs * 1e9) * 1e6)
        for k, v in (('us', us), ('floor', fl), ('mf', mf), ('hb', hb), ('flops', flops), ('bytes', nbytes), ('algo', algo),
                     ('meas', meas), ('floor_algo', fl_algo), ('floor_meas', fl_meas)):
            tot[k] += v
        print(f'{name}\t{us:.1f}\t{flops / 1e9:.1f}\t{algo / 1e6:.1f}\t{meas / 1e6:.1f}\t{(meas / algo if algo and meas else 0):.2f}\t{mf:.1f}\t{hb:.1f}\t{fl:.1f}\t'
              f'{fl / us if us else 0:.2f}\t{"mfma" if mf >= hb else "hbm"}')
    print(f'TOTAL\t{tot["us"]:.1f}\t{tot["flops"] / 1e9:.1f}\t{tot["algo"] / 1e6:.1f}\t{tot["meas"] / 1e6:.1f}\t'
          f'{(tot["meas"] / tot["algo"] if tot["algo"] and tot["meas"] else 0):.2f}\t{tot["mf"]:.1f}\t{tot["hb"]:.1f}\t{tot["floor"]:.1f}\t'
          f'{tot["floor"] / tot["us"]:.3f}\t-')
    print(f'# peaks: {a.mfma_tflops:.0f} TFLOP/s, {a.hbm_gbs:.0f} GB/s; floors from {"MEASURED (counter)" if a.measured_bytes else "ALGORITHMIC"} bytes: '
          f'sum of per-launch floors / sum of measured = {tot["floor"] / tot["us"]:.3f}; '
          f'whole-forward floors: MFMA {tot["mf"]:.0f} us, HBM {tot["hb"]:.0f} us (the launch set is '
          f'{"HBM" if tot["hb"] > tot["mf"] else "MFMA"}-bound in aggregate)')
    print(f'# both: with algorithmic bytes {tot["floor_algo"] / tot["us"]:.3f} (sum of floors {tot["floor_algo"]:.0f} us), '
          f'with counter bytes {tot["floor_meas"] / tot["us"]:.3f} ({tot["floor_meas"]:.0f} us); sum of measured {tot["us"]:.0f} us')


if __name__ == '__main__':
    main()
