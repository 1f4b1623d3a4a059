"""profiles/roofline_floors.py (the per-launch floors made reproducible) runs without a GPU on the committed
counter tables and reads the workload -- the batch above all: FLOPs scale with it -- off the table's name."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(table, *extra):
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'profiles', 'roofline_floors.py'), table, *extra], capture_output=True, text=True,
                       cwd=ROOT, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [l.split('\t') for l in r.stdout.splitlines() if '\t' in l]
    total = [x for x in rows if x[0] == 'TOTAL'][0]
    return rows, total, r.stdout


def _latest(pattern):
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', pattern)))
    if not files:
        pytest.skip(f'no committed table matches profiles/{pattern}')
    return files[-1]


def test_floors_of_the_batch256_table_use_batch256_flops():
    table = _latest('r0*_b256_pmc_layers.tsv')
    rows, total, out = _run(table)
    # RN50-s16-J17: 15.299 GFLOP per crop x 256 crops (metro_plan_flops_per_image), whatever the launch set looks like
    assert abs(float(total[2]) - 15.299 * 256) < 2.0, total
    assert 'sum of per-launch floors' in out
    # columns: layer us GFLOP algo_MB counter_MB counter/algo mfma_floor hbm_floor floor floor/measured bound
    frac = float(total[9])
    assert 0.3 < frac < 1.0, total
    # the default takes the HBM floor from ALGORITHMIC bytes (measured bytes flatter the score) and the
    # last line prints both sums; --measured-bytes can only raise it (counter bytes >= algorithmic bytes, launch by launch)
    assert 'floors from ALGORITHMIC bytes' in out and '# both: with algorithmic bytes' in out
    assert float(total[4]) >= float(total[3]) > 0, total
    _, total_m, out_m = _run(table, '--measured-bytes')
    assert 'floors from MEASURED' in out_m and float(total_m[9]) >= frac
    # the measured ceilings can only raise the ratio
    _, total2, _ = _run(table, '--mfma-tflops', '1700', '--hbm-gbs', '4500')
    assert float(total2[9]) > frac


def test_floors_of_the_default_table_use_batch64_flops():
    tables = [t for t in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r0*_pmc_layers.tsv')))
              if not any(tag in os.path.basename(t) for tag in ('_b256_', '_c3_', '_c4_', '_c5_'))]
    if not tables:
        pytest.skip('no committed batch-64 table')
    _, total, _ = _run(tables[-1])
    assert abs(float(total[2]) - 15.299 * 64) < 1.0, total


@pytest.mark.parametrize('name', ['knockouts_r02_r04.patch', 'knockouts_r05.patch', 'knockouts_r06.patch'])
def test_knockout_patches_still_apply(name):
    """The timing knock-outs (#ifdef METRO_DBG_*) live OUTSIDE the product kernels, as patches (tools/build_dbg_variants.sh): they
    must keep applying to the sources they instrument, or the knock-out tables of NOTES_dead_ends.md stop being reproducible."""
    import shutil
    if shutil.which('patch') is None:
        pytest.skip('no patch(1) in this image')
    r = subprocess.run(['patch', '-p0', '--dry-run', '-i', os.path.join('tools', name)], capture_output=True, text=True, cwd=ROOT, timeout=60)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'METRO_DBG' not in ''.join(open(f).read() for f in glob.glob(os.path.join(ROOT, 'metro_pose3d_amd', 'csrc', '*.hip')))
