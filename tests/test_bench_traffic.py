"""bench.py attaches rocprofv3-derived HBM traffic only from a profile collected for the SAME kernel sources and batch."""
import json
import os

import bench


def _write(root, name, batch, sha, nbytes):
    os.makedirs(os.path.join(root, 'profiles'), exist_ok=True)
    with open(os.path.join(root, 'profiles', name), 'w') as f:
        json.dump({'batch': batch, 'kernels_sha16': sha, 'conv_launches': 49, 'conv_hbm_bytes_per_forward': nbytes,
                   'conv_hbm_bytes_per_launch_avg': nbytes / 49}, f)


def test_traffic_is_matched_by_batch_and_kernel_hash(tmp_path, monkeypatch):
    sha = bench.kernels_sha16()
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    monkeypatch.setattr(bench, 'kernels_sha16', lambda: sha)
    _write(str(tmp_path), 'x_pmc_traffic.json', 64, sha, 4.0e9)
    _write(str(tmp_path), 'x_b256_pmc_traffic.json', 256, sha, 16.0e9)
    r64, r256, r128 = {}, {}, {}
    bench.attach_traffic(r64, 64)
    bench.attach_traffic(r256, 256)
    bench.attach_traffic(r128, 128)
    assert r64['traffic'] == 4.0e9 and r256['traffic'] == 16.0e9 and r128 == {}


def test_stale_traffic_is_refused(tmp_path, monkeypatch):
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    _write(str(tmp_path), 'old_pmc_traffic.json', 64, 'deadbeefdeadbeef', 5.0e9)
    r = {}
    bench.attach_traffic(r, 64)
    assert 'traffic' not in r and 'stale' in r['traffic_note']


def test_matching_hash_wins_whatever_the_file_times(tmp_path, monkeypatch):
    """After a checkout file times are arbitrary: with several committed profiles of one workload the one collected for the
    current kernel sources is attached, also when a stale one is newer; other workloads are keyed by name."""
    import time
    sha = bench.kernels_sha16()
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    monkeypatch.setattr(bench, 'kernels_sha16', lambda: sha)
    _write(str(tmp_path), 'r03a_pmc_traffic.json', 64, sha, 4.0e9)
    time.sleep(0.02)
    _write(str(tmp_path), 'r02z_pmc_traffic.json', 64, 'deadbeefdeadbeef', 5.0e9)      # newer AND stale
    with open(tmp_path / 'profiles' / 'r03a_c4_pmc_traffic.json', 'w') as f:
        json.dump({'workload': bench.workload_key(101, 8, 'many19', 32), 'batch': 32, 'kernels_sha16': sha, 'conv_launches': 101,
                   'conv_hbm_bytes_per_forward': 9.0e9, 'conv_hbm_bytes_per_launch_avg': 9.0e9 / 101}, f)
    r, r4 = {}, {}
    bench.attach_traffic(r, 64)
    bench.attach_traffic(r4, bench.workload_key(101, 8, 'many19', 32))
    assert r['traffic'] == 4.0e9 and 'r03a_pmc_traffic.json' in r['traffic_note']
    assert r4['traffic'] == 9.0e9
