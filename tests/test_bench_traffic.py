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
