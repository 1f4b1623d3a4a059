import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')
    # the C-ABI library is built in-tree and git-ignored: build it when a fresh checkout lacks it
    # (hipcc cross-compiles gfx950 without a GPU); a stale library is rebuilt from newer sources
    from metro_pose3d_amd.build import build_library
    try:
        build_library()
    except Exception as e:  # noqa: BLE001  (no hipcc: tests that need the library will say so)
        print(f'[conftest] could not build libmetro_hip.so: {e}', file=sys.stderr)


@pytest.fixture(scope='session')
def lib():
    from metro_pose3d_amd import _lib
    return _lib.load()


@pytest.fixture(scope='session')
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail('a -m gpu test ran without a visible HIP device')
    return torch.device('cuda', 0)
