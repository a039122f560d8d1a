import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run on the GPU box with `pytest -m gpu`)')


def load_golden(name):
    path = os.path.join(GOLDEN, name)
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope='session')
def golden():
    return load_golden


def gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(autouse=True)
def _gpu_tests_need_a_gpu(request):
    # `-m gpu` tests are the parity tests proper; fail loudly rather than skip when they are selected
    # on a box without a GPU or without the built library.
    if request.node.get_closest_marker('gpu') is not None:
        import torch
        assert torch.cuda.is_available(), 'gpu-marked test selected but no GPU is visible'
        from sst_amd import _lib
        _lib.load()
    yield


# shared SST-base geometry (configs/sst_refactor/sst_waymoD5_1x_3class_8heads_v2.py:8-22)
VOXEL_SIZE = (0.32, 0.32, 6)
PC_RANGE = [-74.88, -74.88, -2, 74.88, 74.88, 4]
DROP_TRAIN = {
    0: {'max_tokens': 30, 'drop_range': (0, 30)},
    1: {'max_tokens': 60, 'drop_range': (30, 60)},
    2: {'max_tokens': 100, 'drop_range': (60, 100000)},
}
DROP_TEST = {
    0: {'max_tokens': 30, 'drop_range': (0, 30)},
    1: {'max_tokens': 60, 'drop_range': (30, 60)},
    2: {'max_tokens': 100, 'drop_range': (60, 100)},
    3: {'max_tokens': 144, 'drop_range': (100, 100000)},
}
