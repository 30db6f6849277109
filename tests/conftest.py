import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, '3d-fm-gan_amd')
for p in (ROOT, PKG, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    import __graft_entry__
    __graft_entry__.ensure_built()      # fresh checkout: compile libfmgan_hip.so + the oracle once (hipcc needs no GPU)


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    class G:
        def __init__(self):
            self._c = {}

        def __call__(self, name):
            if name not in self._c:
                self._c[name] = np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz'))
            return self._c[name]

        def manifest(self, name):
            import json
            return json.load(open(os.path.join(ROOT, 'tests', 'golden', name + '_manifest.json')))

    return G()
