import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a HIP device (MI355X); run with -m gpu on the GPU box')


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected explicitly with -m gpu; without a device they are skipped, never silently passed
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if not has_gpu:
        skip = pytest.mark.skip(reason='no HIP device in this environment')
        for it in items:
            if 'gpu' in it.keywords:
                it.add_marker(skip)
