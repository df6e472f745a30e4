import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # The built libraries are git-ignored: a fresh checkout builds them once (hipcc cross-compiles gfx950 without a GPU;
    # gcc builds the CPU oracle).  An existing build is used as is.
    from vector_quantization_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    from oracle import c_oracle
    c_oracle.build(force=False)


def _has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason='no GPU in this environment')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')
