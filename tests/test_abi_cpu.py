"""CPU-side checks of the C-ABI boundary: the library loads and exports every symbol include/vqhip.h
declares (no compute calls without a GPU), and the size functions are consistent."""
import os
import re

import pytest

from vector_quantization_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.lib()


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'vqhip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(vqhip_[a-z0-9_]+)\s*\(', text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    names = declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/vqhip.h but not exported by libvqhip.so'
        assert n in _lib.SIGNATURES, f'{n} has no ctypes signature'
    assert sorted(_lib.SIGNATURES) == names


def test_version_and_sizes(lib):
    assert lib.vqhip_version() == _lib.ABI_VERSION == 300
    cb = lib.vqhip_codebook_bytes(16384, 256)
    # fp16 fragment image (K*D*2) + fp32 normalised copy (K*D*4) + norms + aux chunks
    assert cb >= 16384 * 256 * 6 and cb < 16384 * 256 * 7
    ws = lib.vqhip_workspace_bytes(65536, 16384, 256)
    assert ws >= 65536 * 256 * 2
    assert lib.vqhip_workspace_bytes(0, 16384, 256) > 0
    assert lib.vqhip_codebook_bytes(0, 256) == 0


def test_bad_arguments_are_rejected_without_a_gpu(lib):
    # argument validation happens before any HIP call
    rc = lib.vqhip_argmin(None, 0, None, None, 10, 10, 8, 0, None, None, None, None)
    assert rc == -22
    assert b'vqhip_argmin' in lib.vqhip_last_error()
    rc = lib.vqhip_codebook_prepare(None, 10, 8, 0, None, None)
    assert rc == -22


def test_ops_refuse_cpu_tensors():
    import torch
    from vector_quantization_amd import ops
    with pytest.raises(_lib.VqhipError):
        ops.prepare_codebook(torch.zeros(8, 8), 'L2')
