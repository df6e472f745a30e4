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
    assert lib.vqhip_version() == _lib.ABI_VERSION == 600
    cb = lib.vqhip_codebook_bytes(16384, 256)
    # fp16 fragment image (K*D*2) + fp32 normalised copy (K*D*4) + norms + aux chunks
    assert cb >= 16384 * 256 * 6 and cb < 16384 * 256 * 7
    ws = lib.vqhip_workspace_bytes(65536, 16384, 256)
    assert ws >= 65536 * 256 * 2
    assert lib.vqhip_workspace_bytes(0, 16384, 256) > 0
    assert lib.vqhip_codebook_bytes(0, 256) == 0


def test_bad_arguments_are_rejected_without_a_gpu(lib):
    # argument validation happens before any HIP call
    rc = lib.vqhip_argmin(None, 0, None, None, 0, 10, 10, 8, 0, None, None, None, 0, None)
    assert rc == -22
    assert b'vqhip_argmin' in lib.vqhip_last_error()
    rc = lib.vqhip_codebook_prepare(None, 10, 8, 0, None, 0, None)
    assert rc == -22


def test_tuning_knobs_are_host_state_and_unknown_keys_are_refused(lib):
    """vqhip_set_tuning touches no device: every documented A/B key is accepted (and restored), any other key is VQHIP_EINVAL;
    VQHIP_TUNING applies the same knobs at load (vector_quantization_amd/_lib.py)."""
    for key, default in ((2, 0), (5, 1), (6, 2), (8, 1), (9, 1), (10, 1), (11, 1), (13, 1), (15, 1), (17, 1), (18, 1), (12, 0)):
        assert lib.vqhip_set_tuning(key, 0) == 0
        assert lib.vqhip_set_tuning(key, default) == 0
    for key in (-1, 1, 7, 14, 16, 19, 1000):
        assert lib.vqhip_set_tuning(key, 1) == -22
        assert b'vqhip_set_tuning' in lib.vqhip_last_error()


def test_undersized_buffers_are_refused_before_any_launch(lib):
    """Every scratch buffer travels with its size: one byte less than the matching *_bytes function asks for is VQHIP_EINVAL,
    with both numbers in the message — never a kernel writing past the end of a caller's allocation.  (Pointers here are
    fake but non-null: the size check sits in front of the first HIP call.)"""
    import ctypes
    fake = ctypes.c_void_p(0x1000)
    N, K, D = 1000, 512, 64
    cb, ws = lib.vqhip_codebook_bytes(K, D), lib.vqhip_workspace_bytes(N, K, D)
    assert lib.vqhip_codebook_prepare(fake, K, D, 0, fake, cb - 1, None) == -22
    assert b'cb too small' in lib.vqhip_last_error() and str(cb).encode() in lib.vqhip_last_error()
    assert lib.vqhip_argmin(fake, 0, fake, fake, cb, N, K, D, 0, fake, None, fake, ws - 1, None) == -22
    assert b'ws too small' in lib.vqhip_last_error() and str(ws).encode() in lib.vqhip_last_error()
    assert lib.vqhip_argmin(fake, 0, fake, fake, cb - 1, N, K, D, 0, fake, None, fake, ws, None) == -22
    assert lib.vqhip_encode_ex(fake, 0, fake, N, K, D, 0, fake, cb, fake, None, None, fake, ws - 1, 0, None) == -22
    assert lib.vqhip_encode_ex(fake, 0, fake, N, K, D, 0, fake, cb - 1, fake, None, None, fake, ws, 0, None) == -22
    assert lib.vqhip_argmin_exact(fake, 0, fake, N, K, D, 0, fake, None, None, fake, ws - 1, None) == -22
    assert lib.vqhip_distance(fake, 0, fake, N, K, D, 0, fake, fake, ws - 1, None) == -22
    cws = lib.vqhip_col_workspace_bytes(N, K, D)
    assert cws > ws
    assert lib.vqhip_col_argmin(fake, 0, fake, N, K, D, 0, fake, fake, ws, None) == -22       # the row-pass size is NOT enough
    rws = lib.vqhip_col_rows_workspace_bytes(N, 100, D)
    assert 0 < rws < cws
    assert lib.vqhip_col_argmin_rows(fake, 0, fake, fake, fake, 100, N, K, D, 0, fake, fake, rws - 1, None) == -22
    ows = lib.vqhip_order_workspace_bytes(N, K)
    assert lib.vqhip_token_order(fake, N, K, fake, fake, fake, fake, ows - 1, None) == -22
    sws = lib.vqhip_segsum_workspace_bytes(N, D)
    assert lib.vqhip_segsum_rows(fake, fake, fake, fake, N, K, D, fake, fake, sws - 1, None) == -22
    assert lib.vqhip_vq_backward_w_ordered(fake, 0, fake, fake, fake, fake, N, K, D, None, fake, fake, sws - 1, None) == -22
    # limits
    assert lib.vqhip_token_order(fake, N, 40000, fake, fake, fake, fake, 1 << 40, None) == -22 and b'32768' in lib.vqhip_last_error()
    assert lib.vqhip_argmin(fake, 0, fake, fake, 1 << 40, 1 << 31, K, D, 0, fake, None, fake, 1 << 40, None) == -22


def test_ops_refuse_cpu_tensors():
    import torch
    from vector_quantization_amd import ops
    with pytest.raises(_lib.VqhipError):
        ops.prepare_codebook(torch.zeros(8, 8), 'L2')


def test_one_call_forwards_validate_their_argument_blocks(lib):
    """vqhip_cvq_forward / vqhip_vqkd_forward take a struct of pointers and sizes: a block of another size, a missing
    pointer, an undersized workspace or exchange buffer are VQHIP_EINVAL before any launch."""
    import ctypes
    N, K, D = 1000, 512, 64
    a = _lib.CvqForwardArgs()
    assert lib.vqhip_cvq_forward(ctypes.byref(a), None) == -22 and b'struct_bytes' in lib.vqhip_last_error()
    a.struct_bytes = ctypes.sizeof(_lib.CvqForwardArgs)
    a.N, a.K, a.D, a.phases, a.world = N, K, D, _lib.STEP_ALL, 1
    assert lib.vqhip_cvq_forward(ctypes.byref(a), None) == -22 and b'null pointer' in lib.vqhip_last_error()
    for f in ('x', 'w_in', 'p_in', 'w_out', 'p_out', 'rows', 'slot', 'count', 'cb', 'idx', 'hist', 'ws'):
        setattr(a, f, 0x1000)
    need = lib.vqhip_cvq_forward_ws_bytes(N, K, D, K)
    assert need > lib.vqhip_workspace_bytes(N, K, D) + lib.vqhip_col_rows_workspace_bytes(N, K, D)
    a.ws_bytes = lib.vqhip_workspace_bytes(N, K, D) - 1
    assert lib.vqhip_cvq_forward(ctypes.byref(a), None) == -22 and b'ws too small' in lib.vqhip_last_error()
    a.ws_bytes, a.exchange, a.world = need, 1, 2
    assert lib.vqhip_cvq_forward(ctypes.byref(a), None) == -22 and b'packed' in lib.vqhip_last_error()
    a.packed = 0x1000
    assert lib.vqhip_cvq_forward(ctypes.byref(a), None) == -22 and b'communicator' in lib.vqhip_last_error()
    # NearestAnchor(sync=True): the key exchange needs its buffer, this rank's number and rows that fit 24 bits
    a.anchor_sync, a.rank, a.comm = 1, 0, 0x1000
    assert lib.vqhip_cvq_forward(ctypes.byref(a), None) == -22 and b'keys' in lib.vqhip_last_error()
    a.keys, a.rank = 0x1000, 2
    assert lib.vqhip_cvq_forward(ctypes.byref(a), None) == -22 and b'rank' in lib.vqhip_last_error()
    a.rank, a.phases = 1, 5
    assert lib.vqhip_cvq_forward(ctypes.byref(a), None) == -22 and b'phases' in lib.vqhip_last_error()
    a.phases, a.anchor_sync = _lib.STEP_PACK_SYNC, 0
    assert lib.vqhip_cvq_forward(ctypes.byref(a), None) == -22 and b'phases' in lib.vqhip_last_error()      # PACK_SYNC is a sync phase
    a.phases = _lib.STEP_ALL
    fake = ctypes.c_void_p(0x1000)
    assert lib.vqhip_cvq_col_keys(fake, 0, fake, fake, fake, 8, fake, (1 << 24) + 1, K, D, 0, 0, fake, None) == -22 and b'24 bits' in lib.vqhip_last_error()
    assert lib.vqhip_cvq_col_keys(fake, 0, fake, fake, fake, 8, fake, N, K, D, 0, 256, fake, None) == -22
    assert lib.vqhip_cvq_col_keys(fake, 0, fake, fake, fake, K + 1, fake, N, K, D, 0, 0, fake, None) == -22
    assert lib.vqhip_cvq_pack_sync(fake, N, fake, 0, None, fake, 8, 0, K, D, fake, None) == -22
    assert lib.vqhip_allreduce_min_i64(None, 8, fake, None) == -22
    assert lib.vqhip_cvq_apply(fake, fake, fake, fake, fake, N, fake, 0, fake, None, 1, fake, K + 1, K, D, 0.99, 1e-3, None) == -22
    a.D = 12
    assert lib.vqhip_cvq_forward(ctypes.byref(a), None) == -22
    b = _lib.VqkdForwardArgs()
    assert lib.vqhip_vqkd_forward(ctypes.byref(b), None) == -22 and b'struct_bytes' in lib.vqhip_last_error()
    b.struct_bytes = ctypes.sizeof(_lib.VqkdForwardArgs)
    b.N, b.K, b.D, b.phases, b.world, b.metric = N, K, D, _lib.STEP_ALL, 1, 0
    assert lib.vqhip_vqkd_forward(ctypes.byref(b), None) == -22 and b'cosine' in lib.vqhip_last_error()
    b.metric = 1
    for f in ('x', 'w_in', 'w_mid', 'w_out', 'xn', 'xq', 'cb', 'idx', 'hist', 'packed', 'ws'):
        setattr(b, f, 0x1000)
    b.packed_floats = lib.vqhip_pack_floats(K, K, D) - 1
    assert lib.vqhip_vqkd_forward(ctypes.byref(b), None) == -22 and b'packed' in lib.vqhip_last_error()
    b.packed_floats += 1
    b.ws_bytes = lib.vqhip_vqkd_forward_ws_bytes(N, K, D) - 1
    assert lib.vqhip_vqkd_forward(ctypes.byref(b), None) == -22 and b'ws too small' in lib.vqhip_last_error()
    assert lib.vqhip_vqkd_backward(None, 0, None, None, None, N, D, None, None, None, None) == -22
