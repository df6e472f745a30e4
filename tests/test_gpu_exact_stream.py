"""The whole-batch fp32 pass in its two forms: exact_stream_kernel (operands from LDS, the default wherever rows and codes move
as whole 16-byte pieces) against exact_tiled_kernel (the register form, vqhip_set_tuning key 18 = 0) and against the C oracle.

Bar: bit-identical indices, minima, distance matrices and column indices — both forms evaluate the same k-ordered fma chains
(/root/reference/vq/algorithms/vq/distances.py:28-46 is what they restate: cdist / 1 - normalised dot, then argmin)."""
import numpy as np
import pytest
import torch

from oracle import c_oracle as co, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from vector_quantization_amd import ops as _ops
    return _ops


@pytest.fixture()
def forms():
    from vector_quantization_amd import _lib
    L = _lib.lib()

    def both(fn):
        out = []
        try:
            for stream in (1, 0):
                assert L.vqhip_set_tuning(18, stream) == 0
                out.append(fn())
        finally:
            L.vqhip_set_tuning(18, 1)
        return out
    return both


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t.to(dtype) if dtype is not None else t


def same(a, b):
    a, b = a.cpu().numpy(), b.cpu().numpy()
    if a.dtype.kind == 'f':
        a, b = a.view(np.uint32), b.view(np.uint32)
    np.testing.assert_array_equal(a, b)


SHAPES = [  # N, K, D, rows dtype: row blocks and chunks that are not whole, tail blocks of D, one block, many blocks
    (1000, 700, 256, torch.float32), (4100, 1000, 72, torch.bfloat16), (300, 70000, 20, torch.float32),
    (513, 257, 36, torch.float32), (129, 300, 1280, torch.float32), (2500, 260, 8, torch.bfloat16),
    (127, 31, 4, torch.float32), (128, 256, 32, torch.bfloat16), (3072, 16384, 256, torch.bfloat16), (640, 9000, 100, torch.float32),
]


@pytest.mark.parametrize('metric', ['L2', 'Cosine'])
@pytest.mark.parametrize('N,K,D,dtype', SHAPES, ids=[f'{n}x{k}x{d}{"b" if t == torch.bfloat16 else "f"}' for n, k, d, t in SHAPES])
def test_streamed_form_equals_register_form_and_oracle(ops, forms, N, K, D, dtype, metric):
    x, w = synth.make_inputs('normal', 4000 + D, N, K, D)
    w[K // 2] = w[K // 3]                                    # an exact tie: the lower index wins in both forms
    xd, wd = dev(x, dtype), dev(w)
    if metric == 'Cosine':
        xd, wd = ops.normalize_rows(xd), ops.normalize_rows(wd)
    (i1, m1), (i0, m0) = forms(lambda: ops.argmin_exact(xd, wd, metric, return_min=True))
    same(i1, i0)
    same(m1, m0)
    if metric == 'L2':                                       # (the oracle on the rows as the kernel sees them: bf16 values widened)
        np.testing.assert_array_equal(i1.cpu().numpy(), co.l2_argmin(xd.float().cpu().numpy(), w))
    elif dtype == torch.float32:                             # (the oracle normalises for itself)
        np.testing.assert_array_equal(i1.cpu().numpy(), co.cos_argmin(x, w))


@pytest.mark.parametrize('metric', ['L2', 'Cosine'])
@pytest.mark.parametrize('N,K,D', [(200, 333, 32), (515, 258, 36), (130, 1000, 260)])
def test_distance_matrix_and_column_fallback_in_both_forms(ops, forms, N, K, D, metric):
    x, w = synth.make_inputs('normal', 91, N, K, D)
    xd, wd = dev(x), dev(w)
    if metric == 'Cosine':
        xd, wd = ops.normalize_rows(xd), ops.normalize_rows(wd)
    d1, d0 = forms(lambda: ops.distance(xd, wd, metric))
    same(d1, d0)
    d_ref = co.l2_dist(x, w) if metric == 'L2' else None
    if d_ref is not None:
        np.testing.assert_array_equal(d1.cpu().numpy(), d_ref)
    c1, c0 = forms(lambda: ops.col_argmin(xd, wd, metric))   # D % 8 != 0 takes the fp32 column pass (keys per code, atomics per item)
    same(c1, c0)
    np.testing.assert_array_equal(c1.cpu().numpy(), np.argmin(d1.cpu().numpy(), axis=0))


@pytest.mark.parametrize('N,K,D,dtype', [(700, 520, 64, torch.bfloat16), (257, 4096, 32, torch.float32), (1000, 260, 96, torch.bfloat16),
                                         (129, 8, 16, torch.float32), (4100, 1004, 72, torch.float32)])
def test_distance_rows_staged_through_lds(ops, forms, N, K, D, dtype):
    """K % 4 == 0: the streamed form writes d[N, K] as 16-byte row segments through a wave-private LDS tile (32 codes wide for fp32
    rows, 16 for bf16 rows); chunks and row blocks that are not whole, K smaller than a tile."""
    x, w = synth.make_inputs('normal', 333 + K, N, K, D)
    xd, wd = dev(x, dtype), dev(w)
    for metric in ('L2', 'Cosine'):
        xm, wm = (ops.normalize_rows(xd), ops.normalize_rows(wd)) if metric == 'Cosine' else (xd, wd)
        d1, d0 = forms(lambda: ops.distance(xm, wm, metric))
        same(d1, d0)
        if metric == 'L2':
            np.testing.assert_array_equal(d1.cpu().numpy(), co.l2_dist(xd.float().cpu().numpy(), w))


def test_nonfinite_rows_and_codes_in_both_forms(ops, forms):
    x, w = synth.make_inputs('normal', 5, 700, 900, 64)
    x[3, 5] = np.nan; x[10, :] = np.inf; x[11, 0] = -np.inf; x[200, 63] = 3e38
    w[7, 1] = np.inf; w[400, :] = np.nan; w[899, 2] = -3e38
    xd, wd = dev(x), dev(w)
    (i1, m1), (i0, m0) = forms(lambda: ops.argmin_exact(xd, wd, 'L2', return_min=True))
    same(i1, i0)
    same(m1, m0)
    np.testing.assert_array_equal(i1.cpu().numpy(), co.l2_argmin(x, w))


def test_histogram_and_repeat_calls_are_deterministic(ops):
    x, w = synth.make_inputs('normal', 17, 5000, 2048, 128)
    xd, wd = dev(x, torch.bfloat16), dev(w)
    hist = torch.zeros(2048, dtype=torch.int32, device='cuda')
    i1 = ops.argmin_exact(xd, wd, 'L2', hist=hist)
    i2 = ops.argmin_exact(xd, wd, 'L2')
    same(i1, i2)
    np.testing.assert_array_equal(hist.cpu().numpy(), np.bincount(i1.cpu().numpy(), minlength=2048))


_SHARE_CHILD = r'''
import sys, torch
sys.path.insert(0, sys.argv[1])
from vector_quantization_amd import ops, _lib
g = torch.Generator(device='cuda').manual_seed(int(sys.argv[2]))
x = torch.randn(4096, 256, device='cuda', generator=g).bfloat16()
w = torch.randn(16384, 256, device='cuda', generator=g)
L = _lib.lib()
L.vqhip_set_tuning(18, 0)
ref = ops.argmin_exact(x, w, 'L2')
L.vqhip_set_tuning(18, 1)
bad = sum(int((ops.argmin_exact(x, w, 'L2') != ref).sum().item() > 0) for _ in range(25))
print('BAD', bad)
'''


def test_streamed_form_with_eight_processes_on_the_gpu():
    """The tiles of a stage must have LANDED at the barrier that publishes them, however late the memory system delivers them:
    eight processes at once on the one GPU stretch the LDS-DMA round trips far enough to show a barrier without its vmcnt drain
    (round 6: 12 of 32 such processes saw 1-2 % of the rows decided on the previous stage's tiles; vq_dma_barrier)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = [subprocess.Popen([sys.executable, '-c', _SHARE_CHILD, root, str(100 + i)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for i in range(8)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (out, err) in zip(procs, outs):
        assert p.returncode == 0, err[-2000:]
        assert 'BAD 0' in out, out
