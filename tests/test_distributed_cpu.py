"""world_size-2 gloo tests (CPU) of the multi-rank codebook-update path: the packed statistics
all-reduce, QuantStatistics(sync=True), the is_sync invariant and the averaged-anchor exchange.  Compute on each rank
is done by the CPU oracle (there is no CPU product path); what is under test is the host-side exchange logic."""
import json
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import c_oracle as co, synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _run(fn, world=2):
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_entry, args=(fn, r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    errs = [r for r in results if isinstance(r, str)]
    assert not errs, errs
    return results


def _entry(fn, rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        torch.set_num_threads(1)
        dist.init_process_group('gloo', rank=rank, world_size=world)
        out = fn(rank, world)
        dist.barrier()
        dist.destroy_process_group()
        q.put(out)
    except Exception as e:  # pragma: no cover
        import traceback
        q.put('rank %d: %s\n%s' % (rank, e, traceback.format_exc()))


def _vqkd_rank(rank, world):
    from vector_quantization_amd.quantizers.statistics import QuantStatistics
    from vector_quantization_amd.utils import all_reduce_statistics, get_world_size, is_sync
    g = np.load(os.path.join(GOLDEN, 'update_vqkd.npz'))
    spec = json.loads(str(g['spec']))
    N, K, D = spec['N'], spec['K'], spec['D']
    x, w = synth.make_inputs('normal', spec['seed'], N, K, D)
    w = synth.unit_rows(w)
    assert get_world_size() == 2
    xn = co.normalize_rows(x)
    quant = g['quant'].astype(np.int64)
    xr, qr = co.normalize_rows(xn[rank::2]), quant[rank::2]             # this rank's shard (callbacks.py:124)
    hist = torch.from_numpy(co.bincount(qr, K))
    sums = torch.from_numpy(co.scatter_add_rows(xr, qr, K))
    # (1) QuantStatistics with a precomputed local histogram, sync=True: one packed collective
    st = QuantStatistics(quant=torch.from_numpy(qr), codebook_size=K, sync=True, hist=hist)
    np.testing.assert_array_equal(st.bin_count().numpy(), co.bincount(quant, K))
    assert int(st.num_elements()) == N
    np.testing.assert_allclose(st.frequency().numpy(), co.bincount(quant, K) / N)
    # (2) hist ‖ numel and the K×D sums: two collectives
    h2, n2, s2 = all_reduce_statistics(hist, qr.shape[0], sums)
    assert int(n2) == N
    np.testing.assert_array_equal(h2.numpy(), co.bincount(quant, K))
    e = co.kmeans_centroids(None, None, w, h2.numpy(), s2.numpy())
    e = co.normalize_rows(co.ema(w, co.normalize_rows(e), 0.99))
    # (3) every rank ends with the bit-identical codebook (the reference's DRY_RUN is_sync assert, update.py:54-55)
    assert is_sync(torch.from_numpy(e))
    assert not is_sync(torch.full((3,), float(rank)))
    np.testing.assert_allclose(e, g['w_new_2rank'], rtol=0, atol=3e-6)
    return e


def test_vqkd_two_rank_update_gloo():
    a, b = _run(_vqkd_rank)
    np.testing.assert_array_equal(a, b)


def _cvq_rank(rank, world):
    from vector_quantization_amd.quantizers.anchors import NearestAnchor
    from vector_quantization_amd.utils import all_reduce_statistics, is_sync
    g = np.load(os.path.join(GOLDEN, 'update_cvq_l2.npz'))
    spec = json.loads(str(g['spec']))
    N, K, D = spec['N'], spec['K'], spec['D']
    x, w = synth.make_inputs('normal', spec['seed'], N, K, D)
    w = synth.unit_rows(w)
    xr = x[rank::2]
    d = co.l2_dist(xr, w)
    quant = co.row_argmin(d)
    hist, numel, _ = all_reduce_statistics(torch.from_numpy(co.bincount(quant, K)), quant.shape[0])
    assert int(numel) == N
    p = co.ema(np.zeros(K, np.float32), (hist.numpy() / int(numel)).astype(np.float32), 0.99)
    np.testing.assert_allclose(p, g['p_2rank'], rtol=1e-6, atol=1e-9)

    class OracleNearest(NearestAnchor):
        """NearestAnchor with the column argmin / gather done by the oracle (CPU): BaseAnchor.forward's exchange
        (all_reduce then divide by world size, anchors.py:65-67) is what runs over gloo here."""

        def _anchors(self, x, e, d, quant, p, memo):
            idx = co.col_argmin(d.numpy())
            return torch.from_numpy(x.numpy()[idx]), memo

    anchors, _ = OracleNearest(sync=False)(torch.from_numpy(xr), torch.from_numpy(w), torch.from_numpy(d),
                                           torch.from_numpy(quant), torch.from_numpy(p))
    assert is_sync(anchors)
    decay = co.cvq_decay(p, K, 0.99, 1e-3)
    w_new = co.ema(w, anchors.numpy(), decay)
    np.testing.assert_allclose(w_new, g['w_new_2rank'], rtol=0, atol=3e-6)
    if rank == 0:
        np.testing.assert_array_equal(co.col_argmin(d), g['col_idx_rank0'].astype(np.int64))
    return w_new


def _cvq_sync_rank(rank, world):
    """NearestAnchor(sync=True) (configs/cluster/model.py:28): latents / matrix / quant are all-gathered, every rank
    computes the same global anchors (anchors.py:50-57) — no averaging."""
    from vector_quantization_amd.quantizers.anchors import NearestAnchor
    from vector_quantization_amd.utils import is_sync
    g = np.load(os.path.join(GOLDEN, 'update_cvq_l2.npz'))
    spec = json.loads(str(g['spec']))
    N, K, D = spec['N'], spec['K'], spec['D']
    x, w = synth.make_inputs('normal', spec['seed'], N, K, D)
    w = synth.unit_rows(w)
    half = N // 2
    xr = x[rank * half:(rank + 1) * half]          # contiguous shards: the gathered order equals the original order
    d = co.l2_dist(xr, w)

    class OracleNearest(NearestAnchor):
        def _anchors(self, x, e, d, quant, p, memo):
            idx = co.col_argmin(d.numpy())
            return torch.from_numpy(x.numpy()[idx]), memo

    anchors, _ = OracleNearest(sync=True)(torch.from_numpy(xr), torch.from_numpy(w), torch.from_numpy(d),
                                          torch.from_numpy(co.row_argmin(d)), torch.zeros(K))
    assert is_sync(anchors)
    np.testing.assert_array_equal(anchors.numpy(), x[g['col_idx'].astype(np.int64)])   # == single-process result
    return anchors.numpy()


def test_cvq_sync_anchor_gloo():
    a, b = _run(_cvq_sync_rank)
    np.testing.assert_array_equal(a, b)


def test_cvq_two_rank_update_gloo():
    a, b = _run(_cvq_rank)
    np.testing.assert_array_equal(a, b)


def test_single_process_helpers():
    from vector_quantization_amd.utils import EMA, PriorityQueue, all_reduce_statistics, ema, get_rank, get_world_size, is_sync
    assert get_world_size() == 1 and get_rank() == 0 and is_sync(torch.ones(3))
    h, n, s = all_reduce_statistics(torch.tensor([1, 2, 3]), 6, None)
    assert n == 6 and s is None and h.dtype == torch.int64
    e = EMA()
    assert e.decay == 0.99 and torch.equal(e(None, torch.ones(2)), torch.ones(2))
    np.testing.assert_allclose(ema(torch.ones(2), torch.zeros(2), 0.9).numpy(), 0.9)
    pq = PriorityQueue([dict(before_encode=1), dict(), dict(before_encode=-1)], ['a', 'b', 'c'])
    assert pq('before_encode') == ['c', 'b', 'a'] and pq('bind') == ['a', 'b', 'c']
