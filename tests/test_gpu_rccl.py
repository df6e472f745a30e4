"""RCCL really executes: a world-size-1 `nccl` process group on the one GPU of the box.

The child (tests/rccl_ws1_child.py) is started with `python -m torch.distributed.run --nproc-per-node=1` — a fresh process,
launched before any GPU call — and sends the CVQ-VAE and VQ-KD training steps through the multi-rank flow
(VQ_FORCE_EXCHANGE=1): pack -> ONE all-reduce on RCCL -> apply.  Reference call sites: vq/algorithms/vq/utils.py:26-35,
vqkd/quantizers/callbacks.py:63-64, cvqvae/anchors.py:60-67.  Asserted: codebooks, probabilities and tokens equal the
no-process-group result bit for bit — with the collective issued by torch.distributed and by libvqhip's own communicator on
the compute stream (vqhip_allreduce_packed) — one collective per step, and the same through HIP-graph replay with the
collective inside the capture."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@pytest.fixture(scope='module')
def ws1(tmp_path_factory):
    out = str(tmp_path_factory.mktemp('rccl') / 'ws1.json')
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.pop('VQHIP_ALLREDUCE', None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=1', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'tests', 'rccl_ws1_child.py'), '--out', out]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, f'child failed (rc={res.returncode})\n{res.stdout[-3000:]}\n{res.stderr[-3000:]}'
    return json.load(open(out))


def test_world_size_one_group_runs_on_rccl(ws1):
    assert ws1['backend'] == 'nccl' and ws1['world'] == 1 and ws1['rccl_ranks'] == 1


@pytest.mark.parametrize('kind', ['cvq', 'vqkd', 'cvqsync'])
@pytest.mark.parametrize('route', ['torch', 'direct'])
def test_forced_exchange_equals_the_one_rank_flow(ws1, kind, route):
    assert ws1[f'{kind}_{route}_bit_identical'] is True
    # ONE packed all-reduce per training step; NearestAnchor(sync=True) adds the MIN all-reduce of the keys in front of it
    # (direct route: vqhip_allreduce_min_i64 — int64 MIN on the library's own communicator, on RCCL)
    assert ws1[f'{kind}_{route}_collectives_per_step'] == (2.0 if kind == 'cvqsync' else 1.0)
    if route == 'direct':
        st = ws1['status_direct']
        assert st['direct'] is True and st['error'] is None, st         # libvqhip's own communicator was used


@pytest.mark.parametrize('kind', ['cvq', 'vqkd', 'cvqsync'])
def test_graph_replay_with_the_collective_captured(ws1, kind):
    # the direct route puts the collective on the captured stream itself; the torch route is reported alongside
    assert ws1[f'{kind}_direct_graphed_error'] is None, ws1[f'{kind}_direct_graphed_error']
    assert ws1[f'{kind}_direct_graphed_bit_identical'] is True


@pytest.mark.parametrize('kind', ['cvq', 'vqkd'])
@pytest.mark.parametrize('route', ['torch', 'direct'])
def test_quantizer_inside_ddp_on_the_nccl_backend(ws1, kind, route):
    """DistributedDataParallel on the nccl backend with the exchange forced: torch's communicator (DDP's bucket all-reduces)
    and — on the direct route — the library's own alive in one process, under bf16 autocast; bit-identical to the bare module."""
    assert ws1[f'ddp_{kind}_{route}_bit_identical'] is True
    if route == 'direct':
        assert ws1['ddp_status_direct']['direct'] is True


def test_graphed_quantizer_inside_ddp(ws1):
    assert ws1['ddp_graphed_error'] is None, ws1['ddp_graphed_error']
    assert ws1['ddp_graphed_tokens_identical'] is True and ws1['ddp_graphed_max_param_diff'] <= 1e-6


def test_fsdp_use_orig_params_smoke(ws1):
    """FullyShardedDataParallel(use_orig_params=True) around the toy model: construction and two optimizer steps, the second
    of which only sees the first one's codebook update if it landed in FSDP's flat parameter."""
    assert ws1['fsdp_error'] is None, ws1['fsdp_error']
    assert ws1['fsdp_tokens_identical'] is True and ws1['fsdp_loss_diff'] <= 1e-5


def test_bench_multi_rank_blocks_on_rccl_at_world_size_one(tmp_path):
    """The part of `bench.py --gpus N` that only runs with more than one rank — each rank's own clock and kernel time gathered, parity
    gathered, the communicating cvq blocks, and the direct-route sub-block in a child process with its own rendezvous — on the REAL
    backend: one rank under `torch.distributed.run` with VQ_BENCH_FORCE_CVQ=1 and VQ_FORCE_EXCHANGE=1.  The child creates the library's
    own RCCL communicator (VQHIP_ALLREDUCE=direct) on the GPU whose parent process holds torch's: what the first multi-GPU lease will
    do on every rank."""
    env = dict(os.environ, VQ_BENCH_FORCE_CVQ='1', VQ_FORCE_EXCHANGE='1', VQ_BENCH_CVQ_SETTLE='30')
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.pop('VQHIP_ALLREDUCE', None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=1', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1', '--images', '64',
           '--min-seconds', '0', '--no-cpu-baseline']
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, f'{res.stdout[-2000:]}\n{res.stderr[-3000:]}'
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{') and '"metric"' in ln]
    assert len(lines) == 1, res.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec['collective_backend'] == 'nccl' and rec['rccl_ranks'] == 1 and rec['parity']['ranks_checked'] == 1
    assert len(rec['per_rank_ms_per_step']) == 1 and len(rec['roofline']['kernel_ms_per_rank']) == 1
    direct = rec['cvq'].pop('direct_route')
    cluster = rec['cvq'].pop('cluster_sync_6272')
    assert cluster['codebook_in_sync'] is True and cluster['one_call_forward'] is True and cluster['collectives_per_step'] == 2.0, cluster
    for toks, blk in rec['cvq'].items():
        assert blk['codebook_in_sync'] is True and blk['one_call_forward'] is True and blk['collectives_per_step'] == 1.0, (toks, blk)
        assert blk['collective_ms'] is not None and blk['exchange_route']['mode'] in ('auto', 'torch')
    assert direct['ok'] is True and direct['route_requested'] == 'direct', direct
    assert direct['exchange_route']['direct'] is True and direct['exchange_route']['error'] is None, direct
    assert direct['codebook_in_sync'] is True and direct['collectives_per_step'] == 1.0 and direct['ms_per_step'] > 0
