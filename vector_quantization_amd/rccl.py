"""The collective of the one exchange step, issued by libvqhip on the COMPUTE stream (include/vqhip.h: vqhip_allreduce_packed).

The reference all-reduces through ``torch.distributed`` (vq/algorithms/vq/utils.py:34-35, vqkd/quantizers/callbacks.py:63-64,
cvqvae/anchors.py:65-67): ProcessGroupNCCL runs the collective on its own stream, with an event hop from the compute stream
before it and another one back after it.  The packed exchange is one small latency-bound collective between two short
kernels (pack → all-reduce → apply), so here RCCL enqueues it on the very stream those kernels run on: an ``ncclComm`` of
this library's own, created once per process group —

    rank 0: vqhip_rccl_unique_id  →  torch.distributed's store  →  every rank: vqhip_rccl_comm_init

— with RCCL resolved at run time from the ``librccl.so`` PyTorch-ROCm has already mapped (never a second copy).  The step
stays capturable into a HIP graph (the collective is one more node of the captured stream).

Two communicators on one device (torch's and this one) must never have collectives in flight in different orders on
different ranks.  Under DDP they do not: the packed exchange runs inside the quantizer's forward, in program order on every
rank, and nothing of torch's is in flight there (DDP's gradient buckets of the previous step were waited for by the optimizer
step; tests/test_gpu_two_ranks.py wraps the quantizer in DistributedDataParallel, tests/rccl_ws1_child.py does so on the nccl
backend with both communicators alive in one process).  Under FSDP they can: its forward-prefetched all-gathers run on torch's
communicator on a side stream while this one would run on the compute stream — the documented multi-communicator hazard.

What a captured call costs (measured at world size 1, profiles/r04_rccl_ws1.json): a HIP graph that contains the RCCL call
replays ~21 us slower than the same graph without it — a non-kernel node — whichever stream it was captured on.

``VQHIP_ALLREDUCE`` selects the route: ``direct`` (this module; an error if it cannot be set up) or ``torch``
(``dist.all_reduce``, the reference's own route).  ``auto`` (the default) resolves to ``torch``: the direct route has
executed at world size 1 only (one-GPU builder boxes; RCCL refuses two ranks on one device), its bootstrap catches failures
but cannot catch a rank that hangs inside ``ncclCommInitRank``, and FSDP is unsafe next to it (above) — it stays opt-in until
a run with two or more ranks on the nccl backend has been recorded.  gloo groups (CPU tests, the shared-GPU plumbing runs)
always take ``dist.all_reduce``."""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch
import torch.distributed as dist

from . import _lib

_state = {'group': None, 'comm': None, 'error': None, 'generation': 0}


def mode() -> str:
    m = os.environ.get('VQHIP_ALLREDUCE', 'auto').lower()
    if m not in ('auto', 'direct', 'torch'):
        raise ValueError(f"VQHIP_ALLREDUCE must be 'auto', 'direct' or 'torch', got {m!r}")
    return m


def _agree(ok: bool, device) -> bool:
    """True iff ``ok`` on every rank of the default group (the ranks must take the same branch around a collective)."""
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item())


def _librccl_path() -> Optional[bytes]:
    path = os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so')
    return path.encode() if os.path.exists(path) else None


def _bootstrap(device: torch.device) -> int:
    """Create this library's communicator over the default process group; returns the ncclComm_t as an int."""
    L = _lib.lib()
    rank, world = dist.get_rank(), dist.get_world_size()
    err = None
    try:
        _lib.check(L.vqhip_rccl_load(_librccl_path()), 'vqhip_rccl_load')
    except _lib.VqhipError as exc:
        err = str(exc)
    if not _agree(err is None, device):                       # nobody enters ncclCommInitRank unless everybody can
        raise _lib.VqhipError(err or 'vqhip_rccl_load failed on another rank')
    # From here on every rank reaches BOTH agreement points below whatever fails locally: a rank that raised early would
    # leave the others waiting inside ncclCommInitRank or the agreement all-reduce.
    store = dist.distributed_c10d._get_default_store()
    _state['generation'] += 1
    key = f'vqhip/rccl_unique_id/{_state["generation"]}'
    ident, err = ctypes.create_string_buffer(128), None
    try:
        if rank == 0:
            try:
                _lib.check(L.vqhip_rccl_unique_id(ident), 'vqhip_rccl_unique_id')
                store.set(key, ident.raw.hex())
            except Exception:
                store.set(key, 'failed')                     # the other ranks are waiting for this key
                raise
        else:
            word = store.get(key).decode()                   # waits until rank 0 has set it
            if word == 'failed':
                raise _lib.VqhipError('vqhip_rccl_unique_id failed on rank 0')
            ident = ctypes.create_string_buffer(bytes.fromhex(word), 128)
    except Exception as exc:                                  # noqa: BLE001
        err = f'{type(exc).__name__}: {exc}'
    if not _agree(err is None, device):                       # nobody enters ncclCommInitRank unless everybody has the id
        raise _lib.VqhipError(err or 'the RCCL unique id did not reach every rank')
    comm, good = ctypes.c_void_p(), False
    try:
        with torch.cuda.device(device):
            _lib.check(L.vqhip_rccl_comm_init(ctypes.byref(comm), world, ident, rank), 'vqhip_rccl_comm_init')
            probe = torch.ones(8, dtype=torch.float32, device=device)
            stream = torch.cuda.current_stream(device).cuda_stream
            rc = L.vqhip_allreduce_packed(probe.data_ptr(), probe.numel(), comm, stream)
            good = rc == 0 and bool((probe == float(world)).all().item())
    except Exception as exc:                                  # noqa: BLE001
        err = f'{type(exc).__name__}: {exc}'
    if not _agree(good, device):
        if comm.value:
            L.vqhip_rccl_comm_destroy(comm)
        raise _lib.VqhipError(err or f'probe all-reduce through vqhip_allreduce_packed did not return {world} on every rank')
    return comm.value


def communicator(t: torch.Tensor, dtype: torch.dtype = torch.float32) -> Optional[int]:
    """The ncclComm_t for ``t``'s exchange, or None when ``dist.all_reduce`` is to be used (see the module docstring)."""
    m = mode()
    if m != 'direct' or not (dist.is_available() and dist.is_initialized()):      # 'auto' resolves to 'torch' (module docstring)
        return None
    if not (t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        return None
    if dist.get_backend() != 'nccl':
        if m == 'direct':
            raise _lib.VqhipError(f"VQHIP_ALLREDUCE=direct needs the RCCL ('nccl') backend, the process group runs {dist.get_backend()!r}")
        return None
    group = dist.distributed_c10d._get_default_group()
    if _state['group'] is not group:                          # first exchange of this process group (never inside a capture:
        if _state['comm'] is not None:                        # a captured step has run eagerly at least once before)
            try:                                              # the process group was re-created without shutdown(): the old
                torch.cuda.synchronize()                      # communicator is destroyed, not leaked
                _lib.lib().vqhip_rccl_comm_destroy(_state['comm'])
            except Exception:                                 # noqa: BLE001
                pass
        _state.update(group=group, comm=None, error=None)
        try:
            _state['comm'] = _bootstrap(t.device)
        except Exception as exc:                              # noqa: BLE001 — reported through `status()`, or raised in direct mode
            _state['error'] = f'{type(exc).__name__}: {exc}'
    if _state['comm'] is None and m == 'direct':
        raise _lib.VqhipError(f'VQHIP_ALLREDUCE=direct: {_state["error"]}')
    return _state['comm']


def all_reduce(t: torch.Tensor, comm: int) -> torch.Tensor:
    """In-place fp32 SUM of ``t`` over the ranks, enqueued on the current stream."""
    L = _lib.lib()
    _lib.check(L.vqhip_allreduce_packed(t.data_ptr(), t.numel(), comm, torch.cuda.current_stream(t.device).cuda_stream),
               'vqhip_allreduce_packed')
    return t


def all_reduce_min(t: torch.Tensor, comm: int) -> torch.Tensor:
    """In-place int64 MIN of ``t`` over the ranks, enqueued on the current stream (the key exchange of NearestAnchor(sync=True))."""
    L = _lib.lib()
    _lib.check(L.vqhip_allreduce_min_i64(t.data_ptr(), t.numel(), comm, torch.cuda.current_stream(t.device).cuda_stream),
               'vqhip_allreduce_min_i64')
    return t


def status() -> dict:
    """Which route the exchange takes in this process (bench.py / tests)."""
    return {'mode': mode(), 'direct': _state['comm'] is not None, 'error': _state['error']}


def shutdown() -> None:
    """Destroy the communicator (call before ``dist.destroy_process_group``; a process that simply exits need not)."""
    if _state['comm'] is not None:
        torch.cuda.synchronize()
        _lib.lib().vqhip_rccl_comm_destroy(_state['comm'])
    _state.update(group=None, comm=None, error=None)
