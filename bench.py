#!/usr/bin/env python3
"""bench.py — quantized tokens/s of the VQ codebook-lookup hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--images B] [--workload vqgan|cvq|vqkd|tokenize]

`--gpus N` with N > 1 from a bare invocation (no WORLD_SIZE in the environment) starts the N ranks itself: the parent —
which makes NO GPU call — checks that N devices exist, runs `python -m torch.distributed.run --nproc-per-node N bench.py
...` as a child process, relays rank 0's JSON line and exits with the child's code (the reference's analogue:
`auto_torchrun`, docs/training.md:15).  Under an external launcher (RANK/WORLD_SIZE set) it is one rank of the job.

Workloads (BASELINE.json configs; each "step" is one pass of the hot path over one HBM-resident synthetic batch):

  vqgan    (default; configs[1], the metric's config) VQGAN quantizer K=16384 D=256, 256x256 images -> 16x16 tokens, bf16
           latents as under autocast, 2048 images (524 288 tokens) per GPU and step (SURVEY.md §8d lists B in {32, 256,
           2048}; --images selects another).  One step = `VQGANQuantizer.forward(x, memo)` of the drop-in nn.Module in
           eval mode: codebook preparation, fused distance+argmin, embedding gather, straight-through output, VQGAN
           loss.  Ranks are independent (tokens shard embarrassingly): weak scaling, no data-path
           collective.  With more than one rank the line ALSO carries `cvq` — the communicating workload below at 3072
           and 65 536 tokens per rank — so that a scaling run shows the one exchange step of the path, not only
           independent shards.
  cvq      (configs[3]) CVQ-VAE training step: VQGANQuantizer + CVQVAECallback(NearestAnchor), cosine, K=16384 D=256,
           per-rank batch 12 images = 3072 tokens (configs/vqgan/interface.py:8 over 8 ranks); forward (encode, sparse
           anchor list, column argmin over the listed codes, ONE packed all-reduce of histogram ‖ token count ‖ anchors,
           EMA update, decode, loss) + backward from a given upstream gradient.  Weak scaling.
  vqkd     (configs[2] as a TRAINING step) VQ-KD: VQKDQuantizer + VQKDCallback(ema) + CommitmentLoss(norm=True), cosine,
           K=8192 D=32, per-rank batch 64 images = 12 544 tokens (configs/vqkd/interface.py:8: 512 images over 8 ranks, 14x14
           tokens each; configs/vqkd/model.py:20-26); forward (normalise, encode, histogram + centroid sums, ONE packed
           all-reduce, EMA update, decode, normalised commitment loss — one library call) + backward.  Weak scaling.
  tokenize (configs[4]) LlamaGen bulk tokenization: 2048 images per step IN TOTAL, sharded over the ranks
           (`encode` only, D=8 + NormalizeCallback + L2: configs/llamagen/vqgan.py:10-20).  Strong scaling, no collective.

Timing: W warm-up steps, then blocks of EXACTLY K steps, each bracketed by barrier + synchronize on both sides and
reduced with MAX over the ranks; blocks repeat until --min-seconds of timed GPU work have accumulated (the GPU is then
visible to an external sampler and box-to-box variance shows); `value` / `ms_per_step` are those of the MEDIAN block, all
blocks are listed in `repeats`.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  roofline     — the dominant kernel (fp16-MFMA proposal pass) against the dense MFMA peak, timed live with HIP events
                 recorded on the launch stream (libvqhip's vqhip_profile_* hooks)
  parity       — the timed batch's indices checked, outside the timed region, against the all-fp32 route on every row and
                 against the CPU oracle on a row sample; which path every row took; `checker_ms` / `checker_tflops`: what that
                 whole-batch fp32 pass (vqhip_argmin_exact: exact_stream_kernel) took, for the record — never part of `value`
  cpu_baseline — the reference's ATen composition (oracle/torch_ref.py, byte-identical to the reference's own files on
                 every fixture: tests/test_reference_pin.py) timed on this box's host cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import ctypes
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

K_CODES, DIM, TOK_PER_IMAGE = 16384, 256, 256
MFMA_F16_DENSE_PEAK_TFLOPS = 2500.0       # MI355X_MICROARCH.md: ~2.5 PF dense bf16/fp16
SUSTAINED_F16_RANDOM_TFLOPS = 1768.0    # measured, this pool's MI355X (profiles/r05_mfma_random.txt): reported beside the roofline, never as its peak
HBM_PEAK_GBS = 8000.0
EMB = 'torch_nn_modules_sparse_Embedding'


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--images', type=int, default=None,
                    help='images per GPU per step (256 tokens each; vqkd: 196); default 2048 (vqgan), 12 (cvq), 64 (vqkd); '
                         'tokenize: images per step over ALL ranks, default 2048')
    ap.add_argument('--workload', choices=('vqgan', 'cvq', 'vqkd', 'tokenize'), default='vqgan')
    ap.add_argument('--min-seconds', type=float, default=8.0,
                    help='repeat the K-step block until this much timed GPU work has accumulated (0: one block)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--bind', choices=('auto', 'on', 'off'), default='auto',
                    help='pin the rank to one L3 slice of its GPU\'s NUMA node (vector_quantization_amd/affinity.py). auto: on for the '
                         'host-sensitive per-rank training workloads (cvq, vqkd) and for every multi-GPU run (its cvq block is one; ranks '
                         'take successive slices), off for the single-GPU GPU-bound ones')
    ap.add_argument('--no-verify', action='store_true', help='skip the parity self-check after the timed region')
    ap.add_argument('--no-cvq', action='store_true', help='world > 1, vqgan workload: skip the communicating cvq block')
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------------------------
# N-rank launch from a bare `python bench.py --gpus N` (parent never touches the GPU)
# ------------------------------------------------------------------------------------------------------------------

def _free_port() -> int:
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def count_gpus_sysfs():
    """GPUs of this node as the kernel driver lists them — KFD topology nodes with SIMDs (CPU nodes have simd_count 0) — cut to
    the devices a HIP/ROCR visibility mask leaves.  Pure file reads: the launching parent opens no device and loads no GPU
    runtime (round-5 review: `torch.cuda.device_count()` is not guaranteed to stay clear of it on every ROCm build).  None when
    the topology is not readable (the ranks then find out for themselves)."""
    import glob
    if not os.path.isdir('/sys/class/kfd'):          # no amdgpu compute driver on this host: no GPU
        return 0
    nodes = glob.glob('/sys/class/kfd/kfd/topology/nodes/*/properties')
    if not nodes:
        return None
    n = 0
    for path in nodes:
        try:
            for ln in open(path):
                f = ln.split()
                if len(f) == 2 and f[0] == 'simd_count' and int(f[1]) > 0:
                    n += 1
        except OSError:
            return None
    for var in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        mask = os.environ.get(var)
        if mask is not None and mask.strip() != '':
            n = min(n, len([m for m in mask.split(',') if m.strip() != '']))
    return n


def launch_ranks(n: int) -> int:
    have = count_gpus_sysfs()
    if have is not None and have < n and os.environ.get('VQ_BENCH_SHARE_GPU') != '1':      # (share mode: a plumbing check, every rank on cuda:0 over gloo)
        print(f'bench.py: {n}-rank launch failed: --gpus {n} but this node has {have} GPU(s); refusing to start '
              f'(never a silent smaller run)', file=sys.stderr)
        return 2
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)      # stderr passes through
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith('{') and '"metric"' in ln]
    for ln in proc.stdout.splitlines():
        if ln not in lines:
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or len(lines) != 1:
        print(f'bench.py: {n}-rank launch failed (rc={proc.returncode}, json lines={len(lines)})', file=sys.stderr)
        return proc.returncode or 1
    print(lines[0], flush=True)
    return 0


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N=1 only)
# ------------------------------------------------------------------------------------------------------------------

def _time_cpu_forward(tr, x, w, distance, loss, reps):
    import numpy as np
    import torch
    ncpu = os.cpu_count() or 1
    best = None
    # ATen's CPU GEMM does not always scale to every hardware thread: keep the fastest of a few thread counts
    for threads in sorted({ncpu, max(1, ncpu // 2), min(ncpu, 64), min(ncpu, 32)}, reverse=True):
        torch.set_num_threads(threads)
        times = []
        with torch.no_grad():
            for i in range(reps + 1):
                t0 = time.perf_counter()
                out = tr.forward(x, w, distance, loss)
                float(out['loss'])
                t1 = time.perf_counter()
                if i > 0:
                    times.append(t1 - t0)
        med_t = float(np.median(times))
        if best is None or med_t < best[0]:
            best = (med_t, threads)
    return best


def cpu_baseline(reps: int = 5):
    """The reference path (torch.cdist -> argmin -> embedding -> VQGAN loss -> STE) on the host cores, as BASELINE.md §2
    states: median of `reps` forwards after a warm-up at C2 with 32 images (8192 tokens, K=16384, D=256) and at C1."""
    import torch

    from oracle import synth, torch_ref as tr

    n2 = 32 * TOK_PER_IMAGE
    x = torch.from_numpy(synth.bf16_round(synth.normal(3407, n2, DIM)))
    w = torch.from_numpy(synth.normal(3408, K_CODES, DIM))
    med, used = _time_cpu_forward(tr, x, w, 'L2', 'vqgan', reps)
    x1 = torch.from_numpy(synth.normal(3407, 1024, 256))
    w1 = torch.from_numpy(synth.normal(3408, 1024, 256))
    med1, used1 = _time_cpu_forward(tr, x1, w1, 'L2', 'vqgan', reps)
    torch.set_num_threads(used)
    return {
        'value': n2 / med, 'unit': 'tokens/s', 'cores': os.cpu_count() or 1, 'threads_used': used, 'kind': 'port',
        'sample': f'median of {reps} timed forwards (after 1 warm-up) of the reference ATen path (torch.cdist+argmin+'
                  f'embedding+VQGAN loss+STE, fp32; oracle/torch_ref.py, pinned byte-identical to the reference files) '
                  f'over {n2} tokens (32 images), K={K_CODES}, D={DIM}; `cores` = os.cpu_count() of this host (BASELINE.md §2), `threads_used` = the '
                  f'torch thread count that was fastest among a few (ATen\'s CPU GEMM does not scale to every hardware thread)',
        'ms_per_forward': med * 1e3,
        'c1': {'value': 1024 / med1, 'unit': 'tokens/s', 'cores': os.cpu_count() or 1, 'threads_used': used1, 'ms_per_forward': med1 * 1e3,
               'sample': f'same, BASELINE configs[0]: 1024 tokens, K=1024, D=256, median of {reps}'},
    }


# ------------------------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------------------------

def quantizer_cfg(K, D, distance, callbacks=()):
    return dict(type='VQGANQuantizer', embedding=dict(type=EMB, num_embeddings=K, embedding_dim=D),
                distance=dict(type=f'{distance}Distance'), losses=dict(vqgan_loss=dict(type='VQGANLoss')),
                callbacks=list(callbacks))


def build_module(cfg, dev, w, train):
    import torch

    from vector_quantization_amd import Config, build_quantizer
    q = build_quantizer(cfg)
    q.train(train)
    q.init_weights(Config(type='vqgan'))
    q = q.to(dev)
    with torch.no_grad():
        q.embedding.weight.copy_(w)
    return q


class Bench:
    """Process-wide context of one rank: device, process group, timing helpers."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.args = torch, dist, args
        self.rank = int(os.environ.get('RANK', '0'))
        local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        self.world = int(os.environ.get('WORLD_SIZE', '1'))
        if self.world > 1 and self.world != args.gpus:
            args.gpus = self.world
        assert torch.cuda.is_available(), 'bench.py needs MI355X GPUs (no CPU path)'
        # VQ_BENCH_SHARE_GPU=1 is a plumbing check for boxes with fewer GPUs than ranks: every rank uses cuda:0 and the
        # collectives run over gloo (RCCL refuses two ranks on one device).  Never set by the driver.
        self.share_gpu = os.environ.get('VQ_BENCH_SHARE_GPU') == '1'
        if self.share_gpu:
            local_rank = 0
        if local_rank >= torch.cuda.device_count():
            sys.exit(f'bench.py: rank {self.rank} has no device {local_rank} (node has {torch.cuda.device_count()})')
        torch.cuda.set_device(local_rank)
        self.dev = torch.device('cuda', local_rank)
        # launched by torch.distributed.run — or as the direct-route child of such a rank (its own rendezvous, env:// from the parent)
        self.distributed = self.world > 1 or 'TORCHELASTIC_RUN_ID' in os.environ or os.environ.get('VQ_BENCH_CHILD') == '1'
        self.backend = None
        if self.distributed:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29500')
            self.backend = 'gloo' if self.share_gpu else 'nccl'
            if self.share_gpu:
                dist.init_process_group('gloo')
            else:
                dist.init_process_group('nccl', device_id=self.dev)
        self.coll_dev = 'cpu' if self.share_gpu else self.dev
        # host-thread placement (a launcher's job, done here because the rank knows its device only now): see affinity.py
        self.binding = None
        want = getattr(args, 'bind', 'off')
        if want == 'on' or (want == 'auto' and (getattr(args, 'workload', '') in ('cvq', 'vqkd') or (self.world > 1 and not self.share_gpu))):
            from vector_quantization_amd import affinity
            torch.cuda.init()
            self.binding = affinity.bind_rank(local_rank, int(os.environ.get('LOCAL_RANK', '0')), probe=True,
                                              world_on_node=int(os.environ.get('LOCAL_WORLD_SIZE', str(self.world))))
        # ranks that really take part in the collectives (all-reduce of ones)
        self.rccl_ranks = 1
        if self.distributed:
            ones = torch.ones(1, dtype=torch.int64, device=self.coll_dev)
            dist.all_reduce(ones)
            self.rccl_ranks = int(ones.item())
            assert self.rccl_ranks == self.world, f'{self.rccl_ranks} ranks answered the all-reduce, expected {self.world}'

    def barrier(self):
        self.torch.cuda.synchronize()
        if self.distributed:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def all_ranks(self, ok: bool) -> bool:
        """True iff `ok` on every rank (MIN all-reduce): decisions that change which collectives a rank issues."""
        if not self.distributed:
            return ok
        t = self.torch.tensor([1 if ok else 0], dtype=self.torch.int64, device=self.coll_dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return bool(t.item())

    def timed(self, fn, steps, warmup, profile=False):
        """W warm-up steps, then EXACTLY `steps` steps between barrier + synchronize; MAX over the ranks."""
        from vector_quantization_amd import _lib
        L = _lib.lib()
        for _ in range(warmup):
            fn()
        self.barrier()
        if profile:
            L.vqhip_profile_enable(1)
        t0 = time.perf_counter()
        out = None
        for _ in range(steps):
            out = fn()
        self.barrier()
        t1 = time.perf_counter()
        prof = None
        if profile:
            ms_sum, launches = ctypes.c_double(0), ctypes.c_int64(0)
            _lib.check(L.vqhip_profile_collect(ctypes.byref(ms_sum), ctypes.byref(launches)), 'vqhip_profile_collect')
            L.vqhip_profile_enable(0)
            prof = (ms_sum.value, launches.value)
        el = self.torch.tensor([t1 - t0], dtype=self.torch.float64, device=self.coll_dev)
        self.last_rank_seconds = [t1 - t0]
        if self.distributed:
            every = [self.torch.zeros_like(el) for _ in range(self.world)]
            self.dist.all_gather(every, el)                       # each rank's own clock around the same K steps
            self.last_rank_seconds = [float(t.item()) for t in every]
            self.dist.all_reduce(el, op=self.dist.ReduceOp.MAX)
        return float(el.item()), out, prof

    def timed_blocks(self, fn, steps, warmup, min_seconds, max_blocks=200):
        """Blocks of exactly `steps` steps until `min_seconds` of timed work; the proposal kernel is event-timed in the
        first block.  Returns (per-block seconds, last output, (kernel ms sum, launches))."""
        el, out, prof = self.timed(fn, steps, warmup, profile=True)
        blocks = [el]
        self.block_rank_seconds = [self.last_rank_seconds]
        while sum(blocks) < min_seconds and len(blocks) < max_blocks:
            el, out, _ = self.timed(fn, steps, 0)
            blocks.append(el)
            self.block_rank_seconds.append(self.last_rank_seconds)
        return blocks, out, prof

    def gather_floats(self, value: float):
        """[value on rank 0, value on rank 1, ...] on every rank."""
        if not self.distributed:
            return [float(value)]
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, device=self.coll_dev)
        every = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(every, t)
        return [float(v.item()) for v in every]


def _median_block(blocks):
    s = sorted(blocks)
    return s[(len(s) - 1) // 2]             # an actual block (lower median), not an interpolation


def train_cfg(kind):
    """(K, D, quantizer config, callback config list) of the two callback-driven training workloads."""
    if kind == 'vqkd':         # configs/vqkd/model.py:20-26
        cb_cfg = [dict(type='VQKDCallback', ema=dict())]
        K, D = 8192, 32
        cfg = dict(type='VQKDQuantizer', embedding=dict(type=EMB, num_embeddings=K, embedding_dim=D), distance=dict(type='CosineDistance'),
                   losses=dict(commitment_loss=dict(type='CommitmentLoss', mse=dict(norm=True))), callbacks=cb_cfg)
        return K, D, cfg
    if kind == 'cluster':      # configs/cluster/model.py:18-30: CodebookLoss, NearestAnchor(sync=True) — the global nearest latent per code
        K, D = 8192, 768
        cb_cfg = [dict(type='CVQVAECallback', ema=dict(), anchor=dict(type='NearestAnchor', sync=True))]
        cfg = dict(type='VQGANQuantizer', embedding=dict(type=EMB, num_embeddings=K, embedding_dim=D), distance=dict(type='CosineDistance'),
                   losses=dict(vqgan_loss=dict(type='CodebookLoss')), callbacks=cb_cfg)
        return K, D, cfg
    cb_cfg = [dict(type='CVQVAECallback', ema=dict(), anchor=dict(type='NearestAnchor'))]
    return K_CODES, DIM, quantizer_cfg(K_CODES, DIM, 'Cosine', cb_cfg)


def build_train_module(kind, cfg, dev, w):
    import torch

    from vector_quantization_amd import Config, build_quantizer
    q = build_quantizer(cfg)
    q.train(True)
    q.init_weights(Config(type='vqgan') if kind in ('cvq', 'cluster') else Config())
    q = q.to(dev)
    q._forward_pre_hooks.clear()          # VQ-KD: the k-means lazy init is not part of a steady-state step
    with torch.no_grad():
        q.embedding.weight.copy_(w)
    if kind == 'vqkd':                    # configs/vqkd/model.py:76-82: no_grad on the quantizer's parameters
        for p_ in q.parameters():
            p_.requires_grad_(False)
    return q


def run_cvq(B: Bench, tokens: int, steps: int, warmup: int, min_seconds: float, graphs: bool = True, settle: int = 150, kind: str = 'cvq'):
    """The CVQ-VAE (or, kind='vqkd', the VQ-KD) training step at `tokens` per rank (module + callbacks, forward + backward
    from a given upstream gradient), eager and replayed from HIP graphs; exchange accounting; codebook-in-sync check."""
    torch, dist = B.torch, B.dist
    from vector_quantization_amd.utils import exchange_log
    K, D, cfg = train_cfg(kind)
    dev = B.dev
    w = torch.nn.functional.normalize(torch.randn(K, D, device=dev, generator=torch.Generator(device=dev).manual_seed(3407)))
    g = torch.Generator(device=dev).manual_seed(3407 + B.rank)
    # codebook identical on all ranks, latents per rank.  A pool of up to 128 batches drawn around the rows of the initial
    # codebook, another one every step: what the quantizer of a training run sees — data that changes from step to step and a
    # codebook most of whose codes are in use.  (With ONE fixed batch the same <= `tokens` codes are hit for ever, the dead
    # ones are moved onto tokens that already have a code and stay dead: 13 000 codes listed at every step.  A real run is not
    # in that state; `exchange_rows_first_step` reports the worst case — every code listed — which is what step 1 costs.)
    pool = [(w[torch.randint(0, K, (tokens,), device=dev, generator=g)] + 0.05 * torch.randn(tokens, D, device=dev, generator=g))
            .requires_grad_(True) for _ in range(max(8, min(128, (1 << 19) // tokens)))]
    x = pool[0]
    gz = torch.randn(tokens, D, device=dev, generator=g) / (tokens * D)          # what the decoder's backward would hand back
    turn = [0]
    q = build_train_module(kind, cfg, dev, w)
    q_params = [p_ for p_ in q.parameters() if p_.requires_grad]
    cvq_cb = q._callbacks.callbacks[0]

    def make_step(module_call, params):
        def step():
            xin = pool[turn[0] % len(pool)]
            turn[0] += 1
            for p_ in params:
                p_.grad = None
            xin.grad = None
            z, loss, extra_ = module_call(xin)
            torch.autograd.backward([loss, z], [None, gz])
            return z, loss, extra_
        return step

    step = make_step(lambda xin: q(xin, {}), q_params)
    # settle: the probabilities start at 0 (every code listed, a [K, D] exchange); a training run spends its life in the
    # steady state, where only codes that have gone unused for ~100 steps are listed
    rows_first = None
    for i in range(settle):
        step()
        if i == 0:
            rows_first = getattr(cvq_cb, 'last_exchange_rows', None)
    blocks, out, prof = B.timed_blocks(step, steps, warmup, min_seconds)
    el = _median_block(blocks)
    rec = {'tokens_per_rank': tokens, 'ms_per_step': el / steps * 1e3, 'tokens_per_s': tokens * B.world * steps / el,
           'per_rank_ms_per_step': [round(v / steps * 1e3, 5) for v in B.block_rank_seconds[blocks.index(el)]],
           'blocks': len(blocks), 'ms_per_step_min': min(blocks) / steps * 1e3, 'ms_per_step_max': max(blocks) / steps * 1e3,
           'settle_steps': settle, 'exchange_rows_first_step': rows_first, 'exchange_rows': getattr(cvq_cb, 'last_exchange_rows', None),
           'one_call_forward': bool(q._one_call_step(x) is not None)}
    # exchange accounting on a few extra steps (events around the collective: not part of the timed blocks)
    exchange_log.start(timing=B.distributed and not B.share_gpu)
    n_acc = 10
    for _ in range(n_acc):
        step()
    st = exchange_log.stop()
    rec['collectives_per_step'] = st['calls'] / n_acc
    rec['exchange_bytes_per_step'] = st['bytes'] / n_acc
    rec['collective_ms'] = (st['ms'] / n_acc) if st['ms'] is not None else None
    from vector_quantization_amd import rccl
    from vector_quantization_amd.utils import exchanging
    rec['exchange_route'] = ({**rccl.status(), 'what': 'direct = vqhip_allreduce_packed on the compute stream (own ncclComm, '
                              'bootstrapped through the torch store); otherwise torch.distributed.all_reduce'}
                             if exchanging() else None)
    rec['dense_exchange_bytes_per_step'] = 8 * (K + 1) + 4 * K * D if B.world > 1 else 0     # int64[K+1] + fp32[K, D]: the reference's flow
    if kind == 'cluster' and B.world > 1:     # sync=True: the reference all-gathers latents, the [N, K] matrix, tokens and probabilities
        rec['dense_exchange_bytes_per_step'] = 8 * (K + 1) + B.world * (4 * tokens * D + 4 * tokens * K + 8 * tokens + 4 * K)
    rec['loss'] = float(out[1].item())
    rec['kernel_ms'] = prof[0] / max(1, prof[1])
    rec['kernel_launches_per_step'] = prof[1] / max(1, steps)
    # HIP-graph replay of the same step (graphs.py); gloo cannot be captured.  Every rank must take the same branch: the
    # graphs contain the collective
    if graphs and (not B.distributed or B.backend == 'nccl'):
        from vector_quantization_amd.graphs import GraphedQuantizer
        qg = build_train_module(kind, cfg, dev, w)
        qg.load_state_dict(q.state_dict())
        err = None
        try:
            gq = GraphedQuantizer(qg, x.detach())
        except Exception as exc:                          # reported, never silently dropped
            gq, err = None, f'{type(exc).__name__}: {exc}'
        if B.all_ranks(gq is not None):
            gstep = make_step(lambda xin: gq(xin), [p_ for p_ in qg.parameters() if p_.requires_grad])
            gblocks, _, _ = B.timed_blocks(gstep, steps, 3, min_seconds)
            ge = _median_block(gblocks)
            rec['ms_per_step_graphed'] = ge / steps * 1e3
            rec['tokens_per_s_graphed'] = tokens * B.world * steps / ge
            rec['graphed_note'] = ('GraphedQuantizer: forward (with the in-place codebook update and the packed all-reduce) and '
                                   'backward replayed from HIP graphs; CVQ-VAE: captured at 128 / 1024 / 4096 / 8192 / K listed codes, chained through '
                                   'the pinned count word (the replay waits for the previous forward: the eager one-call step is the faster one)')
        else:
            rec['graphed_error'] = err or 'graph capture failed on another rank'
    wsum = q.embedding.weight.detach().double().sum().reshape(1).to(B.coll_dev)
    if B.distributed:                                         # the reference's is_sync invariant (callbacks/update.py:54-55)
        lo, hi = wsum.clone(), wsum.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        rec['codebook_in_sync'] = bool(lo.item() == hi.item())
        assert rec['codebook_in_sync'], 'codebooks diverged across ranks'
    used = out[2].get('encode', {}).get('hist') if isinstance(out[2], dict) else None
    rec['used_codes'] = int((used > 0).sum().item()) if used is not None else None
    return rec, prof


def direct_route_subblock(B: Bench, tokens: int, timeout_s: float = 240.0):
    """The communicating step once more with the exchange on the library's OWN communicator (VQHIP_ALLREDUCE=direct: RCCL enqueued
    on the compute stream, rccl.py) — in a CHILD process per rank with a rendezvous of its own.  That route has executed at world
    size 1 only (one-GPU builder boxes); its bootstrap catches failures but cannot catch a rank that hangs inside
    ncclCommInitRank.  Here such a hang costs `timeout_s` and ends as a reported failure of THIS sub-block: the parent ranks —
    which hold the line's numbers — only wait on their children and kill them (the exact process group they started) when the time
    is up.  A child is a fresh `python bench.py --workload cvq` started with subprocess (never an exec from this process)."""
    import signal
    torch, dist = B.torch, B.dist
    route = os.environ.get('VQ_BENCH_DIRECT_ROUTE') or ('torch' if B.share_gpu else 'direct')      # share mode (gloo): plumbing rehearsal only
    port = torch.tensor([_free_port() if B.rank == 0 else 0], dtype=torch.int64, device=B.coll_dev)
    dist.broadcast(port, 0)
    env = {k: v for k, v in os.environ.items() if not k.startswith('TORCHELASTIC_')}       # (the agent's store is not the child's)
    env.update(VQHIP_ALLREDUCE=route, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(int(port.item())), VQ_BENCH_CHILD='1',
               RANK=str(B.rank), WORLD_SIZE=str(B.world), LOCAL_RANK=os.environ.get('LOCAL_RANK', str(B.rank)))
    cmd = [sys.executable, os.path.abspath(__file__), '--gpus', str(B.world), '--workload', 'cvq', '--images', str(tokens // TOK_PER_IMAGE),
           '--steps', '20', '--warmup', '5', '--min-seconds', '0.5', '--no-cpu-baseline', '--bind', 'off']      # (the child inherits this rank's CPUs)
    t0 = time.perf_counter()
    rec = {'route_requested': route, 'tokens_per_rank': tokens, 'timeout_s': timeout_s,
           'what': 'the cvq step in a child process per rank with VQHIP_ALLREDUCE=' + route + ' (own ncclComm on the compute stream when direct)'}
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = proc.communicate(timeout=timeout_s)
        rec['child_rc'] = proc.returncode
        ok = proc.returncode == 0
        if not ok:
            rec['error'] = f'child exited with code {proc.returncode}: ' + (err or '')[-600:]
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)                   # the session this rank started, nothing else
        except ProcessLookupError:
            pass
        out, err = proc.communicate()
        ok = False
        rec['error'] = f'child did not finish within {timeout_s:.0f} s (killed): ' + (err or '')[-600:]
    rec['seconds'] = time.perf_counter() - t0
    all_ok = B.all_ranks(ok)
    if B.rank == 0 and ok:
        lines = [ln for ln in (out or '').splitlines() if ln.startswith('{') and '"metric"' in ln]
        if len(lines) == 1:
            child = json.loads(lines[0])
            blk = child.get('cvq', {})
            for k in ('ms_per_step', 'tokens_per_s', 'collectives_per_step', 'exchange_bytes_per_step', 'collective_ms', 'exchange_route',
                      'codebook_in_sync', 'exchange_rows', 'one_call_forward'):
                rec[k] = blk.get(k)
            rec['rccl_ranks'] = child.get('rccl_ranks')
        else:
            ok = False
            rec['error'] = f'child printed {len(lines)} JSON lines'
    rec['ok'] = bool(ok and all_ok)
    if not all_ok and 'error' not in rec:
        rec['error'] = 'the child of another rank failed or timed out'
    return rec


def verify_vqgan(B: Bench, q, x, w, out, N, K, D):
    """Outside the timed region: the timed batch's tokens against the all-fp32 route (every row) and the CPU oracle (a row
    sample); the loss against a float64 evaluation; which path the rows took."""
    torch = B.torch
    from oracle import c_oracle as co
    from vector_quantization_amd import ops
    quant = out[2]['quant'].reshape(-1)
    exact = ops.argmin_exact(x, w, 'L2')
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    exact = ops.argmin_exact(x, w, 'L2')                  # (timed on its second run: the whole-batch fp32 route)
    ev1.record()
    torch.cuda.synchronize()
    checker_ms = float(ev0.elapsed_time(ev1))
    mism = int((quant != exact).sum().item())
    rows = torch.linspace(0, N - 1, 256, device=x.device).long()
    ref = co.l2_argmin(x[rows].float().cpu().numpy(), w.cpu().numpy())
    omis = int((quant[rows].cpu().numpy() != ref).sum())
    idx2, st = ops.argmin(x, ops.prepare_codebook(w, 'L2'), return_stats=True)
    same_again = bool(torch.equal(idx2, quant))
    st = st.cpu().tolist()
    zf = w[exact].double()
    loss64 = float(1.25 * ((zf - x.double()) ** 2).mean().item())
    loss = float(out[1].item())
    rec = {'parity_checked_rows': N, 'mismatches': mism, 'checked_against': 'vqhip_argmin_exact (all-fp32 MFMA route) on every row of the timed batch',
           'oracle_rows': int(rows.numel()), 'oracle_mismatches': omis, 'oracle': 'oracle/vq_oracle.c l2_argmin on evenly spaced rows',
           'paths': {'second_proposal_pass_rows': st[0], 'multi_candidate_rerank_rows': st[1], 'whole_codebook_fp32_rows': st[2],
                     'single_candidate_rows': N - st[0] - st[1] - st[2]},
           'checker_ms': checker_ms, 'checker_tflops': 2.0 * N * K * D / checker_ms / 1e9,
           'checker_note': 'one vqhip_argmin_exact call over the whole batch (row norms, exact_stream_kernel, finalize): fp32 MFMA, outside the timed region',
           'deterministic_rerun': same_again, 'loss': loss, 'loss_float64': loss64,
           'loss_rel_err': abs(loss - loss64) / max(1e-30, abs(loss64))}
    rec['ok'] = bool(mism == 0 and omis == 0 and same_again and rec['loss_rel_err'] <= 1e-5)
    return rec


def lib_sha256() -> str:
    from vector_quantization_amd import _lib
    return hashlib.sha256(open(_lib.LIB_PATH, 'rb').read()).hexdigest()


def _claim_stdout():
    """stdout carries exactly ONE line, the JSON record: RCCL prints a version banner to stdout when a communicator is
    created (observed: 'RCCL version : 2.26.6 ... Librccl path : ...'), and any library may do the like.  File descriptor 1
    is pointed at stderr for the whole run; the returned writer puts the record on the real stdout."""
    sys.stdout.flush()
    real = os.dup(1)
    os.dup2(2, 1)

    def emit(line: str) -> None:
        sys.stdout.flush()
        os.write(real, (line + '\n').encode())
    return emit


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ and 'RANK' not in os.environ:
        sys.exit(launch_ranks(args.gpus))          # parent: no torch.cuda / HIP call has been made
    emit = _claim_stdout()

    B = Bench(args)
    torch, dist = B.torch, B.dist
    rank, world, dev = B.rank, B.world, B.dev

    from vector_quantization_amd import _lib, ops

    g = torch.Generator(device=dev).manual_seed(3407 + rank)
    extra = {}
    wl = args.workload
    L = _lib.lib()
    for kv in filter(None, os.environ.get('VQHIP_TUNE', '').split(',')):      # A/B knobs (results unchanged), e.g. VQHIP_TUNE=17=0
        k_, v_ = kv.split('=')
        L.vqhip_set_tuning(int(k_), int(v_))
    if 'VQHIP_TUNE_SLICES' in os.environ:                   # A/B knob (results unchanged): codebook slices
        L.vqhip_set_tuning(2, int(os.environ['VQHIP_TUNE_SLICES']))

    parity = None
    main_rank_seconds = None
    if wl == 'vqgan':
        images = args.images or 2048
        N, K, D = images * TOK_PER_IMAGE, K_CODES, DIM
        w = torch.randn(K, D, device=dev, generator=g)                          # random-init codebook (fp32)
        x = torch.randn(N, D, device=dev, generator=g).bfloat16()               # synthetic latents, bf16 (autocast)
        q = build_module(quantizer_cfg(K, D, 'L2'), dev, w, train=False)

        def step():
            with torch.no_grad():
                return q(x, {})

        def ops_step():
            cb = ops.prepare_codebook(w, 'L2')             # weight may change every step in training: re-prepared
            idx = ops.argmin(x, cb)
            return ops.gather_ste_loss(x, w, idx, need_z=False)

        qt = build_module(quantizer_cfg(K, D, 'L2'), dev, w, train=True)
        xt = x.clone().requires_grad_(True)
        qt_params = list(qt.parameters())

        def train_step():
            for p_ in qt_params:                 # what an optimizer's zero_grad(set_to_none=True) does with its parameter list
                p_.grad = None
            xt.grad = None
            z, loss, _ = qt(xt, {})
            (loss + z.float().mean()).backward()

        blocks, out, prof = B.timed_blocks(step, args.steps, args.warmup, args.min_seconds)
        main_rank_seconds = B.block_rank_seconds
        loss = float(out[1].item())
        hist = ops.hist(out[2]['quant'], K)                   # code-usage statistics outside the timed region
        used_codes = int((hist > 0).sum().item())
        if not args.no_verify:
            parity = verify_vqgan(B, q, x, w, out, N, K, D)
        side_steps = max(5, args.steps // 5)
        e_ops, _, _ = B.timed(ops_step, side_steps, 2)
        e_tr, _, _ = B.timed(train_step, side_steps, 2)
        extra['ops_step'] = {'ms_per_step': e_ops / side_steps * 1e3, 'tokens_per_s': N * world * side_steps / e_ops,
                             'what': 'the same forward through the tensor-level ops layer (round-1 bench step)'}
        extra['module_train'] = {'ms_per_step': e_tr / side_steps * 1e3, 'tokens_per_s': N * world * side_steps / e_tr,
                                 'what': 'VQGANQuantizer.forward in train mode + backward (fused HIP backward, codebook gradient)'}
        # VQ_BENCH_FORCE_CVQ=1: the multi-rank blocks at ANY world size — with `torch.distributed.run --nproc-per-node=1` and
        # VQ_FORCE_EXCHANGE=1 the closest rehearsal of the multi-GPU line a one-GPU box allows ON RCCL: the per-rank gathers, the
        # communicating step through the whole exchange flow, the direct-route children beside the parent's communicator
        if (world > 1 or (B.distributed and os.environ.get('VQ_BENCH_FORCE_CVQ') == '1')) and not args.no_cvq:
            # the communicating workload of the path, so that a scaling run is interpretable
            extra['cvq'] = {}
            # The eager step is ONE library call per forward (vqhip_cvq_forward) and GPU-bound at both sizes; it is what is
            # timed by default.  VQ_BENCH_CVQ_GRAPHS=1 adds the graph-replayed step: capture and replay of a step that contains
            # the collective have executed at world size 1 on RCCL, both routes (tests/test_gpu_rccl.py), never with more ranks
            # (one-GPU builder boxes) — a capture that goes wrong there would cost the whole scaling run.
            cvq_graphs = os.environ.get('VQ_BENCH_CVQ_GRAPHS') == '1'
            for toks in (12 * TOK_PER_IMAGE, 256 * TOK_PER_IMAGE):
                rec, _ = run_cvq(B, toks, max(20, args.steps), 5, min(1.0, args.min_seconds), graphs=cvq_graphs, settle=int(os.environ.get('VQ_BENCH_CVQ_SETTLE', '120')))
                if not cvq_graphs:
                    rec['graphed_note'] = 'not run: set VQ_BENCH_CVQ_GRAPHS=1 (graph replay of an RCCL collective has run at world size 1 only)'
                extra['cvq'][str(toks)] = rec
            # the cluster config's step (configs/cluster/model.py:28: NearestAnchor(sync=True), 6 272 x 8192 x 768 per rank): the key
            # exchange of DESIGN.md §4.5 — a MIN all-reduce of 8-byte keys in front of the packed SUM, no latent gathered
            rec, _ = run_cvq(B, 6272, max(20, args.steps), 5, min(1.0, args.min_seconds), graphs=False,
                             settle=int(os.environ.get('VQ_BENCH_CVQ_SETTLE', '120')), kind='cluster')
            rec['what'] = 'VQGANQuantizer + CVQVAECallback(NearestAnchor(sync=True)) + CodebookLoss, cosine, K=8192 D=768 (configs/cluster/model.py)'
            extra['cvq']['cluster_sync_6272'] = rec
            if os.environ.get('VQ_BENCH_DIRECT', '1') != '0' and os.environ.get('VQ_BENCH_CHILD') != '1':
                extra['cvq']['direct_route'] = direct_route_subblock(B, 12 * TOK_PER_IMAGE,
                                                                     timeout_s=float(os.environ.get('VQ_BENCH_DIRECT_TIMEOUT', '240')))
        tokens_per_step_global = N * world
        scaling = 'weak'
        workload = ('VQGAN K=16384 D=256, 256x256 images -> 16x16 tokens, bf16 latents, VQGANQuantizer.forward of the '
                    'drop-in nn.Module, eval mode (prepare+argmin+gather+STE+loss)')
        parallelism = f'dp{world} (independent shards, no data-path collective)'
        metric = 'quantized tokens/sec, VQGAN quantizer forward K=16384 D=256'
    elif wl == 'vqkd':
        images = args.images or 64
        K, D, _ = train_cfg('vqkd')
        N = images * 196
        rec, prof = run_cvq(B, N, args.steps, args.warmup, args.min_seconds,
                            graphs=(world == 1 or os.environ.get('VQ_BENCH_CVQ_GRAPHS') == '1'), kind='vqkd')
        blocks = [rec['ms_per_step'] * args.steps / 1e3]
        extra['vqkd'] = rec
        loss, used_codes = rec['loss'], rec['used_codes']
        tokens_per_step_global = N * world
        scaling = 'weak'
        workload = ('VQ-KD training step K=8192 D=32 cosine, VQKDQuantizer + VQKDCallback(ema) + CommitmentLoss(norm=True), 64 images x '
                    '14x14 tokens per rank: forward (normalise, encode, histogram + centroid sums, ONE packed all-reduce, EMA update, '
                    'decode, loss: one library call) + backward from a given upstream gradient')
        parallelism = (f'dp{world} (rows sharded, codebook replicated; one fp32 all-reduce of 2K+4+K*D floats per step over '
                       f'{B.backend or "no backend"})')
        metric = 'quantized tokens/sec, VQ-KD quantizer training step K=8192 D=32'
    elif wl == 'cvq':
        images = args.images or 12
        N, K, D = images * TOK_PER_IMAGE, K_CODES, DIM
        rec, prof = run_cvq(B, N, args.steps, args.warmup, args.min_seconds,
                            graphs=(world == 1 or os.environ.get('VQ_BENCH_CVQ_GRAPHS') == '1'))
        blocks = [rec['ms_per_step'] * args.steps / 1e3]
        extra['cvq'] = rec
        loss, used_codes = rec['loss'], rec['used_codes']
        tokens_per_step_global = N * world
        scaling = 'weak'
        workload = ('CVQ-VAE training step K=16384 D=256 cosine, VQGANQuantizer + CVQVAECallback(NearestAnchor): forward '
                    'with the sparse-anchor exchange (ONE packed all-reduce: histogram, token count, anchors of the listed '
                    'codes) and EMA codebook update, then backward from a given upstream gradient')
        parallelism = (f'dp{world} (rows sharded, codebook replicated; one fp32 all-reduce of 2K+4+M*D floats per step over '
                       f'{B.backend or "no backend"})')
        metric = 'quantized tokens/sec, CVQ-VAE quantizer training step K=16384 D=256'
    else:
        total = args.images or 2048
        assert total % world == 0, f'{total} images do not shard over {world} ranks'
        images = total // world
        N, K, D = images * TOK_PER_IMAGE, K_CODES, 8
        w = torch.randn(K, D, device=dev, generator=torch.Generator(device=dev).manual_seed(3407))
        x = torch.randn(N, D, device=dev, generator=g).bfloat16()
        q = build_module(quantizer_cfg(K, D, 'L2', [dict(type='NormalizeCallback')]), dev, w, train=False)

        def step():
            with torch.no_grad():
                return q.encode(x, {})

        blocks, out, prof = B.timed_blocks(step, args.steps, args.warmup, args.min_seconds)
        main_rank_seconds = B.block_rank_seconds
        loss = None
        used_codes = int((ops.hist(out[1], K) > 0).sum().item())
        if not args.no_verify:                                   # tokens of the timed batch against the all-fp32 route
            xn, wn = ops.normalize_rows(x), ops.normalize_rows(w)
            mism = int((out[1].reshape(-1) != ops.argmin_exact(xn, wn, 'L2')).sum().item())
            parity = {'parity_checked_rows': N, 'mismatches': mism, 'ok': mism == 0,
                      'checked_against': 'vqhip_argmin_exact on the normalised operands, every row of the timed batch'}
        tokens_per_step_global = N * world
        scaling = 'strong'
        workload = ('LlamaGen bulk tokenization: 2048 images per step in total, sharded over the ranks; '
                    'VQGANQuantizer.encode with NormalizeCallback, K=16384 D=8, L2')
        parallelism = f'dp{world} (images sharded, no collective)'
        metric = 'quantized tokens/sec, LlamaGen tokenizer encode K=16384 D=8'

    # every rank checked its own timed batch: the line fails (exit code 3, after it is printed) if ANY rank's check failed
    parity_ok = True
    if parity is not None:
        bad = B.gather_floats(0.0 if parity.get('ok') else 1.0)
        mism_all = B.gather_floats(float(parity.get('mismatches', 0)))
        parity['ranks_checked'] = len(bad)
        parity['ranks_failed'] = [i for i, b in enumerate(bad) if b]
        parity['mismatches_all_ranks'] = int(sum(mism_all))
        parity_ok = not parity['ranks_failed']
    kern_ms_ranks = B.gather_floats(prof[0] / max(1, prof[1]))     # the dominant kernel's average launch on each rank
    if rank == 0:
        elapsed = _median_block(blocks)
        tokens = tokens_per_step_global * args.steps
        kern_ms = max(kern_ms_ranks)                               # the SLOWEST rank's kernel prices the roofline
        launches_per_step = prof[1] / max(1, args.steps)
        flops = 2.0 * N * K * D                                          # 2*K*D per token, per launch
        achieved_tf = flops / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else 0.0
        alg_bytes = N * (D * 2 + 8 + D * 4) + K * D * 4                  # §8(d): full forward, bf16 x, + codebook once
        sha = lib_sha256()
        traffic, traffic_source = None, None
        pmc = os.path.join(ROOT, 'profiles', 'pmc_latest.json')
        if wl == 'vqgan' and os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                if int(rec.get('tokens_per_launch', -1)) != N:
                    traffic_source = 'profiles/pmc_latest.json was collected at another batch size: not reported'
                elif rec.get('lib_sha256') != sha:                       # counters of ANOTHER build of the library say nothing about this one
                    traffic_source = ('profiles/pmc_latest.json was collected with another build of libvqhip.so '
                                      f'(sha256 {str(rec.get("lib_sha256"))[:12]}... != {sha[:12]}...): not reported')
                else:
                    traffic = rec.get('coarse_kernel_hbm_bytes_per_launch')
                    traffic_source = ('profiles/pmc_latest.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command '
                                      'with this very libvqhip.so, gfx950-corrected; NOT measured in this run): ' + str(rec.get('source')))
            except Exception:
                traffic = None
        # the proposal kernel's OWN algorithmic bytes (what `traffic` is to be compared with): the latents' 2 bytes per element read once
        # (the bf16 rows themselves since round 6 — the kernel makes its token fragments in its prologue; the fp16 token image before),
        # the fp16 codebook image once, the records written (5 floats per token and codebook slice; two slices at this size)
        dp = 32
        while dp < D:
            dp *= 2
        dp = D if D > 512 else dp
        rec_slices = 2
        kernel_alg_bytes = N * dp * 2 + K * dp * 2 + rec_slices * 5 * 4 * N
        step_ms = elapsed / args.steps * 1e3
        step_tf = flops / (step_ms * 1e-3) / 1e12
        roofline = {'bound': 'mfma', 'kernel': 'coarse_kernel (fp16 MFMA distance+argmin proposals)',
                    'achieved': achieved_tf, 'peak': MFMA_F16_DENSE_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': achieved_tf / MFMA_F16_DENSE_PEAK_TFLOPS, 'traffic': traffic,
                    'traffic_source': traffic_source,
                    'kernel_algorithmic_bytes': kernel_alg_bytes,
                    'traffic_over_kernel_algorithmic': (traffic / kernel_alg_bytes) if traffic else None,
                    'kernel_ms': kern_ms, 'kernel_ms_per_rank': [round(v, 5) for v in kern_ms_ranks],
                    'kernel_ms_source': 'HIP events on the launch stream, this run; with more than one rank `achieved` / `frac` are the SLOWEST rank\'s',
                    'launches_timed': prof[1], 'launches_per_step': launches_per_step,
                    'step_frac': step_tf / MFMA_F16_DENSE_PEAK_TFLOPS,
                    'step_frac_note': 'the same flops over the WHOLE step (ms_per_step: what `value` is), against the same peak',
                    'step_algorithmic_bytes': alg_bytes,
                    'hbm_frac_algorithmic': alg_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                    # context, NOT the roofline: what the same instruction sustains on this chip when nothing else runs
                    'measured_ceiling': {'tflops': SUSTAINED_F16_RANDOM_TFLOPS, 'frac_of_it': achieved_tf / SUSTAINED_F16_RANDOM_TFLOPS,
                                         'what': 'bare v_mfma_f32_16x16x32_f16 on all 1024 SIMDs with N(0,1) fp16 operands held in registers, launches '
                                                 'of 3 ms: matrix pipe 100 % busy at a shader clock of 1.78 GHz (power management; 2.31-2.40 GHz '
                                                 'and 2.34-2.43 PFLOP/s on constant operands) - profiles/r05_mfma_random.txt, tools/micro/mfma_random.hip'}}
        if wl in ('cvq', 'vqkd'):
            roofline['kernel_ms_note'] = ('average over the proposal launches of a step: the row pass (N x K) and, when codes are listed, '
                                          'the role-swapped column pass (listed codes x N) — `achieved` prices the row pass only')
        per_step = sorted(b / args.steps * 1e3 for b in blocks)
        per_rank_ms = None
        if main_rank_seconds is not None:                          # each rank's own clock around the median block's K steps
            per_rank_ms = [round(v / args.steps * 1e3, 5) for v in main_rank_seconds[blocks.index(elapsed)]]
        elif wl in ('cvq', 'vqkd'):
            per_rank_ms = extra[wl].get('per_rank_ms_per_step')
        out_line = {
            'metric': metric,
            'value': tokens / elapsed, 'unit': 'tokens/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
            'scaling': scaling, 'vs_baseline': None, 'dtype': 'f16+f32',
            'dtype_note': 'f16 MFMA (f32 accumulate) proposes candidates under a rigorous bound; the decision is exact f32',
            'data': 'synthetic', 'rccl_ranks': B.rccl_ranks, 'collective_backend': B.backend,
            'per_rank_ms_per_step': per_rank_ms,
            'config': {'workload': workload, 'images_per_gpu': images, 'tokens_per_gpu_per_step': N, 'codebook': [K, D],
                       'parallelism': parallelism},
            'timing_note': f'value / ms_per_step: the median of {len(blocks)} timed blocks of exactly {args.steps} steps each '
                           f'(barrier + synchronize on both sides, MAX over ranks); all blocks in `repeats`',
            'repeats': {'blocks': len(blocks), 'timed_seconds': sum(blocks), 'ms_per_step_median': elapsed / args.steps * 1e3,
                        'ms_per_step_min': per_step[0], 'ms_per_step_max': per_step[-1],
                        'ms_per_step_all': [round(v, 5) for v in (b / args.steps * 1e3 for b in blocks)][:64]},
            'roofline': roofline,
            'lib_sha256': sha,
            'loss': loss, 'used_codes': used_codes,
        }
        if parity is not None:
            out_line['parity'] = parity
        out_line.update(extra)
        if B.binding is not None:
            out_line['host_binding'] = {k: B.binding.get(k) for k in ('cpus', 'numa_local', 'slice', 'slices_on_node', 'position_on_node', 'ranks_on_node')}
            from vector_quantization_amd import affinity
            affinity.restore(B.binding['previous'])              # the CPU baseline below gets every core back
        if not args.no_cpu_baseline and world == 1:          # reported once, at N=1 (rank 0's host cores)
            out_line['cpu_baseline'] = cpu_baseline()
        emit(json.dumps(out_line))
    if B.distributed:
        dist.barrier()
        from vector_quantization_amd import rccl
        rccl.shutdown()
        dist.destroy_process_group()
    if not parity_ok:
        print(f'bench.py: parity self-check FAILED on rank(s) {parity["ranks_failed"]}: {parity}', file=sys.stderr)
        sys.exit(3)


if __name__ == '__main__':
    main()
