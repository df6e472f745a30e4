#!/usr/bin/env python3
"""bench.py — quantized tokens/s of the VQ codebook-lookup hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--images B] [--workload vqgan|cvq|tokenize]

`--gpus N` with N > 1 from a bare invocation (no WORLD_SIZE in the environment) starts the N ranks itself: the parent —
which makes NO GPU call — runs `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child process,
relays rank 0's JSON line and exits with the child's code (the reference's analogue: `auto_torchrun`, docs/training.md:15).
Under an external launcher (RANK/WORLD_SIZE set) it is one rank of the job.

Workloads (BASELINE.json configs; each "step" is one pass of the hot path over one HBM-resident synthetic batch):

  vqgan    (default; configs[1], the metric's config) VQGAN quantizer K=16384 D=256, 256x256 images -> 16x16 tokens, bf16
           latents as under autocast, 2048 images (524 288 tokens) per GPU and step (SURVEY.md §8d lists B in {32, 256,
           2048}; --images selects another).  One step = `VQGANQuantizer.forward(x, memo)` of the drop-in nn.Module in
           eval mode: codebook preparation, fused distance+argmin, embedding gather, straight-through output, VQGAN
           loss.  Ranks are independent (tokens shard embarrassingly): weak scaling, no data-path
           collective.  The same step through the tensor-level `ops` layer and a train-mode forward+backward are
           timed next to it (`ops_step`, `module_train`).
  cvq      (configs[3]) CVQ-VAE training step: VQGANQuantizer + CVQVAECallback(NearestAnchor), cosine, K=16384 D=256,
           per-rank batch 12 images = 3072 tokens (configs/vqgan/interface.py:8 over 8 ranks); forward (encode, histogram
           all-reduce, column argmin, anchor all-reduce, EMA update, decode, loss) + backward.  Weak scaling; the
           collective is RCCL all-reduce of int64[K+1] and fp32[K,D].
  tokenize (configs[4]) LlamaGen bulk tokenization: 2048 images per step IN TOTAL, sharded over the ranks
           (`encode` only, D=8 + NormalizeCallback + L2: configs/llamagen/vqgan.py:10-20).  Strong scaling, no collective.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  roofline     — the dominant kernel (fp16-MFMA proposal pass) against the dense MFMA peak, timed live with HIP events
                 recorded on the launch stream (libvqhip's vqhip_profile_* hooks)
  cpu_baseline — the reference's ATen composition (oracle/torch_ref.py, byte-identical to the reference's own files on
                 every fixture: tests/test_reference_pin.py) timed on this box's host cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import ctypes
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
HBM_PEAK_GBS = 8000.0
EMB = 'torch_nn_modules_sparse_Embedding'


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--images', type=int, default=None,
                    help='images per GPU per step (256 tokens each); default 2048 (vqgan), 12 (cvq); '
                         'tokenize: images per step over ALL ranks, default 2048')
    ap.add_argument('--workload', choices=('vqgan', 'cvq', 'tokenize'), default='vqgan')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------------------------
# N-rank launch from a bare `python bench.py --gpus N` (parent never touches the GPU)
# ------------------------------------------------------------------------------------------------------------------

def _free_port() -> int:
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launch_ranks(n: int) -> int:
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
        'value': n2 / med, 'unit': 'tokens/s', 'cores': used, 'kind': 'port',
        'sample': f'median of {reps} timed forwards (after 1 warm-up) of the reference ATen path (torch.cdist+argmin+'
                  f'embedding+VQGAN loss+STE, fp32; oracle/torch_ref.py, pinned byte-identical to the reference files) '
                  f'over {n2} tokens (32 images), K={K_CODES}, D={DIM}; best of several thread counts',
        'ms_per_forward': med * 1e3,
        'c1': {'value': 1024 / med1, 'unit': 'tokens/s', 'cores': used1, 'ms_per_forward': med1 * 1e3,
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


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ and 'RANK' not in os.environ:
        sys.exit(launch_ranks(args.gpus))          # parent: no torch.cuda / HIP call has been made

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world > 1 and world != args.gpus:
        args.gpus = world
    assert torch.cuda.is_available(), 'bench.py needs MI355X GPUs (no CPU path)'
    # VQ_BENCH_SHARE_GPU=1 is a plumbing check for boxes with fewer GPUs than ranks: every rank uses cuda:0 and the
    # collectives run over gloo (RCCL refuses two ranks on one device).  Never set by the driver.
    share_gpu = os.environ.get('VQ_BENCH_SHARE_GPU') == '1'
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    distributed = world > 1 or 'TORCHELASTIC_RUN_ID' in os.environ      # launched by torch.distributed.run
    backend = None
    if distributed:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        backend = 'gloo' if share_gpu else 'nccl'
        if share_gpu:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)
    coll_dev = 'cpu' if share_gpu else dev

    from vector_quantization_amd import _lib, ops

    # ranks that really take part in the collectives (all-reduce of ones)
    rccl_ranks = 1
    if distributed:
        ones = torch.ones(1, dtype=torch.int64, device=coll_dev)
        dist.all_reduce(ones)
        rccl_ranks = int(ones.item())
        assert rccl_ranks == world, f'{rccl_ranks} ranks answered the all-reduce, expected {world}'

    g = torch.Generator(device=dev).manual_seed(3407 + rank)
    extra = {}
    wl = args.workload
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

        tokens_per_step_global = N * world
        scaling = 'weak'
        workload = ('VQGAN K=16384 D=256, 256x256 images -> 16x16 tokens, bf16 latents, VQGANQuantizer.forward of the '
                    'drop-in nn.Module, eval mode (prepare+argmin+gather+STE+loss)')
        parallelism = f'dp{world} (independent shards, no data-path collective)'
        metric = 'quantized tokens/sec, VQGAN quantizer forward K=16384 D=256'
    elif wl == 'cvq':
        images = args.images or 12
        N, K, D = images * TOK_PER_IMAGE, K_CODES, DIM
        w = torch.nn.functional.normalize(torch.randn(K, D, device=dev, generator=torch.Generator(device=dev).manual_seed(3407)))
        x = torch.randn(N, D, device=dev, generator=g).requires_grad_(True)     # codebook identical on all ranks, latents per rank
        cb_cfg = [dict(type='CVQVAECallback', ema=dict(), anchor=dict(type='NearestAnchor'))]
        q = build_module(quantizer_cfg(K, D, 'Cosine', cb_cfg), dev, w, train=True)
        q_params = list(q.parameters())

        def step():
            for p_ in q_params:
                p_.grad = None
            x.grad = None
            z, loss, memo = q(x, {})
            (loss + z.mean()).backward()
            return z, loss, memo

        ops_step = train_step = None
        graph_step = None
        if not distributed or backend == 'nccl':            # HIP-graph replay of the same step (graphs.py); gloo cannot be captured
            from vector_quantization_amd.graphs import GraphedQuantizer
            qg = build_module(quantizer_cfg(K, D, 'Cosine', cb_cfg), dev, w, train=True)
            xg = x.detach().clone().requires_grad_(True)
            qg_params = list(qg.parameters())
            try:
                gq = GraphedQuantizer(qg, xg.detach())

                def graph_step():
                    for p_ in qg_params:
                        p_.grad = None
                    xg.grad = None
                    z, loss, _ = gq(xg)
                    (loss + z.mean()).backward()
            except Exception as exc:                          # reported, never silently dropped
                extra['module_graphed'] = {'error': f'{type(exc).__name__}: {exc}'}
        tokens_per_step_global = N * world
        scaling = 'weak'
        workload = ('CVQ-VAE training step K=16384 D=256 cosine, VQGANQuantizer + CVQVAECallback(NearestAnchor): forward '
                    'with histogram + anchor all-reduce and EMA codebook update, then backward')
        parallelism = f'dp{world} (rows sharded, codebook replicated; all-reduce of int64[K+1] and fp32[K,D] per step over {backend or "no backend"})'
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

        ops_step = train_step = None
        tokens_per_step_global = N * world
        scaling = 'strong'
        workload = ('LlamaGen bulk tokenization: 2048 images per step in total, sharded over the ranks; '
                    'VQGANQuantizer.encode with NormalizeCallback, K=16384 D=8, L2')
        parallelism = f'dp{world} (images sharded, no collective)'
        metric = 'quantized tokens/sec, LlamaGen tokenizer encode K=16384 D=8'

    def barrier():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps, warmup, profile=False):
        L = _lib.lib()
        for _ in range(warmup):
            fn()
        barrier()
        if profile:
            L.vqhip_profile_enable(1)
        t0 = time.perf_counter()
        out = None
        for _ in range(steps):
            out = fn()
        barrier()
        t1 = time.perf_counter()
        prof = None
        if profile:
            ms_sum, launches = ctypes.c_double(0), ctypes.c_int64(0)
            _lib.check(L.vqhip_profile_collect(ctypes.byref(ms_sum), ctypes.byref(launches)), 'vqhip_profile_collect')
            L.vqhip_profile_enable(0)
            prof = (ms_sum.value, launches.value)
        el = torch.tensor([t1 - t0], dtype=torch.float64, device=coll_dev)
        if distributed:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el.item()), out, prof

    L = _lib.lib()
    if 'VQHIP_TUNE_SLICES' in os.environ:                   # A/B knob (results unchanged): codebook slices
        L.vqhip_set_tuning(2, int(os.environ['VQHIP_TUNE_SLICES']))

    elapsed, out, prof = timed(step, args.steps, args.warmup, profile=True)
    if wl == 'vqgan':
        loss = float(out[1].item())
        hist = ops.hist(out[2]['quant'], K)                   # code-usage statistics outside the timed region
        used_codes = int((hist > 0).sum().item())
        side_steps = max(5, args.steps // 5)
        e_ops, _, _ = timed(ops_step, side_steps, 2)
        e_tr, _, _ = timed(train_step, side_steps, 2)
        extra['ops_step'] = {'ms_per_step': e_ops / side_steps * 1e3, 'tokens_per_s': N * world * side_steps / e_ops,
                             'what': 'the same forward through the tensor-level ops layer (round-1 bench step)'}
        extra['module_train'] = {'ms_per_step': e_tr / side_steps * 1e3, 'tokens_per_s': N * world * side_steps / e_tr,
                                 'what': 'VQGANQuantizer.forward in train mode + backward (fused HIP backward, codebook gradient)'}
    elif wl == 'cvq':
        if graph_step is not None:
            side_steps = max(10, args.steps)
            e_g, _, _ = timed(graph_step, side_steps, 3)
            extra['module_graphed'] = {'ms_per_step': e_g / side_steps * 1e3, 'tokens_per_s': N * world * side_steps / e_g,
                                       'what': 'the same training step replayed from HIP graphs (GraphedQuantizer: forward with '
                                               'in-place codebook update + backward, one launch each)'}
        loss = float(out[1].item())
        used_codes = int((out[2]['encode']['hist'] > 0).sum().item()) if 'hist' in out[2]['encode'] else None
        wsum = q.embedding.weight.detach().double().sum().reshape(1).to(coll_dev)
        if distributed:                                         # the reference's is_sync invariant (callbacks/update.py:54-55)
            lo, hi = wsum.clone(), wsum.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            extra['codebook_in_sync'] = bool(lo.item() == hi.item())
            assert extra['codebook_in_sync'], 'codebooks diverged across ranks'
    else:
        loss = None
        used_codes = int((ops.hist(out[1], K) > 0).sum().item())

    if rank == 0:
        tokens = tokens_per_step_global * args.steps
        kern_ms = prof[0] / max(1, prof[1])
        launches_per_step = prof[1] / max(1, args.steps)
        flops = 2.0 * N * K * D                                          # 2*K*D per token, per launch
        if wl == 'cvq':
            flops = flops                                                # row pass; the column pass is the same kernel, own launch
        achieved_tf = flops / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else 0.0
        alg_bytes = N * (D * 2 + 8 + D * 4) + K * D * 4                  # §8(d): full forward, bf16 x, + codebook once
        traffic, traffic_source = None, None
        pmc = os.path.join(ROOT, 'profiles', 'pmc_latest.json')
        if wl == 'vqgan' and os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                if int(rec.get('tokens_per_launch', -1)) == N:      # counters are per launch of THIS workload
                    traffic = rec.get('coarse_kernel_hbm_bytes_per_launch')
                    traffic_source = ('profiles/pmc_latest.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this '
                                      'command, gfx950-corrected; NOT measured in this run): ' + str(rec.get('source')))
            except Exception:
                traffic = None
        out_line = {
            'metric': metric,
            'value': tokens / elapsed, 'unit': 'tokens/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
            'scaling': scaling, 'vs_baseline': None, 'dtype': 'f16+f32',
            'dtype_note': 'f16 MFMA (f32 accumulate) proposes candidates under a rigorous bound; the decision is exact f32',
            'data': 'synthetic', 'rccl_ranks': rccl_ranks, 'collective_backend': backend,
            'config': {'workload': workload, 'images_per_gpu': images, 'tokens_per_gpu_per_step': N, 'codebook': [K, D],
                       'parallelism': parallelism},
            'roofline': {'bound': 'mfma', 'kernel': 'coarse_kernel (fp16 MFMA distance+argmin proposals)',
                         'achieved': achieved_tf, 'peak': MFMA_F16_DENSE_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': achieved_tf / MFMA_F16_DENSE_PEAK_TFLOPS, 'traffic': traffic,
                         'traffic_source': traffic_source,
                         'kernel_ms': kern_ms, 'launches_timed': prof[1], 'launches_per_step': launches_per_step,
                         'hbm_frac_algorithmic': alg_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS},
            'loss': loss, 'used_codes': used_codes,
        }
        out_line.update(extra)
        if not args.no_cpu_baseline and world == 1:          # reported once, at N=1 (rank 0's host cores)
            out_line['cpu_baseline'] = cpu_baseline()
        print(json.dumps(out_line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
