#!/usr/bin/env python3
"""bench.py — quantized tokens/s of the VQ codebook-lookup hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--images B]

Workload (BASELINE.json configs[1]): VQGAN quantizer, K=16384 codes, D=256, 256x256 images -> 16x16 tokens,
latents bf16-valued as under autocast, 2048 images (524 288 tokens) per GPU and step (SURVEY.md §8d lists
B in {32, 256, 2048} for this config; --images selects another batch).  One step = one quantizer forward over one synthetic batch that is
already resident in HBM: codebook preparation, fused distance+argmin (+ code histogram), embedding gather,
straight-through output and the VQGAN loss sums (the reference's VQGAN forward computes no histogram: vqgan/model.py:230).  Ranks are independent (tokens shard embarrassingly:
SURVEY.md §8e), so N>1 is weak scaling with no data-path collective; the only collectives are the timing
barrier and the max-over-ranks reduction.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  roofline     — the dominant kernel (fp16-MFMA proposal pass) against the dense MFMA peak, timed live with
                 HIP events recorded on the launch stream (libvqhip's vqhip_profile_* hooks)
  cpu_baseline — the reference's own ATen composition (oracle/torch_ref.py) timed on this box's host cores
                 on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

K_CODES, DIM, TOK_PER_IMAGE = 16384, 256, 256
MFMA_F16_DENSE_PEAK_TFLOPS = 2500.0       # MI355X_MICROARCH.md: ~2.5 PF dense bf16/fp16
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--images', type=int, default=2048, help='images per GPU per step (256 tokens each)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    return ap.parse_args()


def cpu_baseline(n_tokens: int = 2048, reps: int = 4):
    """The reference path (torch.cdist -> argmin -> embedding -> losses -> STE) on the host cores."""
    import numpy as np
    import torch

    from oracle import synth, torch_ref as tr

    x = torch.from_numpy(synth.bf16_round(synth.normal(3407, n_tokens, DIM)))
    w = torch.from_numpy(synth.normal(3408, K_CODES, DIM))
    ncpu = os.cpu_count() or 1
    best = None
    # ATen's CPU GEMM does not always scale to every hardware thread: keep the fastest of a few thread counts
    for threads in sorted({ncpu, max(1, ncpu // 2), min(ncpu, 64), min(ncpu, 32)}, reverse=True):
        torch.set_num_threads(threads)
        times = []
        with torch.no_grad():
            for i in range(reps + 1):
                t0 = time.perf_counter()
                out = tr.forward(x, w, 'L2', 'vqgan')
                float(out['loss'])
                t1 = time.perf_counter()
                if i > 0:
                    times.append(t1 - t0)
        med_t = float(np.median(times))
        if best is None or med_t < best[0]:
            best = (med_t, threads)
    med, used = best
    torch.set_num_threads(used)
    return {
        'value': n_tokens / med, 'unit': 'tokens/s', 'cores': torch.get_num_threads(), 'kind': 'port',
        'sample': f'{reps} timed forwards (median) of the reference ATen path (torch.cdist+argmin+embedding+'
                  f'VQGAN loss+STE, fp32) over {n_tokens} tokens, K={K_CODES}, D={DIM}',
        'ms_per_forward': med * 1e3,
    }


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus and world > 1:
        args.gpus = world
    assert torch.cuda.is_available(), 'bench.py needs MI355X GPUs (no CPU path)'
    # VQ_BENCH_SHARE_GPU=1 is a plumbing check for boxes with fewer GPUs than ranks: every rank uses cuda:0 and the
    # two timing collectives run over gloo (RCCL refuses two ranks on one device).  Never set by the driver.
    share_gpu = os.environ.get('VQ_BENCH_SHARE_GPU') == '1'
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    distributed = world > 1 or 'TORCHELASTIC_RUN_ID' in os.environ      # launched by torch.distributed.run
    if distributed:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if share_gpu:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)

    from vector_quantization_amd import _lib, ops

    N = args.images * TOK_PER_IMAGE
    g = torch.Generator(device=dev).manual_seed(3407 + rank)
    w = torch.randn(K_CODES, DIM, device=dev, generator=g)                    # random-init codebook (fp32)
    x = torch.randn(N, DIM, device=dev, generator=g).bfloat16()               # synthetic latents, bf16 (autocast)
    hist = torch.zeros(K_CODES, dtype=torch.int32, device=dev)

    def step():
        cb = ops.prepare_codebook(w, 'L2')                 # weight may change every step in training: re-prepared
        idx = ops.argmin(x, cb)
        z, z_ste, sse = ops.gather_ste_loss(x, w, idx, need_z=False)
        return idx, z_ste, sse

    def barrier():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ops.hist(step()[0], K_CODES, out=hist)                # code-usage statistics outside the timed region
    L = _lib.lib()
    if 'VQHIP_TUNE_SLICES' in os.environ:                   # A/B knob (results unchanged): codebook slices
        L.vqhip_set_tuning(2, int(os.environ['VQHIP_TUNE_SLICES']))
    barrier()
    L.vqhip_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        idx, z_ste, sse = step()
    barrier()
    t1 = time.perf_counter()
    ms_sum, launches = ctypes.c_double(0), ctypes.c_int64(0)
    _lib.check(L.vqhip_profile_collect(ctypes.byref(ms_sum), ctypes.byref(launches)), 'vqhip_profile_collect')
    L.vqhip_profile_enable(0)

    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device='cpu' if share_gpu else dev)
    if distributed:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    loss = 1.25 * float(sse.item()) / (N * DIM)

    if rank == 0:
        tokens = N * world * args.steps
        kern_ms = ms_sum.value / max(1, launches.value)
        flops = 2.0 * N * K_CODES * DIM                                 # 2*K*D per token
        achieved_tf = flops / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else 0.0
        alg_bytes = N * (DIM * 2 + 8 + DIM * 4) + K_CODES * DIM * 4     # §8(d): full forward, bf16 x, + codebook once
        traffic = None
        pmc = os.path.join(ROOT, 'profiles', 'pmc_latest.json')
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                if int(rec.get('tokens_per_launch', -1)) == N:      # counters are per launch of THIS workload
                    traffic = rec.get('coarse_kernel_hbm_bytes_per_launch')
            except Exception:
                traffic = None
        out = {
            'metric': 'quantized tokens/sec, VQGAN quantizer forward K=16384 D=256',
            'value': tokens / elapsed, 'unit': 'tokens/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f16+f32',
            'dtype_note': 'f16 MFMA (f32 accumulate) proposes candidates under a rigorous bound; the decision is exact f32',
            'data': 'synthetic',
            'config': {'workload': 'VQGAN K=16384 D=256, 256x256 images -> 16x16 tokens, bf16 latents, '
                                   'full quantizer forward (prepare+argmin+gather+STE+loss)',
                       'images_per_gpu': args.images, 'tokens_per_gpu_per_step': N, 'codebook': [K_CODES, DIM],
                       'parallelism': f'dp{world} (independent shards, no data-path collective)'},
            'roofline': {'bound': 'mfma', 'kernel': 'coarse_kernel (fp16 MFMA distance+argmin proposals)',
                         'achieved': achieved_tf, 'peak': MFMA_F16_DENSE_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': achieved_tf / MFMA_F16_DENSE_PEAK_TFLOPS, 'traffic': traffic,
                         'kernel_ms': kern_ms, 'launches_timed': launches.value,
                         'hbm_frac_algorithmic': alg_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS},
            'loss': loss, 'used_codes': int((hist > 0).sum().item()),
        }
        if not args.no_cpu_baseline and world == 1:          # reported once, at N=1 (rank 0's host cores)
            out['cpu_baseline'] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
