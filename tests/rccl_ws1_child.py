"""Child process of tests/test_gpu_rccl.py — started by `python -m torch.distributed.run --nproc-per-node=1` BEFORE anything
touches the GPU, so that `init_process_group('nccl', device_id=...)`, the packed device-tensor all-reduce and HIP-graph
capture of a step that contains the collective really execute on RCCL, on the one GPU a test box has.

1. No process group: the CVQ-VAE and VQ-KD training steps through the nn.Modules (one-rank flow: no pack, no collective).
2. `nccl` group of world size 1 with VQ_FORCE_EXCHANGE=1: the same steps through the multi-rank flow (pack -> all-reduce ->
   apply), once with the collective issued by torch.distributed (ProcessGroupNCCL's stream) and once by libvqhip on the
   compute stream (vqhip_allreduce_packed, communicator bootstrapped through the torch store).
3. The step replayed from HIP graphs (graphs.GraphedQuantizer) with the collective inside the capture, both routes.
Codebooks, probabilities and tokens must equal the no-process-group result bit for bit.  Writes one JSON record."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

EMB = 'torch_nn_modules_sparse_Embedding'


def cfg(kind, K, D):
    emb = dict(type=EMB, num_embeddings=K, embedding_dim=D)
    if kind == 'vqkd':       # configs/vqkd/model.py:20-26
        return dict(type='VQKDQuantizer', embedding=emb, distance=dict(type='CosineDistance'),
                    callbacks=[dict(type='VQKDCallback', ema=dict())],
                    losses=dict(commitment_loss=dict(type='CommitmentLoss', mse=dict(norm=True))))
    # 'cvqsync': NearestAnchor(sync=True), the cluster config's anchor (configs/cluster/model.py:28) — the key exchange
    return dict(type='VQGANQuantizer', embedding=emb, distance=dict(type='CosineDistance'),      # configs/cvqvae/quantizer.py
                callbacks=[dict(type='CVQVAECallback', ema=dict(), anchor=dict(type='NearestAnchor', sync=kind == 'cvqsync'))],
                losses=dict(vqgan_loss=dict(type='VQGANLoss')))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', required=True)
    ap.add_argument('--steps', type=int, default=4)
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    from oracle import synth
    from vector_quantization_amd import Config, build_quantizer, rccl
    from vector_quantization_amd.graphs import GraphedQuantizer
    from vector_quantization_amd.utils import exchange_log

    # the one-rank flow must itself be reproducible to be compared bit for bit: VQ-KD's centroid sums take the ordered route
    # (fp32 atomics add in arrival order: include/vqhip.h, "deterministic (ordered) codebook-side sums")
    os.environ['VQHIP_ORDERED'] = '1'
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    N, K, D = 3000, 4096, 64
    gen = synth.rng(404)
    w0 = synth.unit_rows(gen.standard_normal((K, D), dtype=np.float32))
    batches = [torch.from_numpy(gen.standard_normal((N, D), dtype=np.float32) * np.float32(0.3) + w0[gen.integers(0, K // 8, N)]).to(dev)
               for _ in range(args.steps)]

    def build(kind):
        torch.manual_seed(0)
        q = build_quantizer(cfg(kind, K, D))
        q.train()
        q.init_weights(Config(dict(type='vqgan') if kind.startswith('cvq') else {}))
        q = q.to(dev)
        q._forward_pre_hooks.clear()               # start from the given codebook: no lazy k-means init
        with torch.no_grad():
            q.embedding.weight.copy_(torch.from_numpy(w0))
        return q

    def run(kind, graphed=False):
        """`steps` training forwards; returns the tokens of every step, the final codebook (+ probabilities) and the log."""
        q = build(kind)
        call = (lambda x: q(x, {}))
        if graphed:
            gq = GraphedQuantizer(q, batches[0])
            call = (lambda x: gq(x))
        quants, calls, nbytes = [], 0, 0
        for x in batches:
            exchange_log.start()
            out = call(x.clone().requires_grad_(True))
            st = exchange_log.stop()
            calls += st['calls']
            nbytes += st['bytes']
            quants.append((out[2] if graphed else out[2]['quant']).detach().clone().cpu().numpy())
        torch.cuda.synchronize()
        state = {'w': q.embedding.weight.detach().cpu().numpy(), 'quant': np.stack(quants)}
        if kind.startswith('cvq'):
            state['p'] = q.get_buffer('_probability').cpu().numpy()
        return state, calls, nbytes

    def same(a, b):
        return all(a[k].tobytes() == b[k].tobytes() for k in a)

    rec = {'N': N, 'K': K, 'D': D, 'steps': args.steps}
    os.environ['VQ_FORCE_EXCHANGE'] = '0'
    ref = {kind: run(kind)[0] for kind in ('cvq', 'vqkd', 'cvqsync')}   # no process group: the one-rank flow

    dist.init_process_group('nccl', device_id=dev)
    rec['backend'] = dist.get_backend()
    rec['world'] = dist.get_world_size()
    ones = torch.ones(1, dtype=torch.int64, device=dev)
    dist.all_reduce(ones)
    rec['rccl_ranks'] = int(ones.item())
    os.environ['VQ_FORCE_EXCHANGE'] = '1'
    for route in ('torch', 'direct'):
        os.environ['VQHIP_ALLREDUCE'] = route
        for kind in ('cvq', 'vqkd', 'cvqsync'):
            got, calls, nbytes = run(kind)
            rec[f'{kind}_{route}_bit_identical'] = same(ref[kind], got)
            rec[f'{kind}_{route}_collectives_per_step'] = calls / args.steps
            rec[f'{kind}_{route}_bytes'] = nbytes
        rec[f'status_{route}'] = rccl.status()
    # the step replayed from HIP graphs, the collective inside the capture.  Graph replay of the CVQ step sizes its launches
    # for K listed codes; the results are still those of the eager step bit for bit.
    for route in ('direct', 'torch'):
        os.environ['VQHIP_ALLREDUCE'] = route
        for kind in ('cvq', 'vqkd', 'cvqsync'):
            try:
                got, calls, _ = run(kind, graphed=True)
                rec[f'{kind}_{route}_graphed_bit_identical'] = same(ref[kind], got)
                rec[f'{kind}_{route}_graphed_error'] = None
            except Exception as exc:        # noqa: BLE001 — recorded: which route can be captured is what this run finds out
                rec[f'{kind}_{route}_graphed_bit_identical'] = None
                rec[f'{kind}_{route}_graphed_error'] = f'{type(exc).__name__}: {exc}'[:500]
                torch.cuda.synchronize()
    # ---- the quantizer inside DDP / FSDP on the nccl backend (configs/strategies/ddp.py:5-6, fsdp.py:5-8): torch's communicator
    # and the library's own alive in ONE process, the packed exchange on the compute stream between DDP's bucket all-reduces
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from torch.nn.parallel import DistributedDataParallel as DDP
    from toy_model import build_toy, train_steps
    C, B, HW = 8, 6, 8
    w32 = torch.from_numpy(synth.unit_rows(gen.standard_normal((1024, 32), dtype=np.float32)))
    # two optimizer steps: the second sees the first one's codebook update and SGD step.  (More steps compare a chaotic system:
    # the convolutions' weight gradients use float atomics, a 1e-8 difference in a latent flips a NearestAnchor near-tie a few
    # steps later, and the re-anchored code moves by O(1) — the bare model differs from ITSELF that way, tools/debug/ddp_graphed.py)
    images = [torch.from_numpy(gen.standard_normal((B, C, HW, HW), dtype=np.float32)).to(dev) for _ in range(2)]

    def params_of(model):
        return {n: p.detach().clone() for n, p in model.named_parameters()}

    for route in ('direct', 'torch'):
        os.environ['VQHIP_ALLREDUCE'] = route
        for kind in ('cvq', 'vqkd'):
            bare = build_toy(kind, 1024, 32, w32, dev)
            r_bare = train_steps(bare, images, autocast=True)
            wrapped = build_toy(kind, 1024, 32, w32, dev)
            ddp = DDP(wrapped, device_ids=[local], find_unused_parameters=True)
            r_ddp = train_steps(ddp, images, autocast=True)
            # (identical up to the arrival order of the float atomics in the convolutions' weight gradients)
            ok = all(torch.equal(a[1], b[1]) and abs(float(a[0]) - float(b[0])) <= 1e-6 for a, b in zip(r_bare, r_ddp))
            pa, pb = params_of(bare), params_of(wrapped)
            rec[f'ddp_{kind}_{route}_bit_identical'] = bool(ok and all(float((pa[n] - pb[n]).abs().max()) <= 1e-6 for n in pa))
        rec[f'ddp_status_{route}'] = rccl.status()
    # the graphed quantizer inside the DDP-wrapped model (forward + backward replayed, the collective inside the capture)
    os.environ['VQHIP_ALLREDUCE'] = 'direct'
    try:
        bare = build_toy('cvq', 1024, 32, w32, dev)
        r_bare = train_steps(bare, images)
        wrapped = build_toy('cvq', 1024, 32, w32, dev)
        sample = torch.zeros(B * HW * HW, 32, device=dev)
        gq = GraphedQuantizer(wrapped._quantizer, sample)
        wrapped.quant_call = gq
        ddp = DDP(wrapped, device_ids=[local], find_unused_parameters=True)
        r_ddp = train_steps(ddp, images)
        rec['ddp_graphed_tokens_identical'] = all(torch.equal(a[1].reshape(-1), b[1].reshape(-1)) for a, b in zip(r_bare, r_ddp))
        pa, pb = params_of(bare), params_of(wrapped)
        rec['ddp_graphed_max_param_diff'] = max(float((pa[n] - pb[n]).abs().max()) for n in pa)
        rec['ddp_graphed_error'] = None
    except Exception as exc:        # noqa: BLE001
        rec['ddp_graphed_error'] = f'{type(exc).__name__}: {exc}'[:500]
        torch.cuda.synchronize()
    # FSDP (use_orig_params=True, as configs/strategies/fsdp.py:5-8): construction + one optimizer step; the codebook update
    # must survive FSDP's next unshard (the callbacks write FSDP-managed parameters in place)
    try:
        from torch.distributed.fsdp import FullyShardedDataParallel as FSDP
        os.environ['VQHIP_ALLREDUCE'] = 'torch'
        bare = build_toy('cvq', 1024, 32, w32, dev)
        r_bare = train_steps(bare, images)
        inner = build_toy('cvq', 1024, 32, w32, dev)
        fs = FSDP(inner, use_orig_params=True, device_id=dev)
        r_fs = train_steps(fs, images)
        rec['fsdp_tokens_identical'] = all(torch.equal(a[1].reshape(-1), b[1].reshape(-1)) for a, b in zip(r_bare, r_fs))
        rec['fsdp_loss_diff'] = max(abs(float(a[0]) - float(b[0])) for a, b in zip(r_bare, r_fs))
        rec['fsdp_error'] = None
    except Exception as exc:        # noqa: BLE001
        rec['fsdp_error'] = f'{type(exc).__name__}: {exc}'[:500]
        torch.cuda.synchronize()
    with open(args.out, 'w') as fh:
        json.dump(rec, fh, indent=1)
    print(json.dumps(rec))
    rccl.shutdown()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
