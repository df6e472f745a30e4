#!/usr/bin/env python3
"""BASELINE.json configs[4] in miniature: data-sharded bulk tokenization.  Every rank encodes its own shard of latent
maps (synthetic here; the reference's encoder is out of scope) with the HIP quantizer path, writes the reference's
token files (runners/callbacks.py:40-53: tokens/{iter}_{rank}.pth) and accumulates the code histogram; the only
collective is the all-reduce of the histogram at summary time (runners/metrics.py:46-56).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/tokenize_synthetic.py \\
        --work-dir /tmp/tok --iters 4 --images 2048
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--work-dir', default='/tmp/vq_tokens')
    ap.add_argument('--iters', type=int, default=4)
    ap.add_argument('--images', type=int, default=2048, help='images per iteration over ALL ranks (reference batch 2048)')
    ap.add_argument('--codes', type=int, default=16384)
    ap.add_argument('--dim', type=int, default=256)
    args = ap.parse_args(argv)
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (('RANK', 0), ('WORLD_SIZE', 1), ('LOCAL_RANK', 0)))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29500')
        dist.init_process_group('nccl', device_id=dev)
    from vector_quantization_amd import build_quantizer, Config, tokenization as T
    K, D = args.codes, args.dim
    q = build_quantizer(dict(type='VQGANQuantizer',
                             embedding=dict(type='torch_nn_modules_sparse_Embedding', num_embeddings=K, embedding_dim=D),
                             distance=dict(type='L2Distance'), losses=dict(vqgan_loss=dict(type='VQGANLoss')),
                             cache_codebook=True))              # eval: the prepared codebook is reused across batches
    q.init_weights(Config(type='vqgan'))
    q = q.to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(3407)           # same codebook on every rank
    with torch.no_grad():
        q.embedding.weight.copy_(torch.randn(K, D, device=dev, generator=g))
    per_rank = args.images // world                              # DistributedSampler-style contiguous shards
    counts = T.CodebookCounts(K)
    gx = torch.Generator(device=dev).manual_seed(1000 + rank)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad():
        for it in range(1, args.iters + 1):
            x = torch.randn(per_rank, D, 16, 16, device=dev, generator=gx).bfloat16().contiguous(memory_format=torch.channels_last)
            quant, memo = T.encode_to_quant(q, x, {})
            counts.update(quant)
            ids = [f'{it}_{rank}_{i}' for i in range(per_rank)]
            T.save_tokens(args.work_dir, it, ids, torch.zeros(per_rank, dtype=torch.long), quant, memo['quantizer']['x_shape'], rank=rank)
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    summary = counts.summary()                                   # all-reduce of the histogram happens here
    if rank == 0:
        toks = args.iters * per_rank * world * 256
        print(f'ranks={world} iters={args.iters} images/iter={per_rank * world}: {toks / t / 1e6:.1f} M tokens/s including '
              f'torch.save; codebook_usage={summary["codebook_usage"]:.4f} codebook_ppl={summary["codebook_ppl"]:.4f}')
    if world > 1:
        dist.destroy_process_group()
    return summary


if __name__ == '__main__':
    main()
