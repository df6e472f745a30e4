#!/usr/bin/env python3
"""torchrun --nproc-per-node=1: bare toy model vs GraphedQuantizer inside it (no DDP / DDP) vs FSDP, step by step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel as DDP
from oracle import synth
from toy_model import build_toy, train_steps
from vector_quantization_amd.graphs import GraphedQuantizer

local = int(os.environ.get('LOCAL_RANK', '0'))
torch.cuda.set_device(local); dev = torch.device('cuda', local)
dist.init_process_group('nccl', device_id=dev)
os.environ['VQ_FORCE_EXCHANGE'] = os.environ.get('VQ_FORCE_EXCHANGE', '1')
os.environ['VQHIP_ALLREDUCE'] = os.environ.get('VQHIP_ALLREDUCE', 'direct')
gen = synth.rng(404)
w0 = synth.unit_rows(gen.standard_normal((4096, 64), dtype=np.float32))
if os.environ.get('VQ_DBG_SLICED'):
    w32 = torch.from_numpy(w0[:1024, :32].copy())
else:
    w32 = torch.from_numpy(synth.unit_rows(gen.standard_normal((1024, 32), dtype=np.float32)))
C, B, HW = 8, 6, 8
images = [torch.from_numpy(gen.standard_normal((B, C, HW, HW), dtype=np.float32)).to(dev) for _ in range(3)]
def params_of(m): return {n: p.detach().clone() for n, p in m.named_parameters()}
def report(name, ra, rb, pa, pb):
    for t, (a, b) in enumerate(zip(ra, rb)):
        print(name, 'step', t, 'token mismatches', int((a[1].reshape(-1) != b[1].reshape(-1)).sum()), 'loss', float(a[0]), float(b[0]))
    print(name, 'param diffs', {n.replace('module.', ''): float((pa[n] - pb[n]).abs().max()) for n in pa})
bare = build_toy('cvq', 1024, 32, w32, dev); r_bare = train_steps(bare, images); p_bare = params_of(bare)
bare2 = build_toy('cvq', 1024, 32, w32, dev); r2 = train_steps(bare2, images); report('bare vs bare', r_bare, r2, p_bare, params_of(bare2))
g = build_toy('cvq', 1024, 32, w32, dev)
g.quant_call = GraphedQuantizer(g._quantizer, torch.zeros(B * HW * HW, 32, device=dev))
rg = train_steps(g, images); report('graphed, no DDP', r_bare, rg, p_bare, params_of(g))
g2 = build_toy('cvq', 1024, 32, w32, dev)
g2.quant_call = GraphedQuantizer(g2._quantizer, torch.randn(B * HW * HW, 32, device=dev))
rg2 = train_steps(g2, images); report('graphed (random sample), no DDP', r_bare, rg2, p_bare, params_of(g2))
g3 = build_toy('cvq', 1024, 32, w32, dev)
g3.quant_call = GraphedQuantizer(g3._quantizer, torch.randn(B * HW * HW, 32, device=dev))
d3 = DDP(g3, device_ids=[local], find_unused_parameters=True)
rg3 = train_steps(d3, images); report('graphed, DDP', r_bare, rg3, p_bare, params_of(g3))
from torch.distributed.fsdp import FullyShardedDataParallel as FSDP
os.environ['VQHIP_ALLREDUCE'] = 'torch'
inner = build_toy('cvq', 1024, 32, w32, dev)
fs = FSDP(inner, use_orig_params=True, device_id=dev)
print('fsdp flattened flag on the codebook:', getattr(inner._quantizer.embedding.weight, '_fsdp_flattened', None), type(inner._quantizer.embedding.weight))
rf = train_steps(fs, images)
report('fsdp', r_bare, rf, p_bare, {n.replace('_fsdp_wrapped_module.', ''): p.detach().clone() for n, p in fs.named_parameters()} if False else p_bare)
dist.destroy_process_group()
