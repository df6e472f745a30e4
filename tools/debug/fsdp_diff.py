#!/usr/bin/env python3
"""torchrun --nproc-per-node=1: FSDP(use_orig_params=True) toy model vs the bare one, parameter by parameter after every step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, torch.distributed as dist
from torch.distributed.fsdp import FullyShardedDataParallel as FSDP
from oracle import synth
from toy_model import build_toy

local = int(os.environ.get('LOCAL_RANK', '0'))
torch.cuda.set_device(local); dev = torch.device('cuda', local)
dist.init_process_group('nccl', device_id=dev)
gen = synth.rng(404)
w32 = torch.from_numpy(synth.unit_rows(gen.standard_normal((1024, 32), dtype=np.float32)))
C, B, HW = 8, 6, 8
images = [torch.from_numpy(gen.standard_normal((B, C, HW, HW), dtype=np.float32)).to(dev) for _ in range(3)]

def run(model, named, label):
    opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=0.05)
    out = []
    for t, image in enumerate(images):
        opt.zero_grad(set_to_none=True)
        o, ql, quant = model(image)
        loss = o.float().pow(2).mean() + ql
        loss.backward()
        grads = named(model, grad=True)
        opt.step()
        out.append((float(loss), quant.clone(), named(model), grads))
    return out

def named_bare(m, grad=False):
    return {n: (p.grad if grad else p).detach().clone() for n, p in m.named_parameters() if (not grad or p.grad is not None)}
def named_fsdp(m, grad=False):
    with FSDP.summon_full_params(m, with_grads=grad):
        return {n.replace('_fsdp_wrapped_module.', ''): (p.grad if grad else p).detach().clone() for n, p in m.named_parameters()
                if (not grad or p.grad is not None)}

for kind in sys.argv[1:] or ['vqgan', 'cvq']:
    for one_call in ((True, False) if kind == 'cvq' else (True,)):
        bare = build_toy(kind, 1024, 32, w32, dev); bare._quantizer.one_call_steps = one_call
        rb = run(bare, named_bare, 'bare')
        inner = build_toy(kind, 1024, 32, w32, dev); inner._quantizer.one_call_steps = one_call
        fs = FSDP(inner, use_orig_params=True, device_id=dev)
        rf = run(fs, named_fsdp, 'fsdp')
        for t in range(3):
            print(kind, 'one_call' if one_call else 'hooks', 'step', t, 'loss', rb[t][0], rf[t][0], 'token mismatches', int((rb[t][1].reshape(-1) != rf[t][1].reshape(-1)).sum()))
            print('   params', {n: float((rb[t][2][n] - rf[t][2][n]).abs().max()) for n in rb[t][2]})
            print('   grads ', {n: (float((rb[t][3][n] - rf[t][3][n]).abs().max()) if n in rf[t][3] else 'missing') for n in rb[t][3]})
dist.destroy_process_group()
