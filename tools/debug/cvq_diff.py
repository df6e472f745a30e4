"""Why does bench.py --workload cvq report another step time than tools/bench_train_shapes.py cvq?  Same module, both timing loops."""
import ctypes, os, sys, time
_libc = ctypes.CDLL("libc.so.6")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from vector_quantization_amd import _lib

dev = torch.device('cuda', 0)
K, D, cfg = bench.train_cfg('cvq')
tokens = 3072
w = torch.nn.functional.normalize(torch.randn(K, D, device=dev, generator=torch.Generator(device=dev).manual_seed(3407)))
g = torch.Generator(device=dev).manual_seed(3407)
npool = int(os.environ.get('POOL', '128'))
pool = [(w[torch.randint(0, K, (tokens,), device=dev, generator=g)] + 0.05 * torch.randn(tokens, D, device=dev, generator=g)).requires_grad_(True)
        for _ in range(npool)]
gz = torch.randn(tokens, D, device=dev, generator=g) / (tokens * D)
q = bench.build_train_module('cvq', cfg, dev, w)
params = [p for p in q.parameters() if p.requires_grad]
turn = [0]
keep = os.environ.get('KEEP', '1') == '1'

def step():
    xin = pool[turn[0] % npool]
    turn[0] += 1
    for p_ in params:
        p_.grad = None
    xin.grad = None
    if keep:
        z, loss, extra = q(xin, {})
    else:
        z, loss = q(xin, {})[:2]
        extra = None
    torch.autograd.backward([loss, z], [None, gz])
    return z, loss, extra

for _ in range(150):
    step()
for nsteps in ():
    res = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            out = step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        res.append(((t2 - t0) / nsteps * 1e3, (t1 - t0) / nsteps * 1e3))
    print(f'blocks of {nsteps:4d} steps:', ' '.join(f'{a:.4f}/{b:.4f}' for a, b in res), ' listed', q._callbacks.callbacks[0].last_exchange_rows, flush=True)

# ---- where the host's time goes, block by block
from vector_quantization_amd import train_step
from vector_quantization_amd.quantizers import callbacks as cbm
acc = {'call': 0.0, 'refresh': 0, 'fwd': 0.0, 'bwd': 0.0}
orig_call = train_step.cvq_forward
def timed_call(*a, **k):
    t = time.perf_counter(); r = orig_call(*a, **k); acc['call'] += time.perf_counter() - t; return r
train_step.cvq_forward = timed_call
cb = q._callbacks.callbacks[0]
orig_refresh = cb.refresh_list
def counted_refresh():
    acc['refresh'] += 1; return orig_refresh()
cb.refresh_list = counted_refresh

def step2():
    xin = pool[turn[0] % npool]
    turn[0] += 1
    for p_ in params:
        p_.grad = None
    xin.grad = None
    t = time.perf_counter()
    z, loss = q(xin, {})[:2]
    t1 = time.perf_counter()
    torch.autograd.backward([loss, z], [None, gz])
    acc['fwd'] += t1 - t; acc['bwd'] += time.perf_counter() - t1

for blk in range(int(os.environ.get('BLOCKS', '40'))):
    for k_ in acc: acc[k_] = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        step2()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'block {blk:2d}: {(t2 - t0) * 10:.4f} ms/step  host {(t1 - t0) * 10:.4f}  forward {acc["fwd"] * 10:.4f} (library call {acc["call"] * 10:.4f})  '
          f'backward {acc["bwd"] * 10:.4f}  cpu {_libc.sched_getcpu()}', flush=True)
