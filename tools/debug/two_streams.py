"""Does splitting the headline batch in two halves on two streams overlap the bandwidth-bound front / tail of one half with the
MFMA-bound proposal pass of the other?  Same total work; identical tokens required."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

dev = torch.device('cuda', 0)
K, D = 16384, 256
N = 524288
g = torch.Generator(device=dev).manual_seed(3407)
w = torch.randn(K, D, device=dev, generator=g)
x = torch.randn(N, D, device=dev, generator=g).bfloat16()
q = bench.build_module(bench.quantizer_cfg(K, D, 'L2'), dev, w, train=False)
halves = [x[: N // 2].contiguous(), x[N // 2:].contiguous()]
quarters = [x[i * (N // 4):(i + 1) * (N // 4)].contiguous() for i in range(4)]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def full():
    with torch.no_grad():
        return q(x, {})

def split(parts):
    outs = []
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.no_grad():
        for i, p in enumerate(parts):
            with torch.cuda.stream(s1 if i % 2 == 0 else s2):
                outs.append(q(p, {}))
    cur.wait_stream(s1); cur.wait_stream(s2)
    return outs

def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

ref = full()[2]['quant'].reshape(-1)
o = split(halves)
got = torch.cat([a[2]['quant'].reshape(-1) for a in o])
print('same tokens (halves):', bool(torch.equal(ref, got)))
for rnd in range(3):
    print(f'round {rnd}: full batch {timeit(full):.4f} ms   two halves on two streams {timeit(lambda: split(halves)):.4f} ms   four quarters on two streams {timeit(lambda: split(quarters)):.4f} ms')
