import os, sys, time
sys.path.insert(0, '/root/repo')
import torch
from vector_quantization_amd import _lib, ops
L = _lib.lib()
N, K, D, metric = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
g = torch.Generator(device='cuda').manual_seed(3407)
w = torch.randn(K, D, device='cuda', generator=g); x = torch.randn(N, D, device='cuda', generator=g)
if metric == 'Cosine': x = ops.normalize_rows(x)
ref = None
for ns in (0, 2, 4, 8, 0, 4):
    L.vqhip_set_tuning(2, ns)
    for _ in range(5): idx, st = ops.argmin(x, ops.prepare_codebook(w, metric), return_stats=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): ops.argmin(x, ops.prepare_codebook(w, metric))
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 30
    ref = idx if ref is None else ref
    print(f'slices {ns}: encode {t*1e3:.4f} ms  rescan rows {int(st[0])} multi {int(st[1])}  same={torch.equal(idx, ref)}')
L.vqhip_set_tuning(2, 0)
