import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vector_quantization_amd import ops
for seed in range(40):
    g = torch.Generator(device='cuda').manual_seed(seed)
    N, K, D = 280, 8192, 32
    w = torch.randn(K, D, device='cuda', generator=g); x = torch.randn(N, D, device='cuda', generator=g)
    x = x * torch.exp(2 * torch.randn(N, 1, device='cuda', generator=g)); w = w * torch.exp(torch.randn(K, 1, device='cuda', generator=g))
    x, w = x * 1e4, w * 1e4
    got = ops.argmin(x, ops.prepare_codebook(w, 'L2')); ref = ops.argmin_exact(x, w, 'L2')
    torch.cuda.synchronize()
    print(seed, int((got != ref).sum()), flush=True)
