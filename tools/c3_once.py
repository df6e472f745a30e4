import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vector_quantization_amd import ops
N, K, D = 100352, 8192, 32
g = torch.Generator(device='cuda').manual_seed(3407)
w = torch.randn(K, D, device='cuda', generator=g); x = torch.randn(N, D, device='cuda', generator=g)
for _ in range(30):
    idx = ops.encode(x, w, 'Cosine')[0]
torch.cuda.synchronize()
