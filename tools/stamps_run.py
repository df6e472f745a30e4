import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vector_quantization_amd import ops
N, K, D, metric = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
g = torch.Generator(device='cuda').manual_seed(3407)
w = torch.randn(K, D, device='cuda', generator=g); x = torch.randn(N, D, device='cuda', generator=g)
if metric == 'Cosine': x = ops.normalize_rows(x)
cb = ops.prepare_codebook(w, metric)
idx = ops.argmin(x, cb); torch.cuda.synchronize()
