#!/bin/bash
cd /root/repo
O=gpurun_out/r02_av; mkdir -p $O
timeout 500 python tools/fuzz_vs_exact.py 200 223 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -3 $O/fuzz.log | cut -c1-300
python - <<'PY'
import time, torch
from vector_quantization_amd import ops
N, K, D = 100352, 8192, 32
g = torch.Generator(device='cuda').manual_seed(3407)
w = torch.randn(K, D, device='cuda', generator=g); x = torch.randn(N, D, device='cuda', generator=g)
for m in ('Cosine', 'CosineBF16'):
    for _ in range(5): idx, st = None, ops.encode(x, w, m)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): ops.encode(x, w, m)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 30
    xq = ops.normalize_rows(x); 
    if m == 'CosineBF16': xq = xq.bfloat16().float()
    _, stats = ops.argmin(xq, ops.prepare_codebook(w, m), return_stats=True)
    print(m, f'{t*1e3:.3f} ms  {N/t/1e6:.1f} M tok/s  rescan={int(stats[0])} multi={int(stats[1])} exact={int(stats[2])}')
PY
