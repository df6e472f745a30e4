#!/bin/bash
# timing-only builds of the D = 32 proposal kernel (16x16x32 form, key 11 = 0): no DMA requests in the loop / no stage body
cd /root/repo
python - <<'PY'
import ctypes, os, subprocess, sys
for lib in ('shipped', 'build/exp/libvqhip_nodma.so', 'build/exp/libvqhip_nobody.so'):
    env = dict(os.environ)
    if lib != 'shipped': env['VQHIP_LIB'] = os.path.join('/root/repo', lib)
    code = r'''
import ctypes, sys, time
sys.path.insert(0, '/root/repo')
import torch
from vector_quantization_amd import _lib, ops
L = _lib.lib(); L.vqhip_set_tuning(11, 0)
for N in (100352, 65536):
    g = torch.Generator(device='cuda').manual_seed(3407)
    w = torch.randn(8192, 32, device='cuda', generator=g); x = ops.normalize_rows(torch.randn(N, 32, device='cuda', generator=g))
    cb = ops.prepare_codebook(w, 'Cosine')
    for _ in range(5): ops.argmin(x, cb)
    torch.cuda.synchronize(); L.vqhip_profile_enable(1)
    for _ in range(20): ops.argmin(x, cb)
    torch.cuda.synchronize()
    ms, n = ctypes.c_double(0), ctypes.c_int64(0); L.vqhip_profile_collect(ctypes.byref(ms), ctypes.byref(n))
    print(N, 'proposal kernel us', round(ms.value / n.value * 1e3, 1))
'''
    out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True)
    print(lib, out.stdout.strip().replace('\n', ' | '), out.stderr[-200:] if out.returncode else '')
PY
