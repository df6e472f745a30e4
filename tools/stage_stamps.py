#!/usr/bin/env python3
"""What one iteration of the proposal kernel's stage loop spends its time on: a diagnostic build stamps s_memtime (shader clock) at
five points of every iteration of every wave (tools/build_exp.sh stage -DVQ_STAGE_STAMPS):

    VQHIP_LIB=build/exp/libvqhip_stage.so python tools/stage_stamps.py N K D [L2|Cosine] [bf16|fp32]

top of the iteration | the wave's share of the next stage requested (LDS-DMA issue) | stage computed (MFMAs + epilogue) |
own memory operations drained (s_waitcnt) | barrier passed.  Printed: per-phase shader cycles, median over the first 64 workgroups
and the iterations 2 .. last-2, for the leading waves (0-3) and the lagging waves (4-7, one stage behind), and one workgroup's
full trace for wave 0."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vector_quantization_amd import _lib, ops

N, K, D = (int(a) for a in sys.argv[1:4])
metric = sys.argv[4] if len(sys.argv) > 4 else 'L2'
dt = sys.argv[5] if len(sys.argv) > 5 else 'fp32'
g = torch.Generator(device='cuda').manual_seed(3407)
w = torch.randn(K, D, device='cuda', generator=g)
if metric != 'L2':
    w = torch.nn.functional.normalize(w)
x = w[torch.randint(0, K, (N,), device='cuda', generator=g)] + 0.05 * torch.randn(N, D, device='cuda', generator=g)
if dt == 'bf16':
    x = x.bfloat16()
L = _lib.lib()
for kv in filter(None, os.environ.get('VQHIP_TUNE', '').split(',')):
    k, v = kv.split('='); L.vqhip_set_tuning(int(k), int(v))
fn = getattr(ctypes.CDLL(_lib.LIB_PATH), 'vqhip_debug_stage_stamps', None)
if fn is None:
    sys.exit('this library has no stage stamps: tools/build_exp.sh stage -DVQ_STAGE_STAMPS, then VQHIP_LIB=build/exp/libvqhip_stage.so')
cb = ops.prepare_codebook(w, metric)
for _ in range(30):
    ops.argmin(x, cb)
torch.cuda.synchronize()
WGS, ITERS = 64, 24
buf = (ctypes.c_ulonglong * (WGS * 8 * ITERS * 5))()
fn.restype = ctypes.c_int
assert fn(buf, WGS) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(WGS, 8, ITERS, 5).astype(np.int64)
names = ['request next stage', 'compute the stage', 'drain own requests', 'barrier']
print(f'{N} x {K} x {D} {metric} {dt}')
for label, waves in (('waves 0-3 (leading)', range(0, 4)), ('waves 4-7 (one stage behind)', range(4, 8))):
    sel = a[:, list(waves)]                                  # [wg, wave, it, 5]
    full = (sel[..., 0] > 0) & (sel[..., 4] > 0)
    nit = int(full.any(axis=(0, 1)).sum())
    lo, hi = 2, max(3, nit - 2)
    d = np.diff(sel[:, :, lo:hi], axis=-1)                   # [wg, wave, it, 4]
    ok = full[:, :, lo:hi]
    tot = (sel[:, :, lo:hi, 4] - sel[:, :, lo:hi, 0])
    print(f'  {label}: {nit} computed iterations; cycles per iteration, median (10th-90th percentile) over workgroups, waves and iterations {lo}..{hi - 1}')
    for i, nm in enumerate(names):
        v = d[..., i][ok]
        print(f'    {nm:20s}: {np.median(v):7.0f} ({np.percentile(v, 10):.0f}-{np.percentile(v, 90):.0f})')
    v = tot[ok]
    print(f'    {"whole iteration":20s}: {np.median(v):7.0f} ({np.percentile(v, 10):.0f}-{np.percentile(v, 90):.0f})')
for wv in (0, 4):
    tr = a[0, wv]
    t0 = tr[0, 0]
    print(f'  workgroup 0, wave {wv}: iteration: cycles after the first stamp at [top, requested, computed, drained, barrier]')
    for it in range(ITERS):
        if tr[it, 0] == 0:
            break
        print(f'    {it:2d}: ' + ' '.join(f'{int(t - t0):7d}' if t else '      -' for t in tr[it]))
