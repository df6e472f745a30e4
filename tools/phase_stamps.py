#!/usr/bin/env python3
"""Where the time of a SHORT proposal pass goes: a diagnostic build stamps s_memrealtime (100 MHz) at seven points of every
workgroup of coarse_kernel (tools/build_exp.sh phase -DVQ_PHASE_STAMPS):

    VQHIP_LIB=build/exp/libvqhip_phase.so python tools/phase_stamps.py N K D [L2|Cosine] [bf16|fp32]

entry | first stages requested (token fragments requested, margins computed) | first barrier passed (first stage landed) |
stream done | records stored and drained | ticket drawn | end (the workgroup that completes a token block also decides its rows).
Printed per point: microseconds after the FIRST workgroup's entry — minimum, median, maximum over the workgroups — next to the
kernel's duration by HIP events on the same launches."""
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
for kv in filter(None, os.environ.get('VQHIP_TUNE', '').split(',')):      # e.g. VQHIP_TUNE=17=0 (vqhip_set_tuning keys)
    k, v = kv.split('='); L.vqhip_set_tuning(int(k), int(v))
fn = getattr(ctypes.CDLL(_lib.LIB_PATH), 'vqhip_debug_phase_stamps', None)
if fn is None:
    sys.exit('this library has no phase stamps: tools/build_exp.sh phase -DVQ_PHASE_STAMPS, then VQHIP_LIB=build/exp/libvqhip_phase.so')
cb = ops.prepare_codebook(w, metric)
for _ in range(20):
    ops.argmin(x, cb)
torch.cuda.synchronize()
L.vqhip_profile_enable(1)
reps = 50
for _ in range(reps):
    ops.argmin(x, cb)
torch.cuda.synchronize()
ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
L.vqhip_profile_collect(ctypes.byref(ms), ctypes.byref(cnt))
L.vqhip_profile_enable(0)
slots = 4096
buf = (ctypes.c_ulonglong * (8 * slots))()
fn.restype = ctypes.c_int
assert fn(buf, slots) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(slots, 8).astype(np.int64)
a = a[a[:, 0] > 0]
# stale slots of earlier, larger launches: keep the workgroups whose entry lies within 1 ms of the latest entry
a = a[a[:, 0] > a[:, 0].max() - 100000]
t0 = a[:, 0].min()
names = ['entry', 'first stages requested', 'first barrier passed', 'stream done', 'records stored', 'ticket drawn', 'end']
print(f'{N} x {K} x {D} {metric} {dt}: {len(a)} workgroups; proposal kernel {ms.value / max(1, cnt.value) * 1e3:.1f} us by HIP events '
      f'(mean of {cnt.value} launches); {int(a[:, 7].sum())} workgroups decided a token block')
where = a[:, 5].copy() if a[:, 7].max() > 1 else None      # coarse32_kernel's build: slot 5 = XCC_ID << 32 | HW_ID, slot 7 = loop cycles
if where is not None:
    a[:, 5] = 0
for i, nm in enumerate(names):
    col = a[:, i]
    ok = col > 0
    if not ok.any():
        print(f'  {nm:24s}: not stamped (decision stage in its own launch)')
        continue
    us = (col[ok] - t0) / 100.0
    print(f'  {nm:24s}: min {us.min():6.2f}  median {np.median(us):6.2f}  max {us.max():6.2f} us after the first entry')
d = (a[:, 3] - a[:, 2]) / 100.0
print(f'  stream (first barrier -> done) per workgroup: min {d.min():.2f} median {np.median(d):.2f} max {d.max():.2f} us')
pr = (a[:, 2] - a[:, 0]) / 100.0
late = (a[:, 0] - t0) > 500                       # workgroups of the later rounds (they enter as CUs free up)
print(f'  prologue (entry -> first barrier passed) per workgroup: first round min {pr[~late].min():.2f} median {np.median(pr[~late]):.2f} max {pr[~late].max():.2f} us'
      + (f'; later rounds ({int(late.sum())} workgroups) min {pr[late].min():.2f} median {np.median(pr[late]):.2f} max {pr[late].max():.2f} us' if late.any() else ''))
if a[:, 7].max() > 1:            # coarse32_kernel's diagnostic build: slot 7 = shader cycles of the stream loop
    clk = a[:, 7] / np.maximum(d, 1e-9) / 1e3
    print(f'  shader clock inside the stream loop (s_memtime / s_memrealtime): min {clk.min():.2f} median {np.median(clk):.2f} max {clk.max():.2f} GHz; '
          f'loop cycles per workgroup: min {a[:, 7].min()} median {int(np.median(a[:, 7]))} max {a[:, 7].max()}')
    order = np.argsort(a[:, 0])
    late = a[order][:, 0] - t0
    print(f'  workgroups entering later than 5 us after the first: {int((late > 500).sum())} of {len(a)}; entry of the last one {late.max() / 100.0:.2f} us')
    e = (a[:, 6] - a[:, 3]) / 100.0
    print(f'  tail (stream done -> end: drain, requests, records) per workgroup: min {e.min():.2f} median {np.median(e):.2f} max {e.max():.2f} us')
    p = (a[:, 2] - a[:, 0]) / 100.0
    print(f'  prologue (entry -> first barrier passed) per workgroup: min {p.min():.2f} median {np.median(p):.2f} max {p.max():.2f} us')
    # which CU a workgroup ran on: (XCC, SE, SH, CU) from the hardware id; workgroups per CU and the stream time by that count
    hw = where & 0xFFFFFFFF
    xcc = (where >> 32) & 0xF
    cu = (xcc << 12) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF)
    ids, inv, cnt = np.unique(cu, return_inverse=True, return_counts=True)
    print(f'  {len(ids)} CUs used; workgroups per CU: ' + ', '.join(f'{int((cnt == k).sum())} CUs with {k}' for k in sorted(set(cnt.tolist()))))
    for k in sorted(set(cnt.tolist())):
        sel = cnt[inv] == k
        print(f'    workgroups on CUs with {k}: stream min {d[sel].min():.2f} median {np.median(d[sel]):.2f} max {d[sel].max():.2f} us; '
              f'loop cycles median {int(np.median(a[sel, 7]))}')
    for x in sorted(set(xcc.tolist())):
        sel = xcc == x
        print(f'    XCC {x}: {int(sel.sum())} workgroups, stream median {np.median(d[sel]):.2f} max {d[sel].max():.2f} us, clock median {np.median(clk[sel]):.2f} GHz')
    slow = np.argsort(-d)[:12]
    print('  slowest workgroups (stream us, entry us, xcc, cu id, workgroups on that CU): ' +
          '; '.join(f'{d[i]:.1f} {((a[i, 0] - t0) / 100.0):.2f} {int(xcc[i])} {int(cu[i]) & 0xFFF:03x} {int(cnt[inv[i]])}' for i in slow))
dec = a[a[:, 7] == 1]
if len(dec) and (dec[:, 5] > 0).all():
    dd = (dec[:, 6] - dec[:, 5]) / 100.0
    print(f'  decision stage of the deciding workgroups: min {dd.min():.2f} median {np.median(dd):.2f} max {dd.max():.2f} us')
