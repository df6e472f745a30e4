#!/usr/bin/env python3
"""The shader clock the proposal kernel holds INSIDE its stage loop (MI355X guide, DVFS item 6), from a diagnostic build that
stamps s_memtime / s_memrealtime around the loop of every wave (tools/build_exp.sh clock -DVQ_CLOCK_STAMPS):

    VQHIP_LIB=build/exp/libvqhip_clock.so python tools/inkernel_clock.py [images] [seconds of back-to-back launches]

Runs the headline forward (524 288 bf16 tokens x 16384 x 256, random data) back to back for >= 2 s, then reads the stamps of the
last launch: clock = delta s_memtime / delta s_memrealtime x 100 MHz per wave; prints the median / spread over the waves, the wall
time of the stage loop, and what the kernel's MFMA count makes of that clock (busy fraction of the matrix pipe: 2^28 x 8 passes of
4 cycles... = MFMAs x 16 cycles / (waves x loop cycles))."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vector_quantization_amd import _lib, ops

images = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
N, K, D = images * 256, 16384, 256
g = torch.Generator(device='cuda').manual_seed(3407)
w = torch.randn(K, D, device='cuda', generator=g)
x = torch.randn(N, D, device='cuda', generator=g).bfloat16()
L = _lib.lib()
fn = getattr(ctypes.CDLL(_lib.LIB_PATH), 'vqhip_debug_clock_stamps', None)
if fn is None:
    sys.exit('this library has no stamps: build it with tools/build_exp.sh clock -DVQ_CLOCK_STAMPS and point VQHIP_LIB at it')
cb = ops.prepare_codebook(w, 'L2')
for _ in range(5):
    ops.argmin(x, cb)
torch.cuda.synchronize()
t0, n = time.perf_counter(), 0
L.vqhip_profile_enable(1)
while time.perf_counter() - t0 < seconds:
    ops.argmin(x, cb)
    n += 1
torch.cuda.synchronize()
el = time.perf_counter() - t0
ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
L.vqhip_profile_collect(ctypes.byref(ms), ctypes.byref(cnt))
L.vqhip_profile_enable(0)
slots = 16384
buf = (ctypes.c_ulonglong * (2 * slots))()
fn.restype = ctypes.c_int
rc = fn(buf, slots)
assert rc == 0, rc
a = np.frombuffer(buf, dtype=np.uint64).reshape(slots, 2).astype(np.float64)
a = a[(a[:, 1] > 0) & (a[:, 0] > 0)]
clk = a[:, 0] / a[:, 1] * 100e6 / 1e9
loop_us = a[:, 1] / 100.0
kern_ms = ms.value / max(1, cnt.value)
print(f'{n} back-to-back argmin calls in {el:.2f} s ({N} x {K} x {D}, bf16, random data); proposal kernel {kern_ms:.4f} ms by HIP events '
      f'({2.0 * N * K * D / kern_ms / 1e9:.1f} TFLOP/s = {2.0 * N * K * D / kern_ms / 1e9 / 2500:.3f} of 2.5 PFLOP/s)')
print(f'in-kernel clock over {len(clk)} waves of the last launches: median {np.median(clk):.3f} GHz, 5th-95th percentile '
      f'{np.percentile(clk, 5):.3f}-{np.percentile(clk, 95):.3f} GHz; stage loop of a wave: median {np.median(loop_us):.1f} us '
      f'({np.median(a[:, 0]) / 1e3:.1f} k shader cycles)')
# MFMA pipe: a v_mfma_f32_16x16x32_f16 occupies the pipe for 8 passes x 4 cycles = 16 cycles at one per SIMD... per wave and loop:
mfma_per_wave = (K / 2 / 16) * (D / 32) * 4          # (code rows of the slice / 16) x k-steps x 4 token tiles (two slices at this size)
busy = mfma_per_wave * 16 / np.median(a[:, 0])        # two waves share a SIMD: the pipe's busy fraction is twice a wave's share
print(f'MFMAs per wave and loop {mfma_per_wave:.0f} x 16 pipe cycles = {busy:.3f} of the loop cycles of a wave; two waves per SIMD -> '
      f'matrix pipe {2 * busy:.3f} busy at the median clock')
print(f'at 100 % busy and this clock the chip would deliver {np.median(clk) * 1e9 * 1024 * 2 * 16 * 16 * 32 / 16 / 1e12:.0f} TFLOP/s '
      f'(1024 SIMDs x 2*16*16*32 flop per 16 cycles)')
