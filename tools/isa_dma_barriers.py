#!/usr/bin/env python3
"""LDS-DMA hygiene of a libvqhip device assembly (VQ_KEEP_TEMPS=1 bash vector_quantization_amd/csrc/build.sh -> build/asm/*gfx950.s).
A kernel that fills LDS with global_load_lds / buffer_load ... lds and hands the tile to other waves must have drained vmcnt
before the s_barrier: __syncthreads() does not promise that (a release fence owes nothing to outstanding loads; hipcc emits the
s_waitcnt vmcnt(0) only where something else needs it).  Per kernel with LDS-DMA: every s_barrier, whether the straight-line code
in front of it (back to the previous barrier, label or branch) holds an s_waitcnt vmcnt(0), and whether an LDS-DMA was issued in
that stretch.  A barrier marked `?` starts a basic block: what reaches it has to be read in the listing.
usage: isa_dma_barriers.py [asm file] [name filter]"""
import glob, re, sys
f = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith('.s') else glob.glob('build/asm/*gfx950.s')[0]
flt = sys.argv[-1] if len(sys.argv) > 1 and not sys.argv[-1].endswith('.s') else ''
s = open(f).read()
for m in re.finditer(r'\n(_Z\w+|\w+_kernel\w*):[^\n]*\n(.*?)\.Lfunc_end', s, re.S):
    name, body = m.group(1), m.group(2)
    if flt and flt not in name:
        continue
    if 'global_load_lds' not in body and not re.search(r'buffer_load\w+ .* lds', body):
        continue
    lines = [l.strip() for l in body.split('\n')]
    nbar = ok = 0
    notes = []
    for i, l in enumerate(lines):
        if not l.startswith('s_barrier'):
            continue
        nbar += 1
        drained = dma = False
        how = '?'
        for j in range(i - 1, -1, -1):
            t = lines[j]
            if re.match(r's_waitcnt .*vmcnt\(0\)', t):
                drained = True; how = 'drained'; break
            if 'global_load_lds' in t or re.search(r'buffer_load\w+ .* lds', t):
                dma = True
            if t.startswith('s_barrier') or t.startswith('s_cbranch') or t.startswith('s_branch') or re.match(r'\.LBB\d+_\d+:', t):
                how = 'block start' if re.match(r'\.LBB', t) else 'after ' + t.split()[0]
                break
        if drained:
            ok += 1
        else:
            notes.append(f'line {i}: no vmcnt(0) back to {how}' + (' — LDS-DMA issued in between!' if dma else ''))
    print(f'{name[:90]:92s} barriers {nbar:3d}  with vmcnt(0) in front {ok:3d}')
    for n in notes:
        print('      ' + n)
