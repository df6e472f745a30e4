#!/usr/bin/env python3
"""Per-kernel resource table of a libvqhip device assembly (VQ_KEEP_TEMPS=1 bash vector_quantization_amd/csrc/build.sh ->
build/asm/*gfx950.s): VGPRs, spills, scratch, LDS, instruction and MFMA counts.  usage: isa_stats.py [asm file] [name filter]"""
import glob, re, sys
f = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith('.s') else glob.glob('build/asm/*gfx950.s')[0]
flt = sys.argv[-1] if len(sys.argv) > 1 and not sys.argv[-1].endswith('.s') else ''
s = open(f).read()
meta = {}
for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.wavefront_size', s, re.S):
    b = m.group(2)
    g = lambda k: (re.search(k + r':\s+(\d+)', b) or [None, '?'])[1]
    meta[m.group(1)] = (g(r'\.vgpr_count'), g(r'\.vgpr_spill_count'), g(r'\.private_segment_fixed_size'), g(r'\.group_segment_fixed_size'))
for name in sorted(meta):
    if flt and flt not in name:
        continue
    try:
        i = s.index('\n' + name + ':'); j = s.index('.Lfunc_end', i)
    except ValueError:
        continue
    body = s[i:j].split('\n')
    insts = sum(1 for l in body if re.match(r'\s+[vsdgb]_?\w', l) and not l.strip().startswith('.'))
    mfma = sum('v_mfma' in l for l in body)
    v, sp, sc, lds = meta[name]
    print(f'{name[:100]:102s} vgpr {v:>4s} spill {sp:>3s} scratch {sc:>4s} lds {lds:>6s} insts {insts:6d} mfma {mfma:4d}')
