#!/usr/bin/env python3
"""Instruction histogram of the basic block of a kernel that holds the most MFMAs (the stage body of a proposal kernel).
usage: isa_body.py <asm file> <mangled-name prefix>"""
import collections, re, sys
s = open(sys.argv[1]).read()
nm = sys.argv[2]
i = s.index('\n' + nm); i = s.index(':', i); j = s.index('.Lfunc_end', i)
blk = []; cur = ['entry', []]
for l in s[i:j].split('\n'):
    if re.match(r'^\.LBB\d+_\d+:', l):
        blk.append(cur); cur = [l.strip(), []]
    else:
        cur[1].append(l)
blk.append(cur)
m = re.search(r'\.vgpr_count:\s*(\d+)', s[j:j + 20000])
best = max(blk, key=lambda b: sum('v_mfma' in x for x in b[1]))
ops = collections.Counter()
for l in best[1]:
    l = l.strip()
    if not l or l.startswith(';') or l.startswith('.'):
        continue
    ops[l.split()[0]] += 1
n_mfma = sum(v for k, v in ops.items() if k.startswith('v_mfma'))
valu = sum(v for k, v in ops.items() if k.startswith('v_') and not k.startswith('v_mfma'))
print(best[0].split(';')[0], 'instructions', sum(ops.values()), 'mfma', n_mfma, 'other VALU', valu, f'= {valu / max(1, n_mfma):.2f} per MFMA')
print('  ' + ', '.join(f'{k} {v}' for k, v in sorted(ops.items(), key=lambda kv: -kv[1])[:24]))
