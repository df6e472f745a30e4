#!/usr/bin/env python3
"""Per basic block of one kernel of a device assembly: MFMA, scratch (spill) traffic, LDS reads, AGPR moves.
usage: isa_blocks.py <asm file> <mangled-name prefix>"""
import re, sys
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
for n, ls in blk:
    c = lambda k: sum(k in x for x in ls)
    if c('v_mfma') or c('scratch_'):
        print(n, len(ls), 'mfma', c('v_mfma'), 'scratch_load', c('scratch_load'), 'scratch_store', c('scratch_store'), 'ds_read', c('ds_read'),
              'acc_read', c('v_accvgpr_read'), 'acc_write', c('v_accvgpr_write'), 'global_load', c('global_load_dwordx4'), 's_nop', c('s_nop'))
