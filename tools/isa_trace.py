#!/usr/bin/env python3
"""Compressed instruction-class trace of one kernel in a hipcc -S listing (V valu, r/w LDS read/write, G/S global
load/store, X scratch, |..| s_waitcnt, B barrier, J branch, s other scalar): shows where LDS/global latency is exposed.
usage: isa_trace.py listing.s kernel_name_substring"""
import sys
L = open(sys.argv[1]).read().split('\n')
start = [i for i, l in enumerate(L) if l.startswith('_Z') and sys.argv[2] in l and ':' in l][0]
end = [i for i, l in enumerate(L[start:]) if 's_endpgm' in l][0] + start
body = [l for l in L[start:end] if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
def cls(l):
    op = l.split()[0]
    if op.startswith('v_'): return 'V'
    if op.startswith('ds_read'): return 'r'
    if op.startswith('ds_write'): return 'w'
    if op.startswith('global_load'): return 'G'
    if op.startswith('global_store'): return 'S'
    if op.startswith('scratch'): return 'X'
    if op == 's_waitcnt': return '|' + l.split(None, 1)[1].strip().replace('lgkmcnt', 'l').replace('vmcnt', 'v').replace(' ', '') + '|'
    if op == 's_barrier': return 'B'
    if op.startswith('s_cbranch') or op == 's_branch': return 'J'
    return 's'
out, prev, cnt = [], None, 0
for l in body:
    c = cls(l)
    if c == prev: cnt += 1
    else:
        if prev is not None: out.append(prev + (str(cnt) if cnt > 1 else ''))
        prev, cnt = c, 1
out.append(prev + str(cnt))
print(' '.join(out))
