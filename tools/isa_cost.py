#!/usr/bin/env python3
"""Weighted VALU cost per kernel of a hipcc -S listing: half-rate 64-bit/multiply forms count 2, other VALU 1
(measured with tools/microbench/valu_cost.hip on gfx950)."""
import sys, collections, re
HALF = ("v_lshl_add_u64", "v_mad_u64_u32", "v_mad_i64_i32", "v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32")
def cost(op):
    if not op.startswith("v_"): return 0
    if op in HALF or (op.startswith("v_cmp") and "64" in op): return 2
    return 1
L = open(sys.argv[1]).read().split('\n')
idx = {l.split(':')[0]: i for i, l in enumerate(L) if l.startswith('_Z') and ':' in l}
names = sorted(idx.items(), key=lambda kv: kv[1])
base = None
for (n, a), (n2, b) in zip(names, names[1:] + [('end', len(L))]):
    ins = [l.split()[0] for l in L[a:b] if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    c = sum(cost(i) for i in ins); nv = sum(1 for i in ins if i.startswith('v_')); nop = sum(1 for i in ins if i == 's_nop')
    print(f"{n[:44]:46s} valu {nv:6d}  weighted {c:6d}  nop {nop:5d}")
