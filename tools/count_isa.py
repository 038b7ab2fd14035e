#!/usr/bin/env python3
"""instruction mix per kernel of a hipcc -S listing"""
import sys, collections
L = open(sys.argv[1]).read().split('\n')
idx = {l.split(':')[0]: i for i, l in enumerate(L) if l.startswith('_ZN') and ':' in l}
names = sorted(idx.items(), key=lambda kv: kv[1])
for (n, a), (n2, b) in zip(names, names[1:] + [('end', len(L))]):
    ins = [l.split()[0] for l in L[a:b] if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    c = collections.Counter(ins)
    print(f"{n[20:52]:34s}{len(ins):7d}  mad64 {c['v_mad_u64_u32']:5d}  mul32 {c['v_mul_lo_u32']+c['v_mul_hi_u32']:5d}  nop {c['s_nop']:5d}  cnd {c['v_cndmask_b32_e32']+c['v_cndmask_b32_e64']:6d}  cmp64 {sum(v for k,v in c.items() if k.startswith('v_cmp') and 'u64' in k):5d}  add64 {c['v_lshl_add_u64']:5d}")
