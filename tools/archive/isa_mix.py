#!/usr/bin/env python3
"""Instruction mix of the kernels in a gfx950 assembly file (hipcc -S --cuda-device-only): totals per kernel and for the
largest loop bodies (a loop = a label that a later branch jumps back to).  usage: tools/archive/isa_mix.py file.s [name-regex]"""
import collections, re, sys

def group(op):
    if op.startswith('v_pk'): return 'v_pk'
    if 'dpp' in op: return 'dpp'
    if op.startswith('v_mov_b32') or op.startswith('v_accvgpr'): return 'v_mov'
    if op.startswith('v_cndmask'): return 'cndmask'
    if op.startswith('v_cmp'): return 'v_cmp'
    if op.startswith(('v_rsq', 'v_sqrt', 'v_rcp')): return 'trans'
    if op.startswith('v_') and op.endswith('f64'): return 'v_f64'
    if op.startswith('v_'): return 'valu'
    if op.startswith('ds_bpermute'): return 'ds_bpermute'
    if op.startswith('ds_'): return 'ds'
    if op.startswith(('global_load', 'buffer_load')): return 'vm_load'
    if op.startswith(('global_store', 'buffer_store')): return 'vm_store'
    if op.startswith('scratch_'): return 'scratch'
    if op.startswith('s_waitcnt'): return 's_waitcnt'
    if op.startswith('s_barrier'): return 's_barrier'
    if op.startswith(('s_cbranch', 's_branch')): return 'branch'
    if op.startswith('s_'): return 'salu'
    return op

def mix(lines):
    c = collections.Counter(group(l.split()[0]) for l in lines)
    return dict(sorted(c.items(), key=lambda kv: -kv[1]))

txt = open(sys.argv[1]).read()
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
parts = re.split(r'\n(_Z[^\s:]*):[^\n]*\n', txt)
for name, body in zip(parts[1::2], parts[2::2]):
    if pat and not pat.search(name):
        continue
    body = body.split('s_endpgm')[0]
    rows = body.split('\n')
    ins, labels = [], {}
    for l in rows:
        m = re.match(r'^(\.LBB[0-9_]+):', l)
        if m:
            labels[m.group(1)] = len(ins)
        elif l.startswith('\t') and l.strip() and not l.strip().startswith(('.', ';')):
            ins.append(l.strip())
    print(name[:100])
    print('  total %d:' % len(ins), mix(ins))
    loops = []
    for i, l in enumerate(ins):
        m = re.match(r'^s_c?branch\S*\s+(\.LBB[0-9_]+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] <= i:
            loops.append((i - labels[m.group(1)], labels[m.group(1)], i))
    for n, a, b in sorted(loops, reverse=True)[:4]:
        print('  loop of %d instructions:' % n, mix(ins[a:b + 1]))
