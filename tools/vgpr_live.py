#!/usr/bin/env python3
"""Approximate VGPR liveness along a kernel's disassembly (linear order, loop handled by two
backward passes): tools/kinfo.sh obj filter out.s; tools/vgpr_live.py out.s <kernel substring>"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
start = [i for i, l in enumerate(lines) if sys.argv[2] in l and l.endswith('>:')][0]
end = start + 1
while end < len(lines) and not lines[end].endswith('>:'):
    end += 1
body = [l.split('//')[0].strip() for l in lines[start + 1:end] if l.strip()]
def regs(tok):
    out = []
    for a, b in re.findall(r'\bv\[(\d+):(\d+)\]', tok):
        out += list(range(int(a), int(b) + 1))
    for a in re.findall(r'\bv(\d+)\b', tok):
        out.append(int(a))
    return out
NODEF = ('global_store', 'ds_write', 'ds_add', 'buffer_store', 'scratch_store', 'v_cmp', 's_', 'v_writelane', 'v_readlane', 'v_readfirstlane', 'ds_max', 'ds_min')
ins = []
for l in body:
    parts = l.split(None, 1)
    op = parts[0]
    ops = parts[1].split(',') if len(parts) > 1 else []
    if op.startswith(NODEF) and not op.startswith('v_writelane'):
        d, u = [], sum((regs(o) for o in ops), [])
        if op.startswith('v_readlane') or op.startswith('v_readfirstlane'):
            u = sum((regs(o) for o in ops[1:]), [])
    elif op.startswith('v_writelane'):
        d, u = [], regs(ops[0])  # partial def: keeps the register live
    else:
        d = regs(ops[0]) if ops else []
        u = sum((regs(o) for o in ops[1:]), [])
        if 'fmac' in op or 'mac_' in op: u += d
    ins.append((op, set(d), set(u)))
live = set()
counts = [0] * len(ins)
for _ in range(2):
    for i in range(len(ins) - 1, -1, -1):
        op, d, u = ins[i]
        live -= d
        live |= u
        counts[i] = len(live)
W = int(sys.argv[3]) if len(sys.argv) > 3 else 100
for i in range(0, len(ins), W):
    c = counts[i:i + W]
    ops = [x[0] for x in ins[i:i + W]]
    def n(p): return sum(1 for o in ops if o.startswith(p))
    print(f"{i:5d} live max {max(c):3d} min {min(c):3d} | f64 {n('v_fma_f64')+n('v_mul_f64')+n('v_add_f64'):3d} dsr {n('ds_read'):2d} dsw {n('ds_write'):2d} dsadd {n('ds_add'):2d} gld {n('global_load'):2d} gst {n('global_store'):2d} bar {n('s_barrier')} salu {n('s_'):3d} rdlane {n('v_readlane'):2d}")
print("peak", max(counts), "at", counts.index(max(counts)))
