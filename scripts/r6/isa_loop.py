#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel (llvm-objdump -d of the gfx950 code object): every backward branch = a loop, its body = [target, branch].
usage: isa_loop.py <disassembly> <kernel-name-substring> [min_body]"""
import re, sys, collections
path, name = sys.argv[1], sys.argv[2]
minb = int(sys.argv[3]) if len(sys.argv) > 3 else 200
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^[0-9a-f]+ <', l) and name in l)
end = next((i for i in range(start + 1, len(lines)) if re.match(r'^[0-9a-f]+ <', lines[i])), len(lines))
ins = []
for l in lines[start + 1:end]:
    m = re.match(r'^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):', l)
    if m: ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
addr2idx = {a: i for i, (a, _, _) in enumerate(ins)}
def cls(op, args):
    if op.startswith('v_readlane') or op.startswith('v_readfirstlane'): return 'readlane'
    if op.startswith('v_writelane'): return 'writelane'
    if op.startswith('v_cndmask'): return 'select'
    if op.startswith('v_mov') or op.startswith('v_accvgpr'): return 'mov_dpp' if 'dpp' in op or 'quad_perm' in args or 'row_' in args else 'mov'
    if op.startswith('v_cmp'): return 'cmp'
    if op.startswith('v_'):
        return 'valu_dpp' if ('row_' in args or 'quad_perm' in args) else 'valu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith('global_') or op.startswith('flat_') or op.startswith('buffer_') or op.startswith('scratch_'): return 'vmem'
    if op.startswith('s_waitcnt'): return 'waitcnt'
    if op.startswith('s_nop'): return 'nop'
    if op.startswith('s_load') or op.startswith('s_buffer'): return 'smem'
    if op.startswith('s_'): return 'salu'
    return 'other'
for i, (a, op, args) in enumerate(ins):
    if op.startswith('s_cbranch') or op == 's_branch':
        off = int(args.split()[0])
        if off >= 32768:
            tgt = a + 4 + (off - 65536) * 4
            j = addr2idx.get(tgt)
            if j is None or i - j < minb: continue
            body = ins[j:i + 1]
            c = collections.Counter(cls(o, g) for _, o, g in body)
            tot = len(body); vec = sum(v for k, v in c.items() if k in ('readlane', 'writelane', 'select', 'mov', 'mov_dpp', 'cmp', 'valu', 'valu_dpp'))
            print(f"loop {tgt:x}..{a:x}: {tot} instructions, {vec} vector-ALU;", ' '.join(f"{k} {v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1])))
            ops = collections.Counter(o for _, o, _ in body)
            print('   top ops:', ' '.join(f"{k} {v}" for k, v in ops.most_common(28)))
