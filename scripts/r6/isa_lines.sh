#!/bin/bash
# scripts/r6/isa_lines.sh "<extra flags>" <kernel-substring> <loop-start-hex> <loop-end-hex>: vector-ALU instructions of one loop per SOURCE LINE (line tables, llvm-objdump -l)
set -e
cd "$(dirname "$0")/../.."
EXTRA=$1; K=$2
T=$(mktemp -d /tmp/isal.XXXX)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -gline-tables-only -Wno-unused-result -Wno-unused-value -Ildso_amd/csrc -Iinclude $EXTRA -c ldso_amd/csrc/ba_linearize.hip -o $T/k.o
B=/opt/rocm/lib/llvm/bin
$B/llvm-objcopy --dump-section .hip_fatbin=$T/k.fat $T/k.o
$B/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/k.fat --output=$T/k.co --unbundle
$B/llvm-objdump -d -l $T/k.co > $T/k.s
python scripts/r6/isa_loop.py $T/k.s "$K" ${MINB:-1000} | grep -v "top ops"
python - "$T/k.s" "$K" <<'PY'
import re, sys, collections
path, name = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^[0-9a-f]+ <', l) and name in l)
end = next((i for i in range(start + 1, len(lines)) if re.match(r'^[0-9a-f]+ <', lines[i])), len(lines))
# find the biggest inner loop: last backward branch with body >= 1000 whose body is smallest
ins = []; cur = None
for l in lines[start + 1:end]:
    m = re.match(r'^; .*?([^/]+):(\d+)$', l)
    if m: cur = int(m.group(2)) if m.group(1).startswith('ba_linearize') else -1; continue
    m = re.match(r'^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):', l)
    if m: ins.append((int(m.group(3), 16), m.group(1), m.group(2), cur))
idx = {a: i for i, (a, _, _, _) in enumerate(ins)}
best = None
for i, (a, op, args, _) in enumerate(ins):
    if op.startswith('s_cbranch') or op == 's_branch':
        off = int(args.split()[0])
        if off >= 32768:
            j = idx.get(a + 4 + (off - 65536) * 4)
            if j is not None and i - j >= 1000 and (best is None or i - j < best[1] - best[0]): best = (j, i)
j, i = best
c = collections.Counter(); kinds = collections.defaultdict(collections.Counter)
for a, op, args, ln in ins[j:i + 1]:
    if op.startswith('v_'):
        c[ln] += 1
        kinds[ln]['mov' if op.startswith('v_mov_b32_e32') or op.startswith('v_mov_b64') else 'sel' if op.startswith('v_cndmask') else 'cmp' if op.startswith('v_cmp') else 'rdl' if 'lane' in op else 'dpp' if ('row_' in args or 'quad_perm' in args) else 'alu'] += 1
print("vector instructions per source line (loop %x..%x, %d total):" % (ins[j][0], ins[i][0], sum(c.values())))
for ln, n in sorted(c.items()):
    print(f"  line {ln:5d}: {n:4d}  " + ' '.join(f"{k} {v}" for k, v in kinds[ln].most_common()))
PY
echo "disassembly: $T/k.s"
