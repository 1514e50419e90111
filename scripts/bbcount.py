#!/usr/bin/env python3
"""Per-basic-block instruction counts of a gfx950 .s file (CPU-only analysis aid).
usage: bbcount.py file.s [first_line last_line]
Prints, for every basic block in the range: first line, label, VALU / SALU / LDS / VMEM counts, the marks
it contains and how it ends -- enough to add up the hot path of the worker loop by hand."""
import re, sys
path = sys.argv[1]
lines = open(path).read().split('\n')
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 1
hi = int(sys.argv[3]) if len(sys.argv) > 3 else len(lines)
blocks = []
cur = None
def new(label, i):
    global cur
    cur = dict(label=label, line=i, valu=0, salu=0, lds=0, vmem=0, marks=[], end='', rept=0)
    blocks.append(cur)
new('(start)', lo)
for i in range(lo - 1, hi):
    l = lines[i]
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        new(m.group(1), i + 1); continue
    m = re.match(r'^; %bb\.(\d+):', l)
    if m:
        new('bb.' + m.group(1), i + 1); continue
    m = re.search(r'; MSJ_MARK (\w+)', l)
    if m: cur['marks'].append(m.group(1)); continue
    if re.match(r'\s*\.rept', l): cur['rept'] += 1
    m = re.match(r'\s+([a-z_0-9]+)', l)
    if not m: continue
    op = m.group(1)
    if op.startswith('v_'): cur['valu'] += 1
    elif op.startswith('s_cbranch') or op.startswith('s_branch') or op.startswith('s_setpc') or op.startswith('s_swappc'):
        cur['salu'] += 1; cur['end'] += ' ' + l.strip().replace('\t', ' ')
    elif op.startswith('s_'): cur['salu'] += 1
    elif op.startswith('ds_'): cur['lds'] += 1
    elif op.startswith(('global_', 'flat_', 'buffer_', 'scratch_')): cur['vmem'] += 1
for b in blocks:
    print(f"{b['line']:6d} {b['label']:12s} valu {b['valu']:4d} salu {b['salu']:4d} lds {b['lds']:3d} vmem {b['vmem']:3d}"
          f"{' rept x' + str(b['rept']) if b['rept'] else ''} {'marks ' + ','.join(b['marks']) if b['marks'] else ''} ->{b['end']}")
