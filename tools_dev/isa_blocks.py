"""dev tool: basic blocks of one kernel in a `hipcc -S` listing: instructions, VALU instructions, memory ops, branches.
   usage: python tools_dev/isa_blocks.py listing.s mangled-name-substring"""
import re, sys
text = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(text) if l.startswith('_Z') and key in l and l.rstrip().endswith(':') or (l.startswith('_Z') and key in l and ': ' in l))
blocks = []; cur = ['entry', 0, 0, []]
for ln in text[start + 1:]:
    s = ln.strip()
    m = re.match(r'^(\.LBB\d+_\d+):', s)
    if m:
        blocks.append(cur); cur = [m.group(1), 0, 0, []]; continue
    if s.startswith('.Lfunc_end'): break
    if not s or s.startswith(';') or s.startswith('.'): continue
    op = s.split()[0]
    cur[1] += 1
    if op.startswith('v_'): cur[2] += 1
    if 'branch' in op: cur[3].append(op[2:] + '->' + s.split()[-1])
    if op.startswith(('global_load', 'ds_', 'global_store', 'scratch', 'buffer_')): cur[3].append(op)
blocks.append(cur)
tot = 0
for b in blocks:
    tot += b[1]; print(b[0], b[1], b[2], ' '.join(b[3]))
print('total', tot)
