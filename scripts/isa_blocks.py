"""Instruction mix per basic block of one kernel in a -save-temps .s file (measurement helper).
usage: python scripts/isa_blocks.py file.s <kernel-name-substring> [top]"""
import re, sys, collections
path, needle = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 14
lines = open(path).read().split('\n')
start = [i for i, l in enumerate(lines) if needle in l and re.match(r'^_Z\S+:', l)][0]
end = [i for i in range(start, len(lines)) if 's_endpgm' in lines[i]][0]
body = lines[start:end]
def kind(l):
    l = l.strip()
    if not l or l.startswith(';') or l.startswith('.') or l.endswith(':'): return None
    op = l.split()[0]
    if op.startswith('v_'): return 'valu'
    if op.startswith(('s_cbranch', 's_branch')): return 'branch'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith(('s_swappc', 's_setpc')): return 'call'
    if op.startswith('s_nop'): return 'nop'
    if op.startswith('s_'): return 'salu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('scratch', 'buffer')): return 'scratch'
    return 'other'
print(len(body), 'lines', dict(collections.Counter(k for k in map(kind, body) if k)))
blocks = []; cur = ['entry', collections.Counter(), 0]
for i, l in enumerate(body):
    if re.match(r'^\.LBB\d+_\d+:', l):
        blocks.append(cur); cur = [l.split(':')[0], collections.Counter(), i]
    else:
        k = kind(l)
        if k: cur[1][k] += 1
blocks.append(cur)
for name, cc, at in sorted(blocks, key=lambda b: -b[1]['valu'])[:top]:
    print(f"{name:12s} @{at:5d}", dict(cc))
