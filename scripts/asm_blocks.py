"""usage: asm_blocks.py <asm> <kernel prefix>: per-block stats of the loop with most FMAs (the iteration loop)"""
import re, collections, sys
src = open(sys.argv[1]).read().split('\n')
start = next(i for i, l in enumerate(src) if l.startswith(sys.argv[2]) and ':' in l)
end = next(i for i in range(start, len(src)) if 's_endpgm' in src[i])
lines = src[start:end]
blk = {}; cur = None; order = []
for l in lines:
    m = re.match(r'\.(LBB\d+_\d+):', l)
    if m: cur = m.group(1); blk[cur] = []; order.append(cur)
    if cur: blk[cur].append(l)
loops = collections.defaultdict(list)
for b in order:
    head = ' '.join(blk[b][:4])
    m = re.search(r'Header=(BB\d+_\d+) Depth=(\d+)', head)
    if m: loops[m.group(1)].append(b)
    if 'Loop Header' in head: loops[b[1:]].append(b)
best = max(loops, key=lambda h: sum(sum('v_fma' in l for l in blk[b]) for b in loops[h]))
print('iteration loop', best)
for b in loops[best]:
    c = collections.Counter()
    for l in blk[b]:
        t = l.strip().split()
        if not t or t[0][0] in ';.': continue
        op = t[0]
        if op.startswith('scratch_load'): c['sld'] += 1
        elif op.startswith('scratch_store'): c['sst'] += 1
        elif op.startswith('v_fma'): c['fma'] += 1
        elif op.startswith('ds_'): c['ds'] += 1
        elif op == 's_barrier': c['bar'] += 1
        elif op.startswith('v_'): c['valu'] += 1
        elif op.startswith('s_'): c['salu'] += 1
    if sum(c.values()) > 25 or c.get('sld') or c.get('sst'): print(b, dict(c))
