"""usage: asm_loops.py <asm file> <mangled kernel prefix>  -- per-loop instruction histogram (scratch ops flag spills in hot loops)"""
import re, collections, sys
src = open(sys.argv[1]).read().split('\n')
pref = sys.argv[2]
start = next(i for i, l in enumerate(src) if l.startswith(pref) and ":" in l)
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
    if m: loops[(m.group(1), int(m.group(2)))].append(b)
    m2 = re.search(r'Loop Header: Depth=(\d+)', head)
    if m2: loops[(b[1:], int(m2.group(1)))].append(b)
def hist(bs):
    c = collections.Counter()
    for b in bs:
        for l in blk[b]:
            t = l.strip().split()
            if not t or t[0][0] in ';.': continue
            op = t[0]
            if op.startswith('scratch_'): c['scratch'] += 1
            elif op.startswith('v_fma'): c['fma'] += 1
            elif op.startswith('ds_'): c['ds'] += 1
            elif op == 's_barrier': c['bar'] += 1
            elif op.startswith('global_'): c['glb'] += 1
            elif op.startswith('v_'): c['valu'] += 1
            elif op.startswith('s_'): c['salu'] += 1
    return dict(c)
for (h, d), bs in sorted(loops.items(), key=lambda x: int(x[0][0].split('_')[1])):
    hh = hist(bs)
    if hh.get('scratch') or hh.get('bar') or hh.get('fma', 0) > 8:
        print(h, 'depth', d, 'blocks', len(bs), hh)
print('total', hist(order))
if len(sys.argv) > 3:
    for b in sys.argv[3:]:
        print('\n'.join(blk[b]))
