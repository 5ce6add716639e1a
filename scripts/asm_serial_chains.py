"""usage: asm_serial_chains.py <asm file> <mangled kernel prefix> [min_run]
Heuristic scan of a kernel's ISA for SERIALISED memory round trips: runs of >= min_run (default 3) consecutive "one or two loads, then s_waitcnt ...cnt(0)"
groups with no other load in flight -- the pattern that cost k_fwd2 ten LDS round trips in its A p_x phase (docs/ROUND_NOTES_r3_r4.md).  Prints the basic blocks with
such runs, their loop depth and the first instructions of the run; use scripts/asm_loops.py <asm> <kernel> <block> to read the block."""
import re, sys
src = open(sys.argv[1]).read().split('\n')
pref = sys.argv[2]
min_run = int(sys.argv[3]) if len(sys.argv) > 3 else 3
start = next(i for i, l in enumerate(src) if l.startswith(pref) and ":" in l)
end = next(i for i in range(start, len(src)) if 's_endpgm' in src[i])
blk = {}; cur = None; order = []; depth = {}
for l in src[start:end]:
    m = re.match(r'\.(LBB\d+_\d+):', l)
    if m:
        cur = m.group(1); blk[cur] = []; order.append(cur)
        d = re.search(r'Depth=(\d+)', l); depth[cur] = int(d.group(1)) if d else 0
    elif cur:
        d = re.search(r'Depth=(\d+)', l)
        if d and not blk[cur]: depth[cur] = max(depth[cur], int(d.group(1)))
        t = l.split(';')[0].strip()
        if t and not t.startswith('.'): blk[cur].append(t)
def is_load(op): return op.startswith(('ds_read', 'ds_bpermute', 'global_load', 'buffer_load', 'scratch_load', 's_load', 'flat_load'))
for b in order:
    ins = blk[b]; runs = []; i = 0; run = 0; first = None; inflight = 0
    for k, t in enumerate(ins):
        op = t.split()[0]
        if is_load(op):
            inflight += 1
        elif op == 's_waitcnt' and re.search(r'(lgkmcnt|vmcnt)\(0\)', t):
            if 1 <= inflight <= 2:
                run += 1
                if first is None: first = k
            else:
                if run >= min_run: runs.append((run, first))
                run = 0; first = None
            inflight = 0
    if run >= min_run: runs.append((run, first))
    for r, f in runs:
        print(f"{b:12s} depth {depth.get(b, 0)}  {r:3d} serialised round trips, from: {' | '.join(ins[max(0, f - 2):f + 1])[:150]}")
