#!/bin/bash
# usage: asm_hotloop.sh <mangled-prefix>   -- dumps the kernel's asm and reports scratch ops inside the Depth=2 hot loop
cd /root/repo/cvxpylayers_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -I../../include -S --cuda-device-only -o /tmp/asm/ce3.s cone_engine.hip 2>/dev/null
awk "/^$1/,/s_endpgm/" /tmp/asm/ce3.s > /tmp/asm/rt.s
python3 - <<'PY'
import re
lines = open('/tmp/asm/rt.s').read().split('\n')
his = [i for i,l in enumerate(lines) if 'This Loop Header: Depth=2' in l]
for hi in his:
    lab=None
    for k in range(hi, hi-6, -1):
        m = re.match(r'\.(LBB\d+_\d+):', lines[k])
        if m: lab = m.group(1); break
    idx=[i for i,l in enumerate(lines) if f'Header={lab[1:]}' in l]
    if not idx: continue
    last=max(idx)
    sc=[(i-hi) for i in range(hi,last) if 'scratch_' in lines[i]]
    bars=[(i-hi) for i in range(hi,last) if 's_barrier' in lines[i]]
    print(lab, "lines", hi, last, "scratch at", sc[:40], "barriers at", bars[:12])
print("total lines", len(lines), "total scratch", sum('scratch_' in l for l in lines))
PY
