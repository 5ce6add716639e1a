import csv, re, sys
rows=sorted(csv.DictReader(open(sys.argv[1])), key=lambda r:int(r["Start_Timestamp"]))
def short(nm):
    m=re.search(r"\bk_\w+",nm)
    if m: return m.group(0)
    m=re.search(r"(\w+)<",nm) or re.search(r"(\w+)\(",nm)
    return (m.group(1) if m else nm)[:40]
names=[short(r["Kernel_Name"]) for r in rows]
starts=[i for i,n in enumerate(names) if n.startswith("k_fwd2")]
# pick a fwd2 followed by a backward within the next 40 kernels
for a in starts[::-1]:
    seg=names[a:a+40]
    if any(s.startswith("k_backward") for s in seg):
        break
b=a+40
prev=None
t0=int(rows[a-6]["Start_Timestamp"])
for k in range(max(a-6,0),min(b,len(rows))):
    r=rows[k]; s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print(f"{(s-t0)/1e3:9.1f} {names[k]:42s} gap {((s-prev)/1e3 if prev else 0):8.1f} us  dur {(e-s)/1e3:8.1f} us")
    prev=e
