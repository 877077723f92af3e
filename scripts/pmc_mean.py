import csv,sys
rows={}
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"]
    if sys.argv[2] not in k: continue
    d=rows.setdefault(int(r["Dispatch_Id"]),{"t":float(r["End_Timestamp"])-float(r["Start_Timestamp"])})
    d[r["Counter_Name"]]=d.get(r["Counter_Name"],0.0)+float(r["Counter_Value"])
# the longest launches (dec4a-like, ~220 us)
sel=[d for d in rows.values() if 180e3<d["t"]<260e3]
if not sel: sel=list(rows.values())
names=sorted({k for d in sel for k in d if k!="t"})
print("launches %d, mean %.1f us"%(len(sel),sum(d["t"] for d in sel)/len(sel)/1e3))
for n in names:
    print("  %-44s %.4g"%(n,sum(d.get(n,0) for d in sel)/len(sel)))
