import csv,glob,collections
f=glob.glob("gpurun_out/probe/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "sgm_score" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
acc=collections.defaultdict(list)
for r in rows:
    n=r["Kernel_Name"]; n=n[n.index("sgm_score"):][:58]
    acc[n].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for n,v in acc.items(): print(n, "launches", len(v), "mean us", round(sum(v)/len(v),1), "min", round(min(v),1), "max", round(max(v),1), "total ms", round(sum(v)/1e3,2))
b=[r for r in rows if "band" in r["Kernel_Name"]]
gaps=[(int(b[k+1]["Start_Timestamp"])-int(b[k]["End_Timestamp"]))/1e3 for k in range(len(b)-1)]
gaps=[g for g in gaps if g<100]
print("gap between band launches us: mean", round(sum(gaps)/len(gaps),2))
