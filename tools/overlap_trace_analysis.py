import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
ev=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]) for r in rows]
bl=[e for e in ev if "blend_walk_kernel" in e[2]]
pr=[e for e in ev if "preprocess_kernel" in e[2]]
import statistics as st
def dur(x): return [(e[1]-e[0])/1e3 for e in x]
print("blend launches",len(bl),"median dur",st.median(dur(bl)),"min",min(dur(bl)),"max",max(dur(bl)))
print("preprocess launches",len(pr),"median dur",st.median(dur(pr)),"min",min(dur(pr)),"max",max(dur(pr)))
# for each preprocess launch: how much of it overlaps any blend launch, and durations in that case
ov=[]
for p in pr:
    o=0
    for b in bl:
        lo=max(p[0],b[0]); hi=min(p[1],b[1])
        if hi>lo: o+=hi-lo
    ov.append((o/1e3,(p[1]-p[0])/1e3))
both=[x for x in ov if x[0]>0.5*x[1]]
print("preprocess launches mostly under a blend:",len(both),"their median duration",st.median([x[1] for x in both]) if both else None)
alone=[x for x in ov if x[0]==0]
print("preprocess launches with no blend running:",len(alone),"median duration",st.median([x[1] for x in alone]) if alone else None)
# blend durations when a preprocess overlaps
bo=[]
for b in bl:
    o=sum(max(0,min(p[1],b[1])-max(p[0],b[0])) for p in pr)
    bo.append((o/1e3,(b[1]-b[0])/1e3))
w=[x[1] for x in bo if x[0]>100]; wo=[x[1] for x in bo if x[0]==0]
print("blend durations with >100us of preprocess beside:",len(w),st.median(w) if w else None,"| with none:",len(wo),st.median(wo) if wo else None)
