import csv, sys, collections, glob
rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if "sweep" not in k: continue
    key = (k[:90], r.get("Grid_Size"), r.get("VGPR_Count"), r.get("LDS_Block_Size"))
    agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, cs in sorted(agg.items()):
    print(key)
    for c, v in sorted(cs.items()):
        print(f"    {c:32s} {sum(v)/len(v):16.0f}  (n={len(v)})")
