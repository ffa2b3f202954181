"""mean per dispatch of every counter in a rocprofv3 counter_collection.csv, per kernel (first 40 chars)"""
import csv, sys, collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"][:44]; acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in acc:
    if len(sys.argv)>2 and sys.argv[2] not in k: continue
    print(k, "dispatches", len(n[k]))
    for c,v in acc[k].items(): print(f"    {c:40s} {v/len(n[k]):16.0f}")
