"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel name (mean per dispatch)."""
import csv, glob, sys, collections
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if pat and pat not in k:
            continue
        acc[k[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(k)
        for c, v in sorted(cs.items()):
            print(f"   {c:32s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
