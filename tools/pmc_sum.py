"""Sum of each rocprofv3 --pmc counter over ALL dispatches of the kernels whose name contains a substring.  python tools/pmc_sum.py <prof dir> <kernel substring>"""
import csv, glob, os, sys, collections, json
d, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float)
n = collections.defaultdict(set)
for f in sorted(glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))):
    for row in csv.DictReader(open(f)):
        if pat in row["Kernel_Name"]:
            acc[row["Counter_Name"]] += float(row["Counter_Value"])
            n[row["Counter_Name"]].add(row["Dispatch_Id"])
print(json.dumps({k: v for k, v in acc.items()}), "dispatches", max(len(v) for v in n.values()) if n else 0)
