"""Summarise rocprofv3 --pmc CSVs: mean counter value per dispatch of one kernel.  python tools/pmc_summary.py <prof dir> [kernel substring]"""
import csv, glob, os, sys, collections, json
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "render_tiles_kernel<false, false>"
acc = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))):
    per = collections.defaultdict(float)
    for row in csv.DictReader(open(f)):
        if pat in row["Kernel_Name"]:
            per[(row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
    for (disp, name), v in per.items():
        acc[name].append(v)
out = {k: sum(v) / len(v) for k, v in acc.items()}
print(json.dumps(out, indent=1))
