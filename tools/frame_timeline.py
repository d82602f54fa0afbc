"""Dev tool: the launches of the last frame in a rocprofv3 kernel trace, with start offsets and durations.
python tools/frame_timeline.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "tile_table" in r["Kernel_Name"]]
lo = idx[-1]
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:]:
    n = r["Kernel_Name"].replace("stream::", "").split("(")[0][:60]
    print(f"{(int(r['Start_Timestamp'])-t0)/1e6:8.3f} +{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6:7.3f} ms  {n}")
