"""Dev tool: per-dispatch timeline of the last frame in a rocprofv3 --kernel-trace CSV.  python tools/timeline.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "tile_table" in r["Kernel_Name"]]
start = idx[-1]
t0 = int(rows[start]["Start_Timestamp"])
prev_end = None
for r in rows[start:]:
    n = r["Kernel_Name"].split("(")[0].replace("void stream::", "").replace("stream::", "")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.
    print(f"{(s - t0) / 1e6:8.3f} ms  dur {(e - s) / 1e3:9.1f} us  gap {gap:6.1f}  grid {r['Grid_Size_X']:>8s}  {n[:48]}")
    prev_end = e
