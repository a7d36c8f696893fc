"""Print the tail of a rocprofv3 kernel trace as a timeline: start, duration, gap to the previous
kernel's end (all us), grid size, kernel name.   usage: timeline.py <kernel_trace.csv> [n_last] [anchor]"""
import csv, sys
path = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 40
anchor = sys.argv[3] if len(sys.argv) > 3 else None
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
if anchor:                      # window = two consecutive launches of the anchor kernel
    idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
    mid = len(idx) // 2             # two steps from the middle of the run (steady state)
    rows = rows[idx[mid]:idx[mid + 2]] if len(idx) >= 4 else rows[-n_last:]
else:
    rows = rows[-n_last:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
print("# start_us  dur_us  gap_us   grid  kernel")
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
    print("%9.1f %7.1f %7.1f %8d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, grid, r["Kernel_Name"][:70]))
    prev_end = max(e, prev_end or e)
