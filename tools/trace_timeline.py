#!/usr/bin/env python3
"""Print a per-kernel timeline (start offset, duration, gap to previous end on the same queue) from a
rocprofv3 kernel_trace.csv:  python tools/trace_timeline.py trace.csv [first_row] [count] [name_filter]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 60
flt = sys.argv[4] if len(sys.argv) > 4 else ""
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
sel = [r for r in rows if flt in r["Kernel_Name"]]
prev_end = {}
for r in sel[first:first + count]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r.get("Queue_Id", "0")
    gap = s - prev_end.get(q, s)
    prev_end[q] = e
    name = r["Kernel_Name"].split("(")[0].replace("ppsfm::", "")[:28]
    print("%10.1f us  dur %7.1f  gap %7.1f  q%s  grid %s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, q, r.get("Grid_Size", "?"), name))
