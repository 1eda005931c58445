"""Timeline of steady-state LM iterations from a rocprofv3 kernel trace of `bench.py --no-ransac --no-cpu-baseline`:
python tools/lm_timeline.py <kernel_trace.csv> [iteration index]  — the column steps of a Cholesky are folded into one line."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_potrf64" in r["Kernel_Name"]]
it = int(sys.argv[2]) if len(sys.argv) > 2 else 8
a, b = idx[it], idx[it + 1]
t0 = int(rows[a]["Start_Timestamp"])
prev = None
gaps = 0.0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("ppsfm::", "")[:34]
    if "k_column_step" in name and prev is not None and s - prev < 500:
        prev = e
        continue
    gap = (s - prev) / 1e3 if prev else 0
    gaps += max(gap, 0)
    print("%9.1f dur %7.1f gap %6.1f %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, name))
    prev = e
end = int(rows[b]["Start_Timestamp"])
print("--- iteration span %.1f us, idle gaps %.1f us (+%.1f before the next factorisation)" % ((end - t0) / 1e3, gaps, (end - prev) / 1e3))
