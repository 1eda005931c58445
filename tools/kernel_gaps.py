"""durations of the last kernels of a rocprofv3 --kernel-trace run and the idle time of the device in front of each:  python tools/kernel_gaps.py <dir> [count]"""
import csv, sys, glob
f = glob.glob((sys.argv[1] if len(sys.argv) > 1 else "/tmp/prof_s") + "/**/*kernel_trace.csv", recursive=True)[0]
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# take the last 400 kernels: steady-state LM iterations of the last problem
tail = rows[-300:]
prev_end = None
out = []
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("ppsfm::", "").replace("void ", "")[:34]
    out.append((name, (e - s) / 1e3, (s - prev_end) / 1e3 if prev_end else 0.0))
    prev_end = e
for n, d, g in out[-count:]: print("%-36s dur %6.2f us  gap-before %6.2f us" % (n, d, g))
