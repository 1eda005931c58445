import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_column_step" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
T = int(sys.argv[2]) if len(sys.argv) > 2 else 46
last = rows[-T:]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in last]
gap = [(int(last[i + 1]["Start_Timestamp"]) - int(last[i]["End_Timestamp"])) / 1e3 for i in range(len(last) - 1)]
print("k_column_step durations (us), k = 0..:", " ".join("%.1f" % v for v in d))
print("sum %.1f us, mean %.2f; gaps mean %.2f us" % (sum(d), sum(d) / len(d), sum(gap) / len(gap)))
