"""Per-kernel averages of a `rocprofv3 --pmc ... --output-format csv` counter_collection file.
    python tools/pmc_summary.py <counter_collection.csv>            one line per kernel
    python tools/pmc_summary.py --json <counter_collection.csv>     JSON: {kernel: {grid: {counter: mean, "launches": n, "mean_ns": t}}}
Kernels are keyed by their short name AND grid size (k_line_eval at 200k and at 2M observations are different rows)."""
import collections
import csv
import json
import sys

args = [a for a in sys.argv[1:] if a != "--json"]
as_json = "--json" in sys.argv
out = {}
for f in args:
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    for r in rows:
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ppsfm::", "")
        key = (name, int(r["Grid_Size"]))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        try:
            dur[key][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        except (KeyError, ValueError):
            pass
    for (name, grid), cs in sorted(agg.items()):
        n = max(len(v) for v in cs.values())
        d = {c: sum(v) / len(v) for c, v in cs.items()}
        d["launches"] = n
        if dur[(name, grid)]:
            d["mean_ns_under_pmc"] = sum(dur[(name, grid)].values()) / len(dur[(name, grid)])
        out.setdefault(name, {})[str(grid)] = d
        if not as_json:
            print(name, grid, d)
if as_json:
    print(json.dumps(out, indent=1, sort_keys=True))
