import csv, sys, glob, collections
for f in sys.argv[1:]:
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        name = r["Kernel_Name"].split("(")[0][-40:]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name, cs in agg.items():
        if "schur" in name or "k_reduce" in name or "line_eval" in name or "obs_prepare" in name:
            print(name, {c: (sum(v) / len(v), len(v)) for c, v in cs.items()})
