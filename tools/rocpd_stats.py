"""Per-kernel totals from a rocprofv3 rocpd database (when --output-format csv was not asked for): python tools/rocpd_stats.py <results.db> [top]"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
name = 'display_name' if 'display_name' in cols else 'kernel_name'
rows = cur.execute(f"select s.{name}, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.{name} order by 3 desc").fetchall()
total = sum(r[2] for r in rows)
print("%-70s %7s %12s %10s %8s %8s %6s" % ("kernel", "calls", "total us", "avg us", "min", "max", "%"))
for n, c, t, mn, mx in rows[:top]:
    n = re.sub(r"\(.*", "", n)[:70]
    print("%-70s %7d %12.1f %10.2f %8.2f %8.2f %6.2f" % (n, c, t / 1e3, t / c / 1e3, mn / 1e3, mx / 1e3, 100 * t / total))
print("total %.1f us" % (total / 1e3))
