"""Summarise a rocprofv3 run (rocpd sqlite .db, ROCm 7.2 default output) as a per-kernel table:
calls, total ms, average / min / max us, share -- the `--stats` view -- plus PMC counter sums when present.
Usage: python scripts/prof_summary.py <dir-with-db> [min_us]"""
import glob
import sqlite3
import sys

d = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
dbs = glob.glob(d + "/**/*.db", recursive=True)
if not dbs:
    sys.exit("no .db under " + d)
cur = sqlite3.connect(dbs[0]).cursor()
rows = cur.execute(
    "select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
    "from kernels where (end-start) >= ? group by name order by 3 desc", (min_us * 1e3,)).fetchall()
tot = sum(r[2] for r in rows) or 1.0
print(f"# {dbs[0]}  (kernels with duration >= {min_us} us)")
print(f"{'kernel':100s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}")
for r in rows[:40]:
    print(f"{r[0][:100]:100s} {r[1]:6d} {r[2]:10.2f} {r[3]:9.1f} {r[4]:9.1f} {r[5]:9.1f} {100 * r[2] / tot:6.1f}")
try:
    cols = [c[1] for c in cur.execute("pragma table_info(counters_collection)")]
    if cols:
        q = cur.execute("select * from counters_collection limit 1").fetchall()
        if q:
            name_col = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else None)
            cn = "counter_name" if "counter_name" in cols else None
            val = "value" if "value" in cols else ("counter_value" if "counter_value" in cols else None)
            if name_col and cn and val:
                print("\n# PMC counters (sum over dispatches / dispatches)")
                for r in cur.execute(f"select {name_col}, {cn}, count(*), sum({val}), avg({val}) from counters_collection "
                                     f"group by {name_col}, {cn} order by 4 desc limit 60"):
                    print(f"{str(r[0])[:90]:90s} {r[1]:14s} n={r[2]:6d} sum={r[3]:.6g} avg={r[4]:.6g}")
            else:
                print("\n# counters_collection columns:", cols)
except Exception as e:  # pragma: no cover
    print("# (no counter table:", e, ")")
