#!/usr/bin/env python3
"""Per-kernel totals of a rocprofv3 results.db (rocpd sqlite): tools/prof_db.py <dir-or-db> [top]"""
import glob, os, sqlite3, sys
src = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dbs = [src] if src.endswith(".db") else glob.glob(os.path.join(src, "**", "*.db"), recursive=True)
for f in dbs:
    db = sqlite3.connect(f)
    t = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [x for x in t if "kernel_dispatch" in x][0]
    ks = [x for x in t if "kernel_symbol" in x][0]
    rows = list(db.execute(f"select s.kernel_name, count(*), sum(d.end-d.start)/1e3, avg(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id group by 1 order by 3 desc"))
    tot = sum(r[2] for r in rows)
    print(f"{f}: {len(rows)} kernels, total {tot:.1f} us")
    for r in rows[:top]:
        print(f"{r[0][:100]:100s} n={r[1]:5d} tot={r[2]:10.1f}us avg={r[3]:9.1f}us")
