#!/usr/bin/env python3
"""Kernel table of a rocprofv3 run that wrote the rocpd SQLite format (<dir>/*_results.db): name, launches, total ms, average us.
usage: python tools/rocprof_db_stats.py <dir-or-db> [divide-by-iterations]"""
import glob, os, sqlite3, sys
p = sys.argv[1]
db = p if p.endswith(".db") else sorted(glob.glob(os.path.join(p, "**", "*_results.db"), recursive=True))[-1]
iters = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"{'kernel':100s} {'launches':>8s} {'total ms':>10s} {'avg us':>10s} {'share':>7s}")
for r in rows[:24]:
    print(f"{r[0][:100]:100s} {r[1]:8d} {r[2]:10.3f} {r[3]:10.1f} {100 * r[2] / tot:6.1f}%")
print(f"total {tot:.3f} ms over {sum(r[1] for r in rows)} launches; per iteration ({iters:g}): {tot / iters:.3f} ms")
