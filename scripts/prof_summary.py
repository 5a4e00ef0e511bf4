#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel trace) into a per-kernel table (CSV to stdout)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print("kernel,calls,total_ms,avg_us,min_us,max_us,pct")
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print('"%s",%d,%.3f,%.2f,%.2f,%.2f,%.2f' % (r[0][:110], r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100 * r[2] / tot))
