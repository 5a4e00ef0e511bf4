#!/usr/bin/env python3
"""Per-kernel AND per-launch-shape table from a rocprofv3 rocpd database: `prof_by_grid.py trace.db <name substring>` groups the
launches of the matching kernels by grid size (e.g. the attention launch of each call of a variable-length step)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
gcols = [c for c in cols if "grid" in c.lower() or "workgroup" in c.lower()]
q = "select name, %s, count(*), avg(end-start), min(end-start), max(end-start) from kernels where name like ? group by name, %s order by 1, 2" % (
    ", ".join(gcols), ", ".join(gcols))
print("kernel," + ",".join(gcols) + ",calls,avg_us,min_us,max_us")
for r in db.execute(q, ("%" + sys.argv[2] + "%",)):
    n = len(gcols)
    print('"%s",%s,%d,%.2f,%.2f,%.2f' % (r[0][:90], ",".join(str(x) for x in r[1:1 + n]), r[1 + n], r[2 + n] / 1e3, r[3 + n] / 1e3, r[4 + n] / 1e3))
