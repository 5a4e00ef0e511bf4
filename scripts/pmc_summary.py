#!/usr/bin/env python3
"""Per-kernel averages of PMC counters from rocprofv3 rocpd databases (one db per --pmc pass)."""
import sqlite3
import sys
from collections import defaultdict

agg = defaultdict(lambda: defaultdict(list))
for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    if "counters_collection" in tabs:
        cols = [d[1] for d in db.execute("pragma table_info(counters_collection)")]
        namecol = "kernel_name" if "kernel_name" in cols else "name"
        q = f"select {namecol}, counter_name, value, dispatch_id from counters_collection"
        per_dispatch = defaultdict(float)
        for kn, cn, v, did in db.execute(q):
            per_dispatch[(kn, cn, did)] += float(v)          # sum over XCDs / instances
        for (kn, cn, did), v in per_dispatch.items():
            agg[kn][cn].append(v)
    else:
        print("# no counters_collection view in", path, "tables:", [t for t in tabs if "pmc" in t or "counter" in t])
print("kernel,counter,avg_per_dispatch,median_per_dispatch,max_per_dispatch,dispatches")
for kn in sorted(agg, key=lambda k: -sum(sum(a) for a in agg[k].values())):
    for cn, vals in sorted(agg[kn].items()):
        vs = sorted(vals)
        print('"%s",%s,%.1f,%.1f,%.1f,%d' % (kn[:100], cn, sum(vs) / len(vs), vs[len(vs) // 2], vs[-1], len(vs)))
