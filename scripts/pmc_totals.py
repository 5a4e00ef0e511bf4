#!/usr/bin/env python3
"""Sum a rocprofv3 --pmc counter over every dispatch of a run, per kernel and in total.
usage: pmc_totals.py results.db COUNTER [scale]   (FETCH_SIZE: KiB, x2 on gfx950 -> scale 2048; WRITE_SIZE: KiB -> 1024)"""
import sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1])
counter, scale = sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
cols = [d[1] for d in db.execute("pragma table_info(counters_collection)")]
namecol = "kernel_name" if "kernel_name" in cols else "name"
acc, n = defaultdict(float), defaultdict(set)
for kn, cn, v, did in db.execute(f"select {namecol}, counter_name, value, dispatch_id from counters_collection"):
    if cn == counter:
        acc[kn] += float(v) * scale
        n[kn].add(did)
tot = sum(acc.values())
print(f"{counter} total {tot / 1e9:.3f} GB")
for kn, v in sorted(acc.items(), key=lambda kv: -kv[1])[:8]:
    print(f"  {v / 1e9:9.3f} GB  {len(n[kn]):6d} dispatches  {kn[:90]}")
