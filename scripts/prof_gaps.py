#!/usr/bin/env python3
"""From a rocprofv3 rocpd database (kernel trace): the busy and idle time of the GPU timeline over the LAST `n` kernels
(the steady state of a repeated call) -- sum of kernel durations, span, and the gaps between consecutive kernels."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
rows = list(db.execute("select name, start, end from kernels order by start"))[-n:]
busy = sum(e - s for _, s, e in rows)
span = rows[-1][2] - rows[0][1]
gaps = [rows[i + 1][1] - rows[i][2] for i in range(len(rows) - 1)]
pos = sorted(g for g in gaps if g > 0)
print(f"kernels {len(rows)}: busy {busy / 1e6:.3f} ms, span {span / 1e6:.3f} ms, idle {100 * (span - busy) / span:.1f} %; "
      f"gap median {pos[len(pos) // 2] / 1e3:.2f} us, p90 {pos[int(len(pos) * 0.9)] / 1e3:.2f} us, max {pos[-1] / 1e3:.1f} us")
