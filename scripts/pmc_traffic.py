#!/usr/bin/env python3
"""HBM bytes per GEMM launch from the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of scripts/gpu_pmc.sh
(bench.py --steps 1 --warmup 1 --no-varlen: (2 timed-loop + 2 host-leg) steps x 4 calls x 12 blocks >= 96 full-size launches per
projection, issued after the short query-encode launches, so the LAST 96 dispatches of a kernel are the T = 131 072-token ones; the
residual kernel serves two projections per block, alternating out-proj / fc2).  Corrections as calibrated on
layernorm_kernel on gfx950: FETCH_SIZE (KiB) x 2, WRITE_SIZE (KiB) as reported.
usage: pmc_traffic.py FETCH.db WRITE.db > profiles/rNN_pmc_traffic.json"""
import json
import sqlite3
import sys
from collections import defaultdict

T, D, FFN, LAUNCHES = 131072, 768, 3072, 96


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    cols = [d[1] for d in db.execute("pragma table_info(counters_collection)")]
    namecol = "kernel_name" if "kernel_name" in cols else "name"
    acc = defaultdict(float)
    for kn, cn, v, did in db.execute(f"select {namecol}, counter_name, value, dispatch_id from counters_collection"):
        if cn == counter:
            acc[(kn, did)] += float(v)
    out = defaultdict(list)
    for (kn, did), v in sorted(acc.items(), key=lambda kv: kv[0][1]):
        out[kn].append(v)
    return out


def median(v):
    v = sorted(v)
    return v[len(v) // 2]


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")


import re  # noqa: E402


def pick(table, key, n, phase=None, period=1):
    # kernel names: gemm256d_kernel<operand type, EPI, out type, SWAP, DEEP_A>; key = the EPI code
    names = [k for k in table if re.search(r"gemm256d_kernel<[^,]+, %s," % key, k)]
    names = sorted(names, key=lambda k: -len(table[k]))[:1]      # the operand type / ring orientation the encoder uses
    assert len(names) == 1, (key, list(table)[:20])
    vals = table[names[0]][-n:]
    if phase is not None:
        vals = vals[phase::period]
    return median(vals) * 1024.0


shapes = [  # name, kernel-name key, launches to take, (phase, period), algorithmic bytes
    ("qk_proj  [T,1536]x768 store 16-bit", "0", LAUNCHES, None, T * D * 2 + T * 2 * D * 2 + 2 * D * D * 2),
    ("v_proj   [T,768]x768  V^T 16-bit", "4", LAUNCHES, None, T * D * 2 + T * D * 2 + D * D * 2),
    ("out_proj [T,768]x768  +bias+resid fp32", "2", 2 * LAUNCHES, (0, 2), T * D * 2 + 2 * T * D * 4 + D * D * 2),
    ("fc1      [T,3072]x768 +bias+gelu 16-bit", "1", LAUNCHES, None, T * D * 2 + T * FFN * 2 + D * FFN * 2),
    ("fc2      [T,768]x3072 +bias+resid fp32", "2", 2 * LAUNCHES, (1, 2), T * FFN * 2 + 2 * T * D * 4 + D * FFN * 2),
]
per = {}
for name, key, n, ph, alg in shapes:
    rd = 2.0 * pick(fetch, key, n, *(ph or (None, 1)))
    wr = pick(write, key, n, *(ph or (None, 1)))
    per[name] = {"hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes": rd + wr, "algorithmic_bytes": alg}
avg = sum(v["hbm_bytes"] for v in per.values()) / len(per)
alg = sum(v["algorithmic_bytes"] for v in per.values()) / len(per)
ln = [k for k in fetch if "layernorm_kernel" in k]
cal = {}
if ln:
    cal = {"layernorm_read_bytes": 2.0 * median(fetch[ln[0]][-2 * LAUNCHES:]) * 1024, "layernorm_write_bytes":
           median(write[ln[0]][-2 * LAUNCHES:]) * 1024, "layernorm_algorithmic_read": T * D * 4, "layernorm_algorithmic_write": T * D * 2}
print(json.dumps({"note": __doc__.split("usage:")[0].strip(), "tokens_per_launch": T, "per_launch": per,
                  "gemm_avg_hbm_bytes_per_launch": avg, "gemm_avg_algorithmic_bytes_per_launch": alg,
                  "calibration": cal}, indent=1))
