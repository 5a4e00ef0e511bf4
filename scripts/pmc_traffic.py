#!/usr/bin/env python3
"""HBM bytes per GEMM launch from the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of scripts/gpu_pmc.sh
(bench.py --steps 1 --warmup 1 --no-varlen: (2 timed-loop + 2 host-leg) steps x 4 calls x 12 blocks >= 96 full-size launches per
projection, issued after the short query-encode launches, so the LAST 96 dispatches of a kernel are the T = 131 072-token ones; the
residual kernel serves two projections per block, alternating out-proj / fc2).  Corrections as calibrated on
layernorm_kernel on gfx950: FETCH_SIZE (KiB) x 2, WRITE_SIZE (KiB) as reported.
usage: pmc_traffic.py FETCH.db WRITE.db [MFMA_BUSY.db] > profiles/rNN_pmc_traffic.json"""
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
    # round 5: q | k | V^T leave ONE launch (EPI_QKV = 7); rounds 1-4 had "0" (Q|K store) + "4" (V^T)
    ("qkv_proj [T,2304]x768 q|k row-major + V^T 16-bit", "7", LAUNCHES, None, T * D * 2 + T * 3 * D * 2 + 3 * D * D * 2),
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


def dispatch_durations(path):
    """dispatch_id -> duration in ns from the kernel-trace half of a --pmc pass (rocpd schema: a `kernels` view or a
    rocpd_kernel_dispatch* table carrying start / end per dispatch)."""
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    for tname in sorted(tabs, key=lambda n: (n != "kernels", "kernel_dispatch" not in n, n)):
        cols = [d[1] for d in db.execute(f"pragma table_info('{tname}')")]
        if "start" in cols and "end" in cols and ("dispatch_id" in cols or "id" in cols):
            key = "dispatch_id" if "dispatch_id" in cols else "id"
            try:
                return {int(k_): float(e - s_) for k_, s_, e in db.execute(f"select {key}, start, end from '{tname}'") if k_ is not None}, tname
            except sqlite3.Error:
                continue
    return {}, "no table with (dispatch_id | id, start, end): " + ", ".join(tabs)


def busy_and_clock(path):
    """Per projection shape: MFMA-busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) (the counter
    is 16 x #MFMA: 16 busy cycles per v_mfma_f32_16x16x32, summed over all SIMDs) and the effective shader clock =
    GRBM_GUI_ACTIVE / 8 / kernel duration; time-weighted over the five launches for the bench line."""
    db = sqlite3.connect(path)
    cols = [d[1] for d in db.execute("pragma table_info(counters_collection)")]
    namecol = "kernel_name" if "kernel_name" in cols else "name"
    acc = defaultdict(lambda: defaultdict(float))
    for kn, cn, v, did in db.execute(f"select {namecol}, counter_name, value, dispatch_id from counters_collection"):
        acc[(kn, did)][cn] += float(v)
    dur, src = dispatch_durations(path)
    by_kernel = defaultdict(list)
    for (kn, did), c in sorted(acc.items(), key=lambda kv: kv[0][1]):
        by_kernel[kn].append((c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0), dur.get(int(did))))
    out, wsum, bsum, csum = {}, 0.0, 0.0, 0.0
    for name, key, n, ph, _ in shapes:
        names = sorted([k for k in by_kernel if re.search(r"gemm256d_kernel<[^,]+, %s," % key, k)], key=lambda k: -len(by_kernel[k]))[:1]
        if not names:
            continue
        rows = by_kernel[names[0]][-n:]
        if ph is not None:
            rows = rows[ph[0]::ph[1]]
        mf, gui = median([r[0] for r in rows]), median([r[1] for r in rows])
        ds = [r[2] for r in rows if r[2]]
        d_ns = median(ds) if ds else None
        busy = mf / (gui / 8.0 * 1024.0) if gui else None
        clk = gui / 8.0 / d_ns if (gui and d_ns) else None              # cycles per ns = GHz
        out[name] = {"mfma_busy_frac": busy, "effective_clock_ghz": clk, "duration_us_under_pmc": None if d_ns is None else d_ns / 1e3}
        w = d_ns if d_ns else gui
        if busy is not None:
            wsum += w; bsum += busy * w
            csum += (clk or 0.0) * w
    return out, (bsum / wsum if wsum else None), (csum / wsum if (wsum and csum) else None), src


extra = {}
if len(sys.argv) > 3:
    try:
        per_busy, busy_avg, clk_avg, src = busy_and_clock(sys.argv[3])
        for name, v in per_busy.items():
            per[name].update(v)
        extra = {"gemm_mfma_busy_frac": busy_avg, "gemm_effective_clock_ghz": clk_avg, "durations_from": src}
    except Exception as e:  # noqa: BLE001 -- the traffic figures stand on their own
        extra = {"mfma_busy_error": f"{type(e).__name__}: {e}"[:300]}
print(json.dumps({"note": __doc__.split("usage:")[0].strip(), "tokens_per_launch": T, "per_launch": per,
                  "gemm_avg_hbm_bytes_per_launch": avg, "gemm_avg_algorithmic_bytes_per_launch": alg,
                  "calibration": cal, **extra}, indent=1))
