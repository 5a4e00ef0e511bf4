#!/usr/bin/env python3
"""Where does the filtered scorer launch lose against the bare k-loop?  sgpt_bench_gemm on the scorer's shape
(M = padded query rows, N = documents of a shard, K = 768, f16): EPI_NONE (no store: the k-loop alone) and EPI_SCORE (fp32 score
tile materialised), next to the rates of the filtered pass from scripts/score_bench.py."""
import ctypes as C
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
from sgpt_amd import get_context  # noqa: E402

ctx = get_context("cuda:0")
K = 768
for M in (256, 512, 1024):
    for N in (124928, 499712, 999936):
        row = []
        for epi, name in ((5, "k-loop"), (3, "score-store")):
            if epi == 3 and M * N * 4 > (3 << 30):
                continue
            ts = []
            for _ in range(3):
                ms = C.c_float(0)
                ctx._chk(ctx.lib.sgpt_bench_gemm(ctx.handle, 3, epi, 3 if epi == 5 else 0, M, N, K, 10, C.byref(ms)), "bench")
                ts.append(ms.value)
            med = statistics.median(ts)
            row.append(f"{name} {med * 1e3:8.1f} us {2.0 * M * N * K / med / 1e9:7.1f} TF")
        print(f"M={M:5d} N={N:7d}: " + " | ".join(row))
