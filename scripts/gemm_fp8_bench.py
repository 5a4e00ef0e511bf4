#!/usr/bin/env python3
"""fp8-MFMA GEMM (gemm256q.hip) vs the 16-bit kernel on the MLP shapes, same process, interleaved rounds."""
import ctypes as C
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
from sgpt_amd import get_context  # noqa: E402

ctx = get_context("cuda:0")
shapes = [("125m fc1", 131072, 3072, 768), ("125m fc2", 131072, 768, 3072), ("7b1 fc1", 32768, 16384, 4096), ("7b1 fc2", 32768, 4096, 16384)]
res = {}
for rnd in range(3):
    for name, m, n, k in shapes:
        for dt, tag in ((1, "bf16"), (4, "fp8")):
            for epi in (5, 1 if n > k else 2):
                ms = C.c_float(0)
                ctx._chk(ctx.lib.sgpt_bench_gemm(ctx.handle, dt, epi, 0 if epi == 2 else (1 if dt == 1 else 4), m, n, k, 5, C.byref(ms)), "bench")
                res.setdefault((name, tag, epi), []).append(ms.value)
for name, m, n, k in shapes:
    for tag in ("bf16", "fp8"):
        for epi in (5, 1 if n > k else 2):
            t = statistics.median(res[(name, tag, epi)])
            print(f"{name} {tag:5s} epi {epi}: {t * 1e3:9.1f} us  {2.0 * m * n * k / t / 1e9:8.1f} TFLOP/s")
