#!/usr/bin/env python3
"""GEMM micro-benchmark over the encoder's five launch shapes (SGPT-125M, 1024 x 128 tokens)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from sgpt_amd import get_context  # noqa: E402

ctx = get_context("cuda:0")
M = int(os.environ.get("M", 131072))
shapes = [("qk    store", 1, 0, 1, M, 1536, 768), ("v     vt   ", 1, 4, 1, M, 768, 768), ("oproj resid", 1, 2, 0, M, 768, 768),
          ("fc1   gelu ", 1, 1, 1, M, 3072, 768), ("fc2   resid", 1, 2, 0, M, 768, 3072)]
if "--fp32" in sys.argv:
    shapes = [("fp32 qkv", 0, 0, 0, 16384, 2304, 768), ("fp32 fc2", 0, 2, 0, 16384, 768, 3072)]
tot_ms = tot_fl = 0
for name, dt, epi, odt, m, n, k in shapes:
    ms = C.c_float(0)
    ctx._chk(ctx.lib.sgpt_bench_gemm(ctx.handle, dt, epi, odt, m, n, k, 10, C.byref(ms)), "bench_gemm")
    fl = 2.0 * m * n * k
    tot_ms += ms.value; tot_fl += fl
    print(f"{name}  M={m} N={n} K={k}: {ms.value*1e3:8.1f} us  {fl/ms.value/1e9:7.1f} TFLOP/s")
print(f"layer GEMM total {tot_ms*1e3:.1f} us -> {tot_fl/tot_ms/1e9:.1f} TFLOP/s")
