#!/usr/bin/env python3
"""One-shape GEMM micro-benchmark: python scripts/gemm_shape.py EPI M N K [out_dtype]  (SGPT_GEMM_DBG=1 for stamps)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from sgpt_amd import get_context
ctx = get_context("cuda:0")
epi, m, n, k = [int(x) for x in sys.argv[1:5]]
odt = int(sys.argv[5]) if len(sys.argv) > 5 else 1
ms = C.c_float(0)
ctx._chk(ctx.lib.sgpt_bench_gemm(ctx.handle, 1, epi, odt, m, n, k, 10, C.byref(ms)), "bench_gemm")
print(f"epi={epi} M={m} N={n} K={k}: {ms.value*1e3:8.1f} us  {2.0*m*n*k/ms.value/1e9:7.1f} TFLOP/s")
