#!/usr/bin/env python3
"""K / N sweep of the bf16 GEMM (fixed-cost vs per-k-step cost)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sgpt_amd import get_context
ctx = get_context("cuda:0")
M = 131072
def run(epi, odt, n, k, m=M):
    ms = C.c_float(0)
    ctx._chk(ctx.lib.sgpt_bench_gemm(ctx.handle, 1, epi, odt, m, n, k, 10, C.byref(ms)), "bench")
    return ms.value
for n in (768,):
    for k in (768, 3072):
        ms = run(0, 1, n, k)
        print(f"store bf16 N={n} K={k}: {ms*1e3:8.1f} us {2.0*M*n*k/ms/1e9:7.1f} TF", flush=True)
for epi, odt, name in ((0, 1, "store"), (5, 1, "none"), (2, 0, "resid"), (1, 1, "gelu"), (4, 1, "vt")):
    ms = run(epi, odt, 768, 768)
    print(f"{name} N=768 K=768: {ms*1e3:8.1f} us {2.0*M*768*768/ms/1e9:7.1f} TF", flush=True)
