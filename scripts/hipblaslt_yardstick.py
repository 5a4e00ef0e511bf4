#!/usr/bin/env python3
"""Yardstick only (never used by the product path): what the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS, bf16,
no epilogue) reaches on the encoder's projection shapes, next to gemm256_kernel's EPI_NONE (no store) rate."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sgpt_amd import get_context
ctx = get_context("cuda:0")
M = 131072
for name, n, k in [("qk", 1536, 768), ("v/out", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)]:
    a = (torch.rand(M, k, device="cuda") - 0.5).to(torch.bfloat16)
    w = ((torch.rand(n, k, device="cuda") - 0.5) * 0.1).to(torch.bfloat16)
    for _ in range(3):
        c = a @ w.T
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        c = a @ w.T
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 10
    ms = C.c_float(0)
    ctx._chk(ctx.lib.sgpt_bench_gemm(ctx.handle, 1, 5, 1, M, n, k, 10, C.byref(ms)), "bench")
    ms0 = C.c_float(0)
    ctx._chk(ctx.lib.sgpt_bench_gemm(ctx.handle, 1, 0, 1, M, n, k, 10, C.byref(ms0)), "bench")
    fl = 2.0 * M * n * k
    print(f"{name:6s} [{M}x{k}]x[{k}x{n}]: vendor bf16 store {fl/dt/1e12:7.1f} TFLOP/s | gemm256 no-store {fl/ms.value/1e9:7.1f} | gemm256 bf16 store {fl/ms0.value/1e9:7.1f}")
