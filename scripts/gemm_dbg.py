import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch
from sgpt_amd import get_context
ctx = get_context("cuda:0")
M = 131072
for name, epi, o16, n, k in (("qk store", 0, True, 1536, 768), ("fc1 gelu", 1, True, 3072, 768), ("oproj resid", 2, False, 768, 768), ("fc2 resid", 2, False, 768, 3072), ("none", 5, True, 3072, 768)):
    ms = C.c_float(0)
    print("====", name, flush=True)
    sys.stderr.flush()
    ctx._chk(ctx.lib.sgpt_bench_gemm(ctx.handle, 3, epi, 3 if o16 else 0, M, n, k, 3, C.byref(ms)), "bench")
    print(name, ms.value * 1e3, "us", flush=True)
